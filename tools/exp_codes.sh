run() { python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', '%.3e prop/s' % d['proposals_per_s'], '%.3e sweeps/s' % d['value'], '%.2f ms' % d['kernel_ms_per_launch'], 'lds', d['config']['lds_bytes_per_workgroup'])"; }
run --ladder-steps 500
run --ladder-steps 200 --L 15 --p 0.18 --Nc 8
run --ladder-steps 200 --L 15 --p 0.18 --Nc 15
run --ladder-steps 200 --L 5 --p 0.10 --Nc 5
run --ladder-steps 200 --code xzzx --L 9 --p 0.15 --eta 100
run --ladder-steps 200 --code xzzx --L 9 --p 0.15
run --ladder-steps 200 --code rotated --L 9 --p 0.17
run --ladder-steps 100 --code rotated --L 21 --p 0.17
