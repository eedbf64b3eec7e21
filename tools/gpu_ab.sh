b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'sweeps %.3e' % d['value'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'], 'lds', d['config']['lds_bytes_per_workgroup'])"; }
mkdir -p gpurun_out
b L15c15 --config 3 --Nc 15
b L13 --L 13
b L11 --L 11
b L7c7 --L 7 --Nc 7
b L5c5 --L 5 --Nc 5 --p 0.10
b xzzx9 --code xzzx --L 9
b rot9 --code rotated --L 9 --p 0.17
b planar9 --code planar --L 9
b cfg2sweep --config 2 --scan sweep
