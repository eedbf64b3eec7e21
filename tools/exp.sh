mkdir -p gpurun_out
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', '%.3e prop/s' % d['proposals_per_s'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
run --ladder-steps 500 "$@"
run --ladder-steps 500 --p-logical 0 "$@"
run --ladder-steps 50 --iters 100 "$@"
run --ladder-steps 50 --iters 100 --p-logical 0 "$@"
run --ladder-steps 500 --syndromes 262144 "$@"
