run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ.get('QECMC_STAGGER'), '$*', '%.3e prop/s' % d['proposals_per_s'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
for s in 0 1 2 3 4; do QECMC_STAGGER=$s run --ladder-steps 500; done
