b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
mkdir -p gpurun_out
b cfg3 --config 3
b cfg3_nolog --config 3 --p-logical 0
b cfg3_it100 --config 3 --iters 100 --ladder-steps 1000
b cfg3_it40 --config 3 --iters 40 --ladder-steps 2500
b cfg3_Nc4 --config 3 --Nc 4
b cfg3_half --config 3 --syndromes 65536
b cfg5 --config 5
b cfg5_nolog --config 5 --p-logical 0
b cfg5_it100 --config 5 --iters 100 --ladder-steps 1000
b cfg2 --config 2
b cfg2_nolog --config 2 --p-logical 0
b cfg2_it100 --config 2 --iters 100 --ladder-steps 1000
