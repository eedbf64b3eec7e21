mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_colour.py -m gpu -q -x > gpurun_out/c8_tests.log 2>&1; rc=$?; echo "colour tests rc=$rc"; tail -4 gpurun_out/c8_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/latency.py --only-colour > gpurun_out/r03_latency_colour.json 2> gpurun_out/r03_latency_colour.err; tail -c 1000 gpurun_out/r03_latency_colour.err
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; b ${tag}_$v --library $E/libqecmc_$v.so "$@"; }
for c in 2 3 4 5; do ab p7 cfg$c --config $c; done
ab p7 cfg2 --config 2
b cfg2_it100 --config 2 --iters 100 --ladder-steps 1000
b cfg2_it100_p7 --config 2 --iters 100 --ladder-steps 1000 --library $E/libqecmc_p7.so
