mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c4_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c4_tests.log
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; b ${tag}_$v --library $E/libqecmc_$v.so "$@"; }
ab base cfg4 --config 4
ab base cfg2 --config 2
ab base xzzx9 --config 2 --code xzzx
ab base rot9 --config 2 --code rotated
ab base planar9 --config 2 --code planar
ab base cfg4 --config 4
ab base cfg2 --config 2
ab base xzzx9 --config 2 --code xzzx
ab base rot9 --config 2 --code rotated
