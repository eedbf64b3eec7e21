# round 3, call 2: the whole GPU test-suite on the rewritten biased path, config 4 / 2 against the round-2 build
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/c2_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c2_tests.log
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; QECMC_LIBRARY=$E/libqecmc_$v.so b ${tag}_$v "$@"; }
ab base cfg4 --config 4
ab base cfg4 --config 4
b cfg4_rot --config 4 --code rotated
b cfg4_L5 --config 4 --L 5 --Nc 5
