mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity_surf.py tests/test_gpu_parity_alpha.py tests/test_gpu_round2.py tests/test_gpu_stats.py -m gpu -q -x -k "biased or alpha or surf or queue or parity or q3" > gpurun_out/c11_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/c11_tests.log; [ $rc -eq 0 ] || exit 1
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>gpurun_out/b_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; b ${tag}_$v --library $E/libqecmc_$v.so "$@"; }
ab r3a cfg4 --config 4
ab r3a cfg4 --config 4
ab r3a cfg4rot --config 4 --code rotated
ab r3a cfg4L5 --config 4 --L 5 --Nc 5
