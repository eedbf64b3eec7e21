mkdir -p gpurun_out
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>gpurun_out/b_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; b ${tag}_$v --library $E/libqecmc_$v.so "$@"; }
for c in 2 3 4 5; do ab p7 cfg$c --config $c; done 2>&1 | tee gpurun_out/c9_p7.log
ab p7 cfg2 --config 2 2>&1 | tee -a gpurun_out/c9_p7.log
for c in 2 3 4 5; do bash tools/profile_round.sh r03 $c > gpurun_out/prof_r03_cfg$c.log 2>&1; echo "profile cfg$c done"; tail -1 gpurun_out/prof_r03_cfg$c.log; done
