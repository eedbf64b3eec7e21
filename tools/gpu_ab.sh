# A/B list of quick bench lines: this build against tools/exp_libs/libqecmc_<v>.so in the same call (boxes differ by 1-2 %)
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; QECMC_LIBRARY=$E/libqecmc_$v.so b ${tag}_$v "$@"; }
for i in 1 2; do
ab base cfg4 --config 4
ab base cfg2 --config 2
done
ab base cfg3 --config 3
ab base cfg5 --config 5
