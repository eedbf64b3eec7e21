# A/B of an experimental library against this build, alternating in one call: tools/gpu_ab2.sh <variant> <bench args...>
v=$1; shift
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
for i in 1 2; do b base "$@"; b $v --library $E/libqecmc_$v.so "$@"; done
