b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; QECMC_LIBRARY=$E/libqecmc_$v.so b ${tag}_$v "$@"; }
ab prephi cfg3 --config 3
ab prephi cfg5 --config 5
ab prephi cfg4 --config 4
ab prephi cfg3 --config 3
ab prephi cfg5 --config 5
ab prephi cfg4 --config 4
