b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; QECMC_LIBRARY=$E/libqecmc_$v.so b ${tag}_$v "$@"; }
ab topprio cfg3 --config 3
ab topprio cfg5 --config 5
ab topprio cfg3 --config 3
ab topprio cfg5 --config 5
ab topprio cfg3c15 --config 3 --Nc 15
ab topprio L13c9 --L 13 --Nc 9
