b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
W=$PWD/tools/exp_libs/libqecmc_c8.so
ab() { tag=$1; shift; b $tag "$@"; QECMC_LIBRARY=$W b ${tag}_c8 "$@"; }
ab cfg3 --config 3
ab cfg5 --config 5
ab cfg3 --config 3
ab cfg5 --config 5
ab L10 --L 10
ab xzzx15 --code xzzx --L 15
ab L13c9 --L 13 --Nc 9
