b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
C=$PWD/tools/exp_libs/libqecmc_noahead.so
ab() { tag=$1; shift; b $tag "$@"; QECMC_LIBRARY=$C b ${tag}_noahead "$@"; }
ab xzzx19 --code xzzx --L 19
ab xzzx21 --code xzzx --L 21
ab planar13 --code planar --L 13
ab rot17 --code rotated --L 17 --p 0.17
ab rot21c12 --config 5 --Nc 12
ab rot9c12 --code rotated --L 9 --p 0.17 --Nc 12
ab L13c9 --L 13 --Nc 9
ab L9c12 --L 9 --Nc 12
ab L16c9 --L 16 --Nc 9
