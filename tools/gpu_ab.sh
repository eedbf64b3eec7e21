b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'], d.get('histogram_match'))"; }
E=$PWD/tools/exp_libs
ab() { v=$1; tag=$2; shift; shift; b $tag "$@"; QECMC_LIBRARY=$E/libqecmc_$v.so b ${tag}_$v "$@"; }
ab prep2 cfg3 --config 3
ab prep2 cfg5 --config 5
ab prep2 cfg3 --config 3
ab prep2 cfg5 --config 5
ab prep2 L10 --L 10
ab prep2 cfg3c15 --config 3 --Nc 15
QECMC_LIBRARY=$E/libqecmc_prep2.so timeout -k 10 300 python -m pytest tests -m gpu -q -x -k "pteq_batch or ladder_step or drawn_ahead or size_sweep" 2>&1 | tail -2
