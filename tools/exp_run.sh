#!/bin/bash
# runs bench.py (2000-step headline launch) against each experiment library in tools/exp_libs/
run() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sweep "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-28s' % '$LBL', '%.3e prop/s' % d['proposals_per_s'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
LBL=base run
for f in tools/exp_libs/libqecmc_*.so; do
  v=$(basename $f .so); v=${v#libqecmc_}
  export QECMC_LIBRARY=$PWD/$f; LBL=$v run; LBL="$v p_logical=0" run --p-logical 0
done
unset QECMC_LIBRARY
LBL="base p_logical=0" run --p-logical 0
LBL="base iters=100" run --iters 100 --ladder-steps 200
