#!/bin/bash
# One parametrised GPU-box script (replaces the numbered gpu_call*.sh of earlier rounds): `tools/gpu_run.sh <what> [args]`, run as
#   gpurun --timeout S -- 'bash tools/gpu_run.sh <what> ...'
# Everything it writes goes under gpurun_out/ (merged back by gpurun).  Steps are joined with && : a failed GPU step ends the call.
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
b() {  # b <tag> <bench args...>: one bench line, condensed
    tag=$1; shift
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 3 "$@" 2> gpurun_out/bench_$tag.err | tee gpurun_out/bench_$tag.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d.get('proposals_per_s', d.get('useful_proposals_per_s', 0.0)), 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'], 'lds', d['config']['lds_bytes_per_workgroup'])"
}
case "$1" in
  wave-first)   # first contact of the scan = wave kernel: its parity tests, then A/B against the random scan on the headline shape
    timeout -k 10 900 python -m pytest tests/test_gpu_wave.py -x -q > gpurun_out/wave_tests.log 2>&1; rc=$?; tail -15 gpurun_out/wave_tests.log
    [ $rc -eq 0 ] && b cfg2_random --config 2 --scan random && b cfg2_wave --config 2 --scan wave && b cfg2_random_b --config 2 --scan random && b cfg2_wave_b --config 2 --scan wave
    ;;
  wave-cfgs)    # scan = wave against the random scan on BASELINE configurations 2, 3, 5 (same box, alternating)
    timeout -k 10 900 python -m pytest tests/test_gpu_wave.py -x -q > gpurun_out/wave_tests.log 2>&1; rc=$?; tail -5 gpurun_out/wave_tests.log
    [ $rc -eq 0 ] && for c in 2 3 5; do b cfg${c}_random --config $c --scan random && b cfg${c}_wave --config $c --scan wave || exit 1; done
    ;;
  tests)        # tests [pytest args]: the GPU suite (or part of it)
    shift
    timeout -k 10 1100 python -m pytest tests -m gpu -q "$@" > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/gpu_tests.log
    ;;
  bench)        # bench <tag> <bench args>
    shift; b "$@"
    ;;
  ab)           # ab <other libqecmc.so> <bench args>: this build against another one of the same ABI, alternating on one box (bench.py --library)
    shift; other=$1; shift
    for i in 1 2; do b this_$i "$@" && b other_$i --library "$other" "$@" || exit 1; done
    ;;
  profiles)     # profiles <tag> <config ...>: tools/profile_round.sh (kernel trace + separate PMC passes + bench) for BASELINE configurations
    shift; tag=$1; shift
    for c in "$@"; do bash tools/profile_round.sh $tag $c > gpurun_out/prof_${tag}_cfg$c.log 2>&1 || exit 1; tail -2 gpurun_out/prof_${tag}_cfg$c.log; done
    ;;
  criterion)    # criterion [bench args]: the reference's default route (bench.py --criterion), the line kept under gpurun_out/
    shift
    timeout -k 10 1000 python bench.py --criterion "$@" > gpurun_out/criterion_bench.json 2> gpurun_out/criterion_bench.err; tail -c 1500 gpurun_out/criterion_bench.json
    ;;
  bench-all)    # every bench mode once, condensed (a regression guard after a kernel change: a layout change that costs one mode a workgroup per CU shows here)
    b cfg2 --config 2 && b cfg3 --config 3 && b cfg4 --config 4 && b cfg5 --config 5 && b nc9 --config 2 --Nc 9 && b xzzx9 --config 2 --code xzzx && \
    b alpha --alpha-route && b crit --criterion --steps 1 --warmup 0 && b critalpha --alpha-route --criterion --steps 1 --warmup 0
    ;;
  round-end)    # the driver's round-end sequence rehearsed: GPU tests, smoke(), the default bench line
    timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -4 gpurun_out/gpu_tests.log
    [ $rc -eq 0 ] && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" && python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err && tail -c 600 gpurun_out/bench_default.json
    ;;
  *) echo "unknown step $1"; exit 2 ;;
esac
