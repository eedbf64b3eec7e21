# GPU call A: the gpu test-suite, one bench line per BASELINE configuration, the full-size evidence runs
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; rc=$?; tail -5 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
for c in 2 3 4 5; do python bench.py --config $c > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err || exit 1; echo "bench cfg$c done"; done
python tests/evidence.py headline --tag ${1:-r02a} > gpurun_out/ev_headline.log 2>&1 || { tail -20 gpurun_out/ev_headline.log; exit 1; }
echo headline done
python tests/evidence.py cfg3 --tag ${1:-r02a} > gpurun_out/ev_cfg3.log 2>&1 || { tail -20 gpurun_out/ev_cfg3.log; exit 1; }
echo cfg3 done
python tests/evidence.py cfg5 --sweeps 1e5 --tag ${1:-r02a} > gpurun_out/ev_cfg5.log 2>&1 || { tail -20 gpurun_out/ev_cfg5.log; exit 1; }
echo cfg5 done
