mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/c7_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -6 gpurun_out/c7_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/latency.py --only-colour > gpurun_out/r03_latency_colour.json 2> gpurun_out/r03_latency_colour.err; tail -c 1200 gpurun_out/r03_latency_colour.err
timeout -k 10 300 python tools/bench_conv.py alpha 5 0.15 100 5 262144 65536 2>&1 | tail -4
timeout -k 10 300 python tools/bench_conv.py alpha 7 0.15 100 7 131072 65536 2>&1 | tail -4
