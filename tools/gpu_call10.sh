mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout -k 10 600 python -m pytest tests/test_gpu_colour.py -m gpu -q -x > gpurun_out/c10_tests.log 2>&1; rc=$?; echo "colour tests rc=$rc"; tail -6 gpurun_out/c10_tests.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python tools/latency.py --only-colour > gpurun_out/r03_latency_colour.json 2> gpurun_out/r03_latency_colour.err; tail -c 1600 gpurun_out/r03_latency_colour.err
QECMC_FUZZ_TRACE=1 timeout -k 10 500 python tests/fuzz_gpu.py 3000 41 > gpurun_out/fuzz_41.log 2>&1; grep -v "^case" gpurun_out/fuzz_41.log | tail -3 | cut -c1-300
timeout -k 10 200 python tests/fuzz_gpu.py other 1000 42 > gpurun_out/fuzz_other_42.log 2>&1; tail -2 gpurun_out/fuzz_other_42.log | cut -c1-300
