mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_colour.py tests/test_gpu_stats.py -m gpu -q > gpurun_out/c5_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/c5_tests.log
timeout -k 10 600 python tools/latency.py --only-colour > gpurun_out/r03_latency_colour.json 2> gpurun_out/r03_latency_colour.err; tail -c 1500 gpurun_out/r03_latency_colour.err
