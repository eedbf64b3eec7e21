mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_stats.py -m gpu -x -q > gpurun_out/c3_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/c3_tests.log
timeout -k 10 600 python tools/f5_margins.py > gpurun_out/f5_margins.json 2> gpurun_out/f5_margins.err; cat gpurun_out/f5_margins.json
