mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/gpu_tests.log
python tools/bench_conv.py 2>&1 | tail -12
