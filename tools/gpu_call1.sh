# round 3, call 1: parity of the rewritten biased / alpha proposal path, then config 4 against the round-2 build
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity_surf.py tests/test_gpu_parity_alpha.py -m gpu -x -q > gpurun_out/c1_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/c1_tests.log
bash tools/gpu_ab.sh 2>&1 | tee gpurun_out/c1_ab.log
