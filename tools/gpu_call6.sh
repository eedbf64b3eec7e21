mkdir -p gpurun_out
export PYTHONFAULTHANDLER=1
timeout -k 10 300 python -m pytest tests/test_gpu_parity_alpha.py -m gpu -x -v > gpurun_out/c6_alpha.log 2>&1; rc=$?; echo "alpha rc=$rc"; tail -5 gpurun_out/c6_alpha.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_parity_surf.py -m gpu -x -v > gpurun_out/c6_surf.log 2>&1; rc=$?; echo "surf rc=$rc"; tail -5 gpurun_out/c6_surf.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -m gpu -x -v -k "work_queue_biased" > gpurun_out/c6_q.log 2>&1; rc=$?; echo "queue rc=$rc"; tail -12 gpurun_out/c6_q.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/c6_r2.log 2>&1; rc=$?; echo "round2 rc=$rc"; tail -5 gpurun_out/c6_r2.log
