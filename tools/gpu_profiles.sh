mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
for c in ${CONFIGS:-2 3 4 5}; do timeout -k 10 300 bash tools/profile_round.sh r02 $c > gpurun_out/prof_cfg$c.log 2>&1; echo "profile cfg$c rc=$?"; done
