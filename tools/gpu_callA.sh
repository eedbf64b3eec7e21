tag=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
for c in 2 3; do timeout -k 10 330 bash tools/profile_round.sh $tag $c > gpurun_out/prof_cfg$c.log 2>&1; echo "profile cfg$c rc=$?"; done
timeout -k 10 300 python tests/evidence.py headline --tag $tag > gpurun_out/ev_headline.log 2>&1; echo "headline rc=$?"; tail -2 gpurun_out/ev_headline.log
