# last call of the round: tests, config 2's profile set, the default bench line
tag=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
timeout -k 10 330 bash tools/profile_round.sh $tag 2 > gpurun_out/prof_cfg2.log 2>&1; echo "profile cfg2 rc=$?"
timeout -k 10 300 python tests/evidence.py headline --tag $tag > gpurun_out/ev_headline.log 2>&1; echo "headline rc=$?"; tail -1 gpurun_out/ev_headline.log | cut -c1-400
