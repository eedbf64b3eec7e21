# final evidence of the round: bench lines, rocprof sets, full-size runs
tag=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
for c in 2 3 4 5; do timeout -k 10 300 bash tools/profile_round.sh $tag $c > gpurun_out/prof_cfg$c.log 2>&1; echo "profile cfg$c rc=$?"; done
timeout -k 10 300 python tests/evidence.py headline --tag $tag > gpurun_out/ev_headline.log 2>&1; echo "headline rc=$?"
timeout -k 10 300 python tests/evidence.py cfg3 --tag $tag > gpurun_out/ev_cfg3.log 2>&1; echo "cfg3 rc=$?"
timeout -k 10 600 python tests/evidence.py cfg5 --sweeps 1e6 --tag $tag 2>&1 | tee gpurun_out/ev_cfg5.log | grep -E "cfg5: (2[0-9]*|4[0-9]*)[0-9]{5} |wrote" | tail -30; echo "cfg5 rc=$?"
