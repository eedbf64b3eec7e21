tag=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 330 bash tools/profile_round.sh $tag 5 > gpurun_out/prof_cfg5.log 2>&1; echo "profile cfg5 rc=$?"
timeout -k 10 600 python tests/evidence.py cfg5 --sweeps 1e6 --tag $tag 2>&1 | tee gpurun_out/ev_cfg5.log | grep -E "wrote" | tail -3; echo "cfg5 rc=$?"
