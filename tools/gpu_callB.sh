tag=${1:-r02}
mkdir -p gpurun_out
for c in 4 5; do timeout -k 10 330 bash tools/profile_round.sh $tag $c > gpurun_out/prof_cfg$c.log 2>&1; echo "profile cfg$c rc=$?"; done
timeout -k 10 300 python tests/evidence.py cfg3 --tag $tag > gpurun_out/ev_cfg3.log 2>&1; echo "cfg3 rc=$?"; tail -2 gpurun_out/ev_cfg3.log
