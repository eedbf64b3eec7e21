tag=${1:-r02}
mkdir -p gpurun_out
for c in ${CONFIGS:-3 4 5}; do timeout -k 10 330 bash tools/profile_round.sh $tag $c > gpurun_out/prof_cfg$c.log 2>&1; echo "profile cfg$c rc=$?"; done
