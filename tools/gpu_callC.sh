tag=${1:-r02}
mkdir -p gpurun_out
timeout -k 10 200 python tools/bench_conv.py > gpurun_out/conv_L5.log 2>&1; echo "conv L5 rc=$?"; tail -1 gpurun_out/conv_L5.log
timeout -k 10 300 python tools/bench_conv.py 9 0.15 8 131072 262144 > gpurun_out/conv_L9.log 2>&1; echo "conv L9 rc=$?"; tail -1 gpurun_out/conv_L9.log
timeout -k 10 600 python tests/evidence.py cfg5 --sweeps 1e6 --tag $tag 2>&1 | tee gpurun_out/ev_cfg5.log | grep -E "cfg5: (2[0-9]*|4[0-9]*)[0-9]{5} |wrote" | tail -30; echo "cfg5 rc=$?"
