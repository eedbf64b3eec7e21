b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x -k "drawn_ahead or pteq_batch or ladder" > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
b cfg3 --config 3
b cfg3 --config 3
b cfg5 --config 5
b cfg5 --config 5
b L13c9 --L 13 --Nc 9
b cfg3c15 --config 3 --Nc 15
