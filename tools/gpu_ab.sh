b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
b cfg4 --config 4
b cfg4it100 --config 4 --iters 100 --ladder-steps 1000
python bench.py --config 4 --steps 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg4 histogram_match', d.get('histogram_match'))"
timeout -k 10 400 python tests/fuzz_gpu.py 1500 9 > gpurun_out/fuzz9.log 2>&1; echo rc=$?; tail -1 gpurun_out/fuzz9.log
