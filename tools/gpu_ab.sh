b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
b cfg2 --config 2
b xzzx9 --code xzzx --L 9
b rot9 --code rotated --L 9 --p 0.17
b planar9 --code planar --L 9
b L7c7 --L 7 --Nc 7
b L5c5 --L 5 --Nc 5 --p 0.10
