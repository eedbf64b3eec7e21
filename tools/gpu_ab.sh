b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/gpu_tests.log
b cfg5 --config 5
b cfg5 --config 5
b xzzx9 --code xzzx --L 9
b rot9 --code rotated --L 9 --p 0.17
b planar9 --code planar --L 9
b rot13 --code rotated --L 13 --p 0.17
b xzzx15 --code xzzx --L 15
b cfg2 --config 2
b cfg3 --config 3
