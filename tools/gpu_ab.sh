b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/gpu_tests.log
b cfg2 --config 2
b xzzx9 --code xzzx --L 9
QECMC_TUNE=8 b xzzx9_nossw --code xzzx --L 9
b rot9 --code rotated --L 9 --p 0.17
QECMC_TUNE=8 b rot9_nossw --code rotated --L 9 --p 0.17
b planar9 --code planar --L 9
QECMC_TUNE=8 b planar9_nossw --code planar --L 9
b L7 --L 7 --Nc 7
QECMC_TUNE=8 b L7_nossw --L 7 --Nc 7
b L10 --L 10
QECMC_TUNE=8 b L10_nossw --L 10
