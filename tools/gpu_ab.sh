b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'], 'lds', d['config']['lds_bytes_per_workgroup'])"; }
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
b cfg5 --config 5
QECMC_TUNE=4 b cfg5_nolut --config 5
b xzzx9 --code xzzx --L 9
QECMC_TUNE=4 b xzzx9_nolut --code xzzx --L 9
b rot9 --code rotated --L 9 --p 0.17
QECMC_TUNE=4 b rot9_nolut --code rotated --L 9 --p 0.17
b planar9 --code planar --L 9
QECMC_TUNE=4 b planar9_nolut --code planar --L 9
b rot13 --code rotated --L 13 --p 0.17
QECMC_TUNE=4 b rot13_nolut --code rotated --L 13 --p 0.17
b xzzx15 --code xzzx --L 15
QECMC_TUNE=4 b xzzx15_nolut --code xzzx --L 15
b cfg5 --config 5
QECMC_TUNE=4 b cfg5_nolut --config 5
b cfg3 --config 3
b cfg2 --config 2
