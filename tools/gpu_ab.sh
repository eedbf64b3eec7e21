b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'], 'lds', d['config']['lds_bytes_per_workgroup'])"; }
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
for r in 1 2 3; do
b cfg2 --config 2
QECMC_TUNE=16 b cfg2_rel --config 2
done
b L7c7 --L 7 --Nc 7
QECMC_TUNE=16 b L7c7_rel --L 7 --Nc 7
b L5c5 --L 5 --Nc 5 --p 0.10
QECMC_TUNE=16 b L5c5_rel --L 5 --Nc 5 --p 0.10
b L9c5 --L 9 --Nc 5
QECMC_TUNE=16 b L9c5_rel --L 9 --Nc 5
