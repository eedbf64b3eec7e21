mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/gpu_tests.log
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'], 'lds', d['config']['lds_bytes_per_workgroup'])"; }
b cfg3_pre --config 3
b cfg5_pre --config 5
b cfg3_Nc15 --config 3 --Nc 15 --syndromes 65536
QECMC_TUNE=2 b cfg3_Nc15_nopre --config 3 --Nc 15 --syndromes 65536
b rot21_Nc12 --config 5 --Nc 12
b L13 --L 13 --syndromes 65536
