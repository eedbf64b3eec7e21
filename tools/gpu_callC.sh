mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/gpu_tests.log
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'])"; }
b cfg2 --config 2
b cfg4 --config 4
QECMC_TUNE=1 b cfg4_topprio --config 4
b cfg3 --config 3
QECMC_TUNE=1 b cfg3_topprio --config 3
b cfg5 --config 5
QECMC_TUNE=1 b cfg5_topprio --config 5
QECMC_TUNE=1 b cfg2_topprio --config 2
timeout -k 10 200 python bench.py --config 4 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; echo "bench cfg4 rc=$?"
timeout -k 10 200 python tests/evidence.py cfg5 --sweeps 1e5 --tag r02 > gpurun_out/ev_cfg5.log 2>&1; echo "cfg5 rc=$?"; tail -3 gpurun_out/ev_cfg5.log | cut -c1-400
