mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/gpu_tests.log
b() { tag=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', '%.3e prop/s' % d['proposals_per_s'], 'frac %.3f' % d['roofline']['frac'], '%.2f ms' % d['kernel_ms_per_launch'], 'lds', d['config']['lds_bytes_per_workgroup'])"; }
b cfg2_delut --config 2
QECMC_TUNE=4 b cfg2_nolut --config 2
b cfg2_delut --config 2
QECMC_TUNE=4 b cfg2_nolut --config 2
b L5 --L 5 --Nc 5 --p 0.1
b cfg3 --config 3
bash tools/pmc.sh cfg2 --config 2 > gpurun_out/pmc_cfg2.log 2>&1; tail -18 gpurun_out/pmc_cfg2.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "torchrun world=1 (RCCL init + gather) rc=$?"; cut -c1-300 gpurun_out/bench_torchrun1.json
