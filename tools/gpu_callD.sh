# the driver's round-end sequence, rehearsed: GPU test-suite, smoke(), the default bench line, the launcher form with one rank
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench_torchrun_world1.json 2> gpurun_out/bench_torchrun.err; echo "torchrun rc=$?"
python tools/check_sharded_gpu.py > gpurun_out/check_sharded.log 2>&1; echo "sharded rc=$?"; tail -2 gpurun_out/check_sharded.log
