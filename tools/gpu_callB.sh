# GPU call B: tests, bench lines, full-size evidence, rocprof per configuration; biased-rule work last
tag=${1:-r02}
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -k "not biased and not alpha and not xzzxb and not surf" > gpurun_out/gpu_tests_main.log 2>&1; echo "tests(main) rc=$?"; tail -3 gpurun_out/gpu_tests_main.log
for c in 2 3 5; do timeout -k 10 240 python bench.py --config $c > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err; echo "bench cfg$c rc=$?"; done
timeout -k 10 300 python tests/evidence.py headline --tag $tag > gpurun_out/ev_headline.log 2>&1; echo "headline rc=$?"
timeout -k 10 300 python tests/evidence.py cfg3 --tag $tag > gpurun_out/ev_cfg3.log 2>&1; echo "cfg3 rc=$?"
timeout -k 10 200 python tests/evidence.py cfg5 --sweeps 1e5 --tag $tag > gpurun_out/ev_cfg5.log 2>&1; echo "cfg5 rc=$?"
for c in 2 3 5; do timeout -k 10 300 bash tools/profile_round.sh $tag $c > gpurun_out/prof_cfg$c.log 2>&1; echo "profile cfg$c rc=$?"; done
python -m pytest tests -m gpu -q -k "biased or alpha or xzzxb or surf" > gpurun_out/gpu_tests_biased.log 2>&1; echo "tests(biased) rc=$?"; tail -15 gpurun_out/gpu_tests_biased.log
timeout -k 10 240 python bench.py --config 4 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err; echo "bench cfg4 rc=$?"
timeout -k 10 300 bash tools/profile_round.sh $tag 4 > gpurun_out/prof_cfg4.log 2>&1; echo "profile cfg4 rc=$?"
