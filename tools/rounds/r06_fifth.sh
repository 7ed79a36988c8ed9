# round 6, fifth lease (sources frozen): rocprofv3 kernel stats + counter passes of the headline command, the whole GPU suite,
# the driver's bench invocation
set -x
mkdir -p gpurun_out/r06e
bash tools/profile_round.sh r06 > gpurun_out/r06e/profile_round.log 2>&1
cp gpurun_out/prof_r06/r06_hbm_traffic.json gpurun_out/r06e/ 2>/dev/null
find gpurun_out/prof_r06/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06e/r06_kernel_stats.csv \;
cp gpurun_out/prof_r06/bench_stats.json gpurun_out/r06e/r06_bench_under_rocprof.json
rm -rf gpurun_out/prof_r06
mkdir -p profiles && cp gpurun_out/r06e/r06_hbm_traffic.json profiles/r06_hbm_traffic.json
timeout 3000 python -m pytest tests -q -m gpu -s --durations=15 -p no:cacheprovider > gpurun_out/r06e/gpu_tests_full.txt 2>&1
tail -25 gpurun_out/r06e/gpu_tests_full.txt | cut -c1-200 > gpurun_out/r06e/gpu_tests_tail.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06e/bench.out 2> gpurun_out/r06e/bench.err ) 2> gpurun_out/r06e/bench_time.txt
cp bench_detail.json gpurun_out/r06e/
tail -1 gpurun_out/r06e/bench.out | cut -c1-1200
cat gpurun_out/r06e/bench_time.txt
tail -4 gpurun_out/r06e/gpu_tests_tail.txt
