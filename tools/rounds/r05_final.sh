set -x
mkdir -p gpurun_out/r05z
python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tail -30 > gpurun_out/r05z/gpu_tests.txt
bash tools/profile_round.sh r05 > gpurun_out/r05z/profile_round.log 2>&1
cp gpurun_out/prof_r05/r05_hbm_traffic.json gpurun_out/r05z/ 2>/dev/null
find gpurun_out/prof_r05/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05z/r05_kernel_stats.csv \;
cp gpurun_out/prof_r05/bench_stats.json gpurun_out/r05z/r05_bench_under_rocprof.json
rm -rf gpurun_out/prof_r05
mkdir -p profiles && cp gpurun_out/r05z/r05_hbm_traffic.json profiles/r05_hbm_traffic.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05z/bench.out 2> gpurun_out/r05z/bench.err
cp bench_detail.json gpurun_out/r05z/
tail -1 gpurun_out/r05z/bench.out | cut -c1-700
tail -4 gpurun_out/r05z/gpu_tests.txt
