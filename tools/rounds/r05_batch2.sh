set -x
mkdir -p gpurun_out/r05e
python -m pytest tests/test_hip_train.py tests/test_hip_config_holes.py tests/test_hip_bwd_bf16.py -q -m gpu -s -k "different_math_modes or full_size_one_hidden or kernel_variants_agree or full_size_step or yardstick" 2>&1 | grep -E "full-size|passed|failed|Error|assert|backward bf16" > gpurun_out/r05e/tests.txt
bash tools/profile_round.sh r05 > gpurun_out/r05e/profile_round.log 2>&1
cp gpurun_out/prof_r05/r05_hbm_traffic.json gpurun_out/r05e/ 2>/dev/null
find gpurun_out/prof_r05/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05e/r05_kernel_stats.csv \;
cp gpurun_out/prof_r05/bench_stats.json gpurun_out/r05e/r05_bench_under_rocprof.json
rm -rf gpurun_out/prof_r05
tail -5 gpurun_out/r05e/tests.txt
