# round 6, sixth lease (FINAL sources): rocprofv3 kernel stats + counter passes of the headline command, the whole GPU suite,
# the driver's bench invocation, counter passes over the AMP-class forward, the AMP shape sweep
set -x
mkdir -p gpurun_out/r06g
bash tools/profile_round.sh r06 > gpurun_out/r06g/profile_round.log 2>&1
cp gpurun_out/prof_r06/r06_hbm_traffic.json gpurun_out/r06g/ 2>/dev/null
find gpurun_out/prof_r06/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06g/r06_kernel_stats.csv \;
cp gpurun_out/prof_r06/bench_stats.json gpurun_out/r06g/r06_bench_under_rocprof.json
rm -rf gpurun_out/prof_r06
mkdir -p profiles && cp gpurun_out/r06g/r06_hbm_traffic.json profiles/r06_hbm_traffic.json
timeout 3000 python -m pytest tests -q -m gpu -s --durations=15 -p no:cacheprovider > gpurun_out/r06g/gpu_tests_full.txt 2>&1
tail -25 gpurun_out/r06g/gpu_tests_full.txt | cut -c1-200 > gpurun_out/r06g/gpu_tests_tail.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06g/bench.out 2> gpurun_out/r06g/bench.err ) 2> gpurun_out/r06g/bench_time.txt
cp bench_detail.json gpurun_out/r06g/
timeout 900 bash tools/pmc_amp_fwd.sh > gpurun_out/r06g/pmc_amp_fwd.log 2>&1
cp gpurun_out/pmc_amp_fwd/summary.json gpurun_out/r06g/pmc_amp_fwd_summary.json
rm -rf gpurun_out/pmc_amp_fwd/sq gpurun_out/pmc_amp_fwd/mem gpurun_out/pmc_amp_fwd/fetch gpurun_out/pmc_amp_fwd/write
PN_SWEEP_MODE=amp timeout 900 python tools/shape_sweep.py gpurun_out/r06g/r06_shape_sweep_amp.json > gpurun_out/r06g/shape_sweep_amp.log 2>&1
tail -1 gpurun_out/r06g/bench.out | cut -c1-1200
cat gpurun_out/r06g/bench_time.txt
tail -4 gpurun_out/r06g/gpu_tests_tail.txt
