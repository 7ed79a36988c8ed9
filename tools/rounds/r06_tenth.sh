# round 6, tenth lease: the translation unit was split into four included parts (device code byte-identical, the source hash is
# new): counter passes + kernel stats of the headline command, smoke, the driver's bench invocation, on the new hash
set -x
mkdir -p gpurun_out/r06k
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06k/smoke.txt 2>&1
bash tools/profile_round.sh r06 > gpurun_out/r06k/profile_round.log 2>&1
cp gpurun_out/prof_r06/r06_hbm_traffic.json gpurun_out/r06k/ 2>/dev/null
find gpurun_out/prof_r06/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06k/r06_kernel_stats.csv \;
cp gpurun_out/prof_r06/bench_stats.json gpurun_out/r06k/r06_bench_under_rocprof.json
rm -rf gpurun_out/prof_r06
mkdir -p profiles && cp gpurun_out/r06k/r06_hbm_traffic.json profiles/r06_hbm_traffic.json
timeout 900 bash tools/pmc_amp_fwd.sh > gpurun_out/r06k/pmc_amp_fwd.log 2>&1
cp gpurun_out/pmc_amp_fwd/summary.json gpurun_out/r06k/pmc_amp_fwd_summary.json
rm -rf gpurun_out/pmc_amp_fwd/sq gpurun_out/pmc_amp_fwd/mem gpurun_out/pmc_amp_fwd/fetch gpurun_out/pmc_amp_fwd/write
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06k/bench.out 2> gpurun_out/r06k/bench.err ) 2> gpurun_out/r06k/bench_time.txt
cp bench_detail.json gpurun_out/r06k/
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_fwd_bf16.py tests/test_hip_bwd_bf16.py -q -p no:cacheprovider > gpurun_out/r06k/tests.txt 2>&1
tail -1 gpurun_out/r06k/bench.out | cut -c1-700
cat gpurun_out/r06k/bench_time.txt gpurun_out/r06k/smoke.txt | tail -6
tail -2 gpurun_out/r06k/tests.txt
