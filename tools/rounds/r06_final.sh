# round 6, final lease: the one-product bf16 kernels issue 16x16x32 MFMAs (source hash 8bc82ce832a81f42; the f32 headline kernels are
# untouched): counter passes + kernel stats of the headline command, smoke, the driver's bench invocation, on the new hash
set -x
mkdir -p gpurun_out/r06q
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06q/smoke.txt 2>&1
bash tools/profile_round.sh r06 > gpurun_out/r06q/profile_round.log 2>&1
cp gpurun_out/prof_r06/r06_hbm_traffic.json gpurun_out/r06q/ 2>/dev/null
find gpurun_out/prof_r06/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06q/r06_kernel_stats.csv \;
cp gpurun_out/prof_r06/bench_stats.json gpurun_out/r06q/r06_bench_under_rocprof.json
rm -rf gpurun_out/prof_r06
mkdir -p profiles && cp gpurun_out/r06q/r06_hbm_traffic.json profiles/r06_hbm_traffic.json
timeout 900 bash tools/pmc_amp_fwd.sh > gpurun_out/r06q/pmc_amp_fwd.log 2>&1
cp gpurun_out/pmc_amp_fwd/summary.json gpurun_out/r06q/pmc_amp_fwd_summary.json
rm -rf gpurun_out/pmc_amp_fwd/sq gpurun_out/pmc_amp_fwd/mem gpurun_out/pmc_amp_fwd/fetch gpurun_out/pmc_amp_fwd/write
export TMPDIR=/tmp
PN_FORWARD_MATH=bf16 PN_STEPS=4 rocprofv3 --kernel-trace --stats -d gpurun_out/r06q/stats -o p --output-format csv -- python tools/amp_step.py > gpurun_out/r06q/amp_step.log 2>&1
find gpurun_out/r06q/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06q/r06_kernel_stats_amp_full.csv \;
rm -rf gpurun_out/r06q/stats
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06q/bench.out 2> gpurun_out/r06q/bench.err ) 2> gpurun_out/r06q/bench_time.txt
cp bench_detail.json gpurun_out/r06q/
( time timeout 1700 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r06q/tests.txt 2>&1 ) 2> gpurun_out/r06q/tests_time.txt
tail -1 gpurun_out/r06q/bench.out | cut -c1-700
cat gpurun_out/r06q/bench_time.txt gpurun_out/r06q/smoke.txt | tail -6
tail -2 gpurun_out/r06q/tests.txt
