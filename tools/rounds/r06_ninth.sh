# round 6, ninth lease: rocprofv3 kernel statistics of the full AMP-class train step (bf16x3 base, forward + backward bf16)
set -x
mkdir -p gpurun_out/r06j
export TMPDIR=/tmp
PN_FORWARD_MATH=bf16 PN_STEPS=4 rocprofv3 --kernel-trace --stats -d gpurun_out/r06j/stats -o p --output-format csv -- python tools/amp_step.py > gpurun_out/r06j/amp_step.log 2>&1
find gpurun_out/r06j/stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06j/r06_kernel_stats_amp_full.csv \;
rm -rf gpurun_out/r06j/stats
head -14 gpurun_out/r06j/r06_kernel_stats_amp_full.csv | cut -c1-200
