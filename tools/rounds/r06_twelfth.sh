# round 6, twelfth lease: what the vendor library reaches on the bf16 GEMM shapes of the AMP-class modes (a yardstick for
# gemm_nt_bf16dma_kernel / gemm_tn_bf16tr_kernel, tools/bf16_gemm_yardstick.py), default BLAS backend and hipBLASLt preferred,
# plus the kernel names of one run (rocprofv3 --kernel-trace --stats)
set -x
mkdir -p gpurun_out/r06m
python tools/bf16_gemm_yardstick.py gpurun_out/r06m/yardstick_default.json > gpurun_out/r06m/yardstick_default.log 2>&1
TORCH_BLAS_PREFER_HIPBLASLT=1 python tools/bf16_gemm_yardstick.py gpurun_out/r06m/yardstick_hipblaslt.json > gpurun_out/r06m/yardstick_hipblaslt.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ys -o ys --output-format csv -- python $GRAFT_REPO_ROOT/tools/bf16_gemm_yardstick.py > /tmp/ys.log 2>&1
cd $GRAFT_REPO_ROOT
find /tmp/ys -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06m/yardstick_kernel_stats.csv \;
cat gpurun_out/r06m/yardstick_default.json gpurun_out/r06m/yardstick_hipblaslt.json | grep -v "min_ms\|max_ms"
head -8 gpurun_out/r06m/yardstick_kernel_stats.csv | cut -c1-300
