# round 6, second lease: the whole GPU suite on the sources with forward_math, the stand-alone conv / residual entry points,
# typed targets, residue ids, in-kernel label noise, deep heads
set -x
mkdir -p gpurun_out/r06b
timeout 3000 python -m pytest tests -q -m gpu -s --durations=15 -p no:cacheprovider > gpurun_out/r06b/gpu_tests_full.txt 2>&1
grep -E "^\[|forward bf16|depth [0-9]|mAP parity|200 steps|full-size train step|passed|failed|FAILED|ERROR" gpurun_out/r06b/gpu_tests_full.txt | cut -c1-420 > gpurun_out/r06b/gpu_tests_lines.txt
tail -40 gpurun_out/r06b/gpu_tests_full.txt | cut -c1-300
