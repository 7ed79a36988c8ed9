# round 6, first lease: the AMP-class forward (forward_math = bf16) on the single-product instantiation of the bf16x3 kernel
set -x
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests/test_hip_fwd_bf16.py -x -q -s --durations=8 2>&1 | tail -60 > gpurun_out/r06a/fwd_bf16_tests.txt
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06a/bench.out 2> gpurun_out/r06a/bench.err
cp bench_detail.json gpurun_out/r06a/
tail -1 gpurun_out/r06a/bench.out | cut -c1-3000
tail -30 gpurun_out/r06a/fwd_bf16_tests.txt
