# round 6, third lease: the staged (materialised-operand) AMP forward, the fixed tests, a short bench, then the TN tail A/B
set -x
mkdir -p gpurun_out/r06c
timeout 2400 python -m pytest tests/test_hip_fwd_bf16.py tests/test_hip_inputs.py tests/test_hip_fuzz_configs.py tests/test_hip_depth.py \
  "tests/test_hip_train.py::test_extra_losses_golden" "tests/test_hip_train.py::test_full_size_train_step_vs_chunked_torch" \
  tests/test_hip_parity.py tests/test_hip_config_holes.py -q -s --durations=10 -p no:cacheprovider > gpurun_out/r06c/tests.txt 2>&1
grep -E "^\[|forward bf16|depth [0-9]|mAP parity|200 steps|full-size train step|staged vs|passed|failed|FAILED|ERROR" gpurun_out/r06c/tests.txt | cut -c1-400 > gpurun_out/r06c/tests_lines.txt
timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06c/bench.out 2> gpurun_out/r06c/bench.err
cp bench_detail.json gpurun_out/r06c/
# TN tail A/B: f32 forward + f32 backward, default build vs -DPN_TN_TAIL_RT=1
PN_AB_FORWARD=f32 PN_AB_BACKWARD=same PN_STEPS=3 timeout 600 python tools/tn_tasks_ab.py gpurun_out/r06c/tn_tail_a.json > gpurun_out/r06c/tn_tail_a.log 2>&1
export PN_EXTRA_HIPCC_FLAGS=-DPN_TN_TAIL_RT=1
timeout 600 python -m protnote_amd.build > gpurun_out/r06c/build_b.log 2>&1
PN_AB_FORWARD=f32 PN_AB_BACKWARD=same PN_STEPS=3 timeout 600 python tools/tn_tasks_ab.py gpurun_out/r06c/tn_tail_b.json > gpurun_out/r06c/tn_tail_b.log 2>&1
unset PN_EXTRA_HIPCC_FLAGS
tail -3 gpurun_out/r06c/tests.txt
tail -1 gpurun_out/r06c/bench.out | cut -c1-1500
