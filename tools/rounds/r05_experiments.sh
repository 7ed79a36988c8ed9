set -x
mkdir -p gpurun_out/r05d
python -m pytest tests/test_hip_train.py tests/test_hip_config_holes.py -q -m gpu -s -k "sgd_state_interchange or different_math_modes or full_size_one_hidden" 2>&1 | grep -E "full-size|passed|failed|Error|assert" > gpurun_out/r05d/tests.txt
python tools/fp16_weight_probe.py > gpurun_out/r05d/fp16_probe.json 2> gpurun_out/r05d/fp16_probe.err
python -m protnote_amd.build > /dev/null 2>&1
PN_STEPS=6 python tools/tn_tasks_ab.py gpurun_out/r05d/tn_tasks_default.json > /dev/null 2> gpurun_out/r05d/ab_a.err
export PN_EXTRA_HIPCC_FLAGS=-DPN_TN_TASKS_T=1
python -m protnote_amd.build > gpurun_out/r05d/build_b.log 2>&1
PN_STEPS=6 python tools/tn_tasks_ab.py gpurun_out/r05d/tn_tasks_transposed.json > /dev/null 2> gpurun_out/r05d/ab_b.err
unset PN_EXTRA_HIPCC_FLAGS
python -m protnote_amd.build > /dev/null 2>&1
PN_STEPS=6 python tools/tn_tasks_ab.py gpurun_out/r05d/tn_tasks_default_again.json > /dev/null 2> gpurun_out/r05d/ab_c.err
python tools/shape_sweep.py gpurun_out/r05d/shape_sweep.json > gpurun_out/r05d/shape_sweep.txt 2>&1
cat gpurun_out/r05d/tests.txt; tail -3 gpurun_out/r05d/shape_sweep.txt
