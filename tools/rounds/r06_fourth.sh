# round 6, fourth lease: re-check the reworked tests, counter passes over the AMP-class forward
set -x
mkdir -p gpurun_out/r06d
timeout 1500 python -m pytest tests/test_hip_fwd_bf16.py tests/test_hip_fuzz_configs.py tests/test_hip_config_holes.py tests/test_hip_forward_modes.py -q -s --durations=6 -p no:cacheprovider > gpurun_out/r06d/tests.txt 2>&1
grep -E "^\[|passed|failed|FAILED|ERROR" gpurun_out/r06d/tests.txt | cut -c1-500 > gpurun_out/r06d/tests_lines.txt
timeout 1200 bash tools/pmc_amp_fwd.sh > gpurun_out/r06d/pmc_amp_fwd.log 2>&1
cp gpurun_out/pmc_amp_fwd/summary.json gpurun_out/r06d/pmc_amp_fwd_summary.json
rm -rf gpurun_out/pmc_amp_fwd/sq gpurun_out/pmc_amp_fwd/mem gpurun_out/pmc_amp_fwd/fetch gpurun_out/pmc_amp_fwd/write
tail -3 gpurun_out/r06d/tests.txt
