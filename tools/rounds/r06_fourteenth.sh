# round 6, lease 14: the weight-gradient kernel (gemm_tn_bf16tr_kernel) in its 16 x 16 x 32 form: tests, A/B of the AMP-class step
set -x
mkdir -p gpurun_out/r06p
timeout 1500 python -m pytest tests/test_hip_bwd_bf16.py tests/test_hip_fwd_bf16.py::test_mfma16_matches_mfma32 -q -x -s -p no:cacheprovider > gpurun_out/r06p/tests.txt 2>&1
export PN_AB_FORWARD=bf16x3 PN_AB_FORWARD_MATH=bf16 PN_AB_BACKWARD=bf16
PN_AB_MFMA16=0 python tools/tn_tasks_ab.py gpurun_out/r06p/ab_mfma32.json > gpurun_out/r06p/ab_mfma32.log 2>&1
PN_AB_MFMA16=1 python tools/tn_tasks_ab.py gpurun_out/r06p/ab_mfma16.json > gpurun_out/r06p/ab_mfma16.log 2>&1
PN_AB_MFMA16=0 python tools/tn_tasks_ab.py gpurun_out/r06p/ab_mfma32_again.json > gpurun_out/r06p/ab_mfma32_again.log 2>&1
PN_AB_MFMA16=1 python tools/tn_tasks_ab.py gpurun_out/r06p/ab_mfma16_again.json > gpurun_out/r06p/ab_mfma16_again.log 2>&1
grep -h "ms_per_step\|flat_w_checksum\|tn:plain" gpurun_out/r06p/ab_*.json
tail -5 gpurun_out/r06p/tests.txt
