# round 6, lease 13: the 16 x 16 x 32 one-product NT kernel (csrc/gemm_bf16_m16.hpp) - lab (kernel level, both chunk sizes),
# the A/B of the AMP-class train step on one box, the tests that touch it
set -x
mkdir -p gpurun_out/r06o
timeout 300 tools/lab_bf16_nt.bin > gpurun_out/r06o/lab.txt 2>&1; echo "rc=$?" >> gpurun_out/r06o/lab.txt
timeout 60 tools/mfma_power_probe.bin bf16 4 > gpurun_out/r06o/probe.txt 2>&1
timeout 60 tools/mfma_power_probe.bin bf16_16 4 >> gpurun_out/r06o/probe.txt 2>&1
export PN_AB_FORWARD=bf16x3 PN_AB_FORWARD_MATH=bf16 PN_AB_BACKWARD=bf16
PN_AB_MFMA16=0 python tools/tn_tasks_ab.py gpurun_out/r06o/ab_mfma32.json > gpurun_out/r06o/ab_mfma32.log 2>&1
PN_AB_MFMA16=1 python tools/tn_tasks_ab.py gpurun_out/r06o/ab_mfma16.json > gpurun_out/r06o/ab_mfma16.log 2>&1
PN_AB_MFMA16=0 python tools/tn_tasks_ab.py gpurun_out/r06o/ab_mfma32_again.json > gpurun_out/r06o/ab_mfma32_again.log 2>&1
timeout 1500 python -m pytest tests/test_hip_fwd_bf16.py tests/test_hip_bwd_bf16.py -q -x -s -p no:cacheprovider > gpurun_out/r06o/tests.txt 2>&1
grep "PRODUCT\|shipped\|must be\|LAB" gpurun_out/r06o/lab.txt
grep -h "ms_per_step\|flat_w_checksum" gpurun_out/r06o/ab_*.json
tail -3 gpurun_out/r06o/tests.txt
