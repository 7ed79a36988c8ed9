# round 6: the bf16 NT GEMM lab (tools/lab_bf16_nt.hip) and the register-only MFMA probe, both built in the build container
set -x
mkdir -p gpurun_out/r06n
if [ -x tools/mfma_power_probe.bin ]; then
  timeout 60 tools/mfma_power_probe.bin bf16 4 > gpurun_out/r06n/probe.txt 2>&1
  timeout 60 tools/mfma_power_probe.bin bf16_16 4 >> gpurun_out/r06n/probe.txt 2>&1
  timeout 60 tools/mfma_power_probe.bin bf16 4 >> gpurun_out/r06n/probe.txt 2>&1
  cat gpurun_out/r06n/probe.txt
fi
timeout 240 tools/lab_bf16_nt.bin $LAB_ARGS > gpurun_out/r06n/lab.txt 2>&1
echo "rc=$?" >> gpurun_out/r06n/lab.txt
cat gpurun_out/r06n/lab.txt
