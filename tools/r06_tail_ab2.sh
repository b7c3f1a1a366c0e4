#!/bin/bash
# round 6: the dense-gradient GEMM at 96 registers (four of its waves + one 80-register update wave fit a SIMD's 512), with and without an
# LDS pad that keeps a fifth GEMM workgroup off the CU, and the update's grid capped at 4 / 2 / 1 workgroups per CU
mkdir -p gpurun_out/r06
cd /root/repo
L=$PWD/sbr_rs_amd
out=gpurun_out/r06/tail_ab2.jsonl
export STEPS=20
for rep in 1 2; do
  EXTRA="--traffic off" bash tools/step_ab.sh $out "8192" "SBR_HIP_LIB=$L/libsbr_hip_wpe5.so" "SBR_HIP_LIB=$L/libsbr_hip_wpe5pad.so" "SBR_HIP_LIB=$L/libsbr_hip_wpe5padc512.so" "SBR_HIP_LIB=$L/libsbr_hip_wpe5padc256.so" | cut -c1-330
done
EXTRA="--traffic off" bash tools/step_ab.sh $out "50000" "SBR_HIP_LIB=$L/libsbr_hip_wpe5.so" "SBR_HIP_LIB=$L/libsbr_hip_wpe5pad.so" | cut -c1-330
