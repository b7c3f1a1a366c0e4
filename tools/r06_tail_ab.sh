#!/bin/bash
# round 6: (a) the headline step's tail — NS segments in flight per lane group of the sparse update and fewer update workgroups beside the
# dense-gradient GEMM; (b) more grid caps of the EWMA whole-sequence form
mkdir -p gpurun_out/r06
cd /root/repo
L=$PWD/sbr_rs_amd
out=gpurun_out/r06/tail_ab.jsonl
export STEPS=20
for rep in 1 2; do
  EXTRA="--traffic off" bash tools/step_ab.sh $out "8192" "SBR_HIP_LIB=$L/libsbr_hip_c512.so" "SBR_HIP_LIB=$L/libsbr_hip_ns2.so" "SBR_HIP_LIB=$L/libsbr_hip_ns2c512.so" "SBR_HIP_LIB=$L/libsbr_hip_ns3c512.so" "SBR_HIP_LIB=$L/libsbr_hip_ns3c256.so" "SBR_HIP_LIB=$L/libsbr_hip_ns4c256.so" | cut -c1-330
done
EXTRA="--traffic off" bash tools/step_ab.sh $out "50000" "SBR_HIP_LIB=$L/libsbr_hip_ns2.so" "SBR_HIP_LIB=$L/libsbr_hip_ns3c512.so" | cut -c1-330
out=gpurun_out/r06/ewma_whole_ab2.jsonl
export STEPS=10
for rep in 1 2; do
  EXTRA="--model ewma --loss hinge --dim 256 --items 10000000 --traffic off" bash tools/step_ab.sh $out "50000" "SBR_HIP_LIB=$L/libsbr_hip_whole384.so" "SBR_HIP_LIB=$L/libsbr_hip_whole512.so" "SBR_HIP_LIB=$L/libsbr_hip_whole640.so" "SBR_HIP_LIB=$L/libsbr_hip_whole768.so" | cut -c1-330
done
EXTRA="--model ewma --loss hinge --dim 128 --items 1000000 --traffic off" bash tools/step_ab.sh $out "50000" "SBR_HIP_LIB=$L/libsbr_hip_whole512.so" "SBR_HIP_LIB=$L/libsbr_hip_whole.so"| cut -c1-330
EXTRA="--model ewma --loss hinge --dim 128 --items 1000000 --traffic off" bash tools/step_ab.sh $out "8192" "SBR_HIP_LIB=$L/libsbr_hip_whole512.so" "SBR_HIP_LIB=$L/libsbr_hip_whole.so"| cut -c1-330
