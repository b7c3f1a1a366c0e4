#!/bin/bash
# Another build of the engine for kernel A/B experiments: tools/build_variant.sh NAME "-DFLAG=..." -> sbr_rs_amd/libsbr_hip_NAME.so
# (select it with SBR_HIP_LIB=$PWD/sbr_rs_amd/libsbr_hip_NAME.so; git-ignored like every built artefact).
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/build/variant_$name; mkdir -p "$out"
for f in sbr_kernels sbr_steps sbr_sort sbr_wave sbr_report sbr_engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -Wno-unused-result \
    -Wno-unused-value $flags -c "$root/sbr_rs_amd/csrc/$f.hip" -o "$out/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o "$root/sbr_rs_amd/libsbr_hip_$name.so" "$out"/*.o -ldl
echo "$root/sbr_rs_amd/libsbr_hip_$name.so"
