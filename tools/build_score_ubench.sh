#!/bin/bash
# builds tools/bin/score_ubench (the library's own score kernels behind a stand-alone timing harness)
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -Wno-unused-value -Wno-unused-function -Wno-unused-variable"
/opt/rocm/bin/hipcc $F $EXTRA_FLAGS -c tools/score_ubench.hip -o /tmp/score_ubench.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/score_ubench.o sbr_rs_amd/csrc/sbr_wave.o sbr_rs_amd/csrc/sbr_report.o sbr_rs_amd/csrc/sbr_sort.o sbr_rs_amd/csrc/sbr_steps.o -o tools/bin/score_ubench${SUFFIX} -ldl -pthread
