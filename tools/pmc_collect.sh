#!/bin/bash
# Collects per-kernel PMC counters of the default bench (one rocprofv3 pass per counter group; PMC passes
# never combined with sys/hip tracing).  Usage on the GPU box: tools/pmc_collect.sh <outdir> [bench args]
out=${1:-gpurun_out/pmc}; shift
mkdir -p "$out"
export TMPDIR=/tmp
args="--steps 2 --warmup 1 --standalone-steps 2 --no-cpu-baseline --no-mrr --cold-items 0 --batch-sweep= $*"
rocprofv3 -L > "$out/counters_available.txt" 2>&1
declare -A G
G[sq_cycles]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
G[sq_insts]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
G[sq_lds]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"
G[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
G[grbm]="GRBM_GUI_ACTIVE GRBM_COUNT"
for g in "${!G[@]}"; do
  timeout 600 rocprofv3 --kernel-trace --pmc ${G[$g]} -d "$out/$g" -o run -- python bench.py $args > "$out/$g.log" 2>&1
  echo "$g rc=$?" >> "$out/status.txt"
done
# (the caller summarises the databases and deletes them: gpurun_out/ has a 64 MiB limit)
ls -laR "$out" > "$out/listing.txt"
