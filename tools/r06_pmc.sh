#!/bin/bash
# round 6: PMC counters of the step's kernels — the headline step (for the record of the shipped build) and the one-pass EWMA step
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
bash tools/pmc_collect.sh /tmp/pmc_lstm --prewarm-seconds 0 > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc_lstm lstm_fwd_seq lstm_bwd_seq lstm_dw score_kernel seg_short > gpurun_out/r06/pmc_kernels_b8192.md
cat /tmp/pmc_lstm/status.txt
bash tools/pmc_collect.sh /tmp/pmc_ewma --prewarm-seconds 0 --model ewma --loss hinge --dim 256 --items 10000000 > /dev/null 2>&1
python tools/pmc_summary.py /tmp/pmc_ewma ewma_seq_kernel seg_short > gpurun_out/r06/pmc_ewma256_kernels.md
cat /tmp/pmc_ewma/status.txt
head -40 gpurun_out/r06/pmc_ewma256_kernels.md
