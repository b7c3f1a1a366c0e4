#!/bin/bash
# The GPU parity suite three times: with the engine's own choice of kernel forms, with the MFMA tile forms forced everywhere
# (the forms that small test shapes no longer reach by default), and with the wave / block forms forced wherever the shape
# allows.  All three must give the same 0 failures: the forms are bit-identical.  (On an MI355X; ~3 minutes.)
cd "$(dirname "$0")/.." || exit 1
set -e
echo "== engine's choice";   python -m pytest tests -m gpu -x -q | tail -2
echo "== tile forms forced"; SBR_WAVE=0 SBR_DW_BLOCK=0 SBR_SMALL_STEP_ROWS=0 SBR_SORT_ITEMS=64 SBR_SCORE_U=1 SBR_EWMA_FUSED=0 SBR_HEADER_ON_MAIN=1 SBR_NO_HOT_PRELIST=1 SBR_NO_SMALL_TAIL=1 SBR_NO_SMALL_BACK=1 python -m pytest tests -m gpu -x -q | tail -2
echo "== wave forms forced"; SBR_WAVE=1 SBR_DW_BLOCK=1 SBR_SCORE_U=2 SBR_EWMA_FUSED=2 python -m pytest tests -m gpu -x -q | tail -2
