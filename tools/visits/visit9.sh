#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
echo "== AE_ATTN_V=3 (all P registers of the last K-step held), 1000 evaluations"; AE_ATTN_V=3 DIAG_RUNS=1000 timeout 900 python tools/diag_unet_det.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -6 | tee $OUT/v9_det_v3.txt
