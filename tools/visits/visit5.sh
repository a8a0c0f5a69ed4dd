#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for d in 0 1 2 3 4 5; do echo "== AE_ATTN_V=3 AE_ATTN_DBG=$d"; DIAG_RUNS=60 DIAG_ONLY_DET=1 AE_ATTN_V=3 AE_ATTN_DBG=$d timeout 300 python tools/diag_attn.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -6 | tee $OUT/v5_diag_dbg$d.txt; done
