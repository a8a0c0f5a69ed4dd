#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for v in 3 1 0; do echo "== AE_ATTN_V=$v"; DIAG_RUNS=100 DIAG_ONLY_DET=1 AE_ATTN_V=$v timeout 300 python tools/diag_attn.py 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -4 | tee $OUT/v6_diag_v$v.txt; done
for v in 0 1 3; do echo "== kbench attention AE_ATTN_V=$v"; AE_ATTN_V=$v python tools/kbench.py "attn self N=4096" 2>&1 | grep -v "^#\|amdgpu.ids"; AE_ATTN_V=$v python tools/kbench.py "attn self N=4096" 2>&1 | grep -v "^#\|amdgpu.ids"; done
( timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "attention or attn" ) 2>&1 | tail -2
