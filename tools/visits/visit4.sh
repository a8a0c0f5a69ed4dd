#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for v in 3 1 0; do echo "== AE_ATTN_V=$v"; AE_ATTN_V=$v timeout 300 python tools/diag_attn.py 2>&1 | grep -v Warn | tee $OUT/v4_diag_attn_v$v.txt; done
echo "== kbench layernorm rows=1/0"; python tools/kbench.py "layernorm" 2>&1 | grep -v "^#"; AE_LN_ROWS=0 python tools/kbench.py "layernorm" 2>&1 | grep -v "^#"
( timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "layernorm or norm" ) 2>&1 | tail -2
