#!/bin/bash
# Round 5, visit 28: attention backward d = 40 with the first product of tile t + 1 under the P block of tile t (PIPE variant, AE_ATTN_BWD_PIPE): timing first.
# Result: 266.3 -> 269.5 us (dQ), 289.3 -> 283.8 us (dK/dV): nothing; the variant (an `if constexpr (PIPE)` main loop in attn_bwd_kernel, three LDS stages) was removed again.
# This script and profiles/r05_v28_attn_bwd_pipe.txt are the record.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for i in 1 2 3; do for v in 0 1; do echo -n "AE_ATTN_BWD_PIPE=$v: "; AE_ATTN_BWD_PIPE=$v python tools/attn_bwd_lab.py 32 4096 40 20 2>/dev/null | grep "attention backward"; done; done 2>&1 | tee $OUT/v28_bwd_pipe.txt
( AE_ATTN_BWD_PIPE=1 timeout 600 python -m pytest tests/test_hip_backward.py -m gpu -q -x -p no:cacheprovider -k "attention" ) > $OUT/v28_pytest.log 2>&1; echo "rc=$?"; tail -3 $OUT/v28_pytest.log
