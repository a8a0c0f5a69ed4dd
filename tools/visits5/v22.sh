#!/bin/bash
# Round 5, visit 22: attention backward d = 40 with 64 fixed rows per block (QF = 1, three blocks per CU) against 128 (QF = 2, two per CU).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for i in 1 2 3; do for v in 0 1; do echo -n "AE_ATTN_BWD_QF1=$v: "; AE_ATTN_BWD_QF1=$v python tools/attn_bwd_lab.py 32 4096 40 20; done; done 2>&1 | tee $OUT/v22_qf1.txt
( AE_ATTN_BWD_QF1=1 timeout 600 python -m pytest tests/test_hip_backward.py -m gpu -q -x -p no:cacheprovider -k "attention or fuzz" ) > $OUT/v22_pytest.log 2>&1; echo "rc=$?"; tail -3 $OUT/v22_pytest.log
for i in 1 2; do for v in 0 1; do echo -n "AE_ATTN_BWD_QF1=$v: "; AE_ATTN_BWD_QF1=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step')"; done; done 2>&1 | tee -a $OUT/v22_qf1.txt
