#!/bin/bash
# Round 5, visit 27: AdamW of each adapter layer on a side stream under the rest of the backward pass (train_step, one process): bit-identical parameters after three
# steps (test green), and +0.45 ms per step in three alternating pairs (22.14 -> 22.61; batch 16 +0.3, checkpointed +0.6: profiles/r05_v27_overlap_optimizer_train_ab.txt) —
# like the ResBlock skip GEMM on a side stream (+0.22 ms, v4): a second queue costs this part more than the idle CUs return.  Removed again; this script is the record.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_sam_anysd.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "train or optimizer" ) > $OUT/v27_pytest.log 2>&1; echo "rc=$?"; tail -3 $OUT/v27_pytest.log
for i in 1 2 3; do for v in 0 1; do echo -n "AE_TRAIN_OVERLAP_OPT=$v: "; AE_TRAIN_OVERLAP_OPT=$v timeout 300 python tools/bench_train.py --steps 10 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"; done; done 2>&1 | tee $OUT/v27_overlap_opt.txt
for a in "--batch 16" "--checkpoint"; do for v in 0 1; do echo -n "$a AE_TRAIN_OVERLAP_OPT=$v: "; AE_TRAIN_OVERLAP_OPT=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 2 $a 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"; done; done 2>&1 | tee -a $OUT/v27_overlap_opt.txt
