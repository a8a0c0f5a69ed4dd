#!/bin/bash
# Round 5, visit 18: gradient accumulation fused into the backward kernels (LayerNorm / GroupNorm dx +=, GEMM residual in place): tests, training A/B.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 1200 python -m pytest tests/test_hip_backward.py tests/test_hip_sam_anysd.py tests/test_hip_fullsize.py -m gpu -q -x -p no:cacheprovider -k "backward or train or adds_into or alias or grad" ) > $OUT/v18_pytest.log 2>&1; echo "rc=$?"; tail -5 $OUT/v18_pytest.log
for i in 1 2 3; do
  for v in 0 1; do
    echo "== AE_TAPE_FUSE_ADD=$v"; AE_TAPE_FUSE_ADD=$v timeout 300 python tools/bench_train.py --steps 8 --warmup 2 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 3), 'ms per step, loss', d['loss'])"
  done
done 2>&1 | tee $OUT/v18_train_ab.txt
