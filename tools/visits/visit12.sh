#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
export AE_LIB_PATH=$R/anyedit_amd/build_abl/libanyedit_hip_abl.so
( AE_GEMM_ABL=20 timeout 600 python -m pytest tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv3x3_bench_plan" ) > $OUT/v12_pytest.log 2>&1; echo "pytest ABL=20 rc=$?"; tail -2 $OUT/v12_pytest.log
for m in 0 20 0 20; do
  echo "== AE_GEMM_ABL=$m"
  AE_GEMM_ABL=$m python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L1"
done 2>&1 | grep -v amdgpu.ids | tee $OUT/v12_kbench_regstage.txt
