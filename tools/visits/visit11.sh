#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( AE_GEMM_MIDB=15 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm or colstats" ) > $OUT/v11_pytest_midb.log 2>&1; echo "pytest MIDB=15 rc=$?"; tail -2 $OUT/v11_pytest_midb.log
for m in 0 15 0 15; do
  echo "== AE_GEMM_MIDB=$m"
  AE_GEMM_MIDB=$m python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L1|L2"
  AE_GEMM_MIDB=$m python tools/kbench.py "gemm " 2>&1 | grep -E "proj L2|skip1x1 L2|ff2 L2|qkv L2|ff2 L1|skip1x1 960"
done 2>&1 | grep -v amdgpu.ids | tee $OUT/v11_kbench_midb.txt
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v11_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/v11_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')"; }
run AE_GEMM_MIDB=0
run AE_GEMM_MIDB=15
run AE_GEMM_MIDB=1
run AE_GEMM_MIDB=3
run AE_GEMM_MIDB=0
run AE_GEMM_MIDB=15
