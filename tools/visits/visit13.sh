#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( AE_GEMM_ILV=15 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm or colstats" ) > $OUT/v13_pytest_ilv.log 2>&1; echo "pytest ILV=15 rc=$?"; tail -2 $OUT/v13_pytest_ilv.log
kb() { python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L1|L2"; python tools/kbench.py "gemm " 2>&1 | grep -E "proj L2|skip1x1 L2|ff2 L2|qkv L2|ff2 L1|skip1x1 960"; }
for m in 0 15 0 15; do
  echo "== AE_GEMM_ILV=$m (fence 1)"; AE_GEMM_ILV=$m kb
done 2>&1 | grep -v amdgpu.ids | tee $OUT/v13_kbench_ilv.txt
echo "== AE_GEMM_ILV=15 (fence 2: sched_group_barrier)" | tee -a $OUT/v13_kbench_ilv.txt
( AE_LIB_PATH=$R/anyedit_amd/build_abl/libanyedit_hip_abl.so AE_GEMM_ILV=15 kb ) 2>&1 | grep -v amdgpu.ids | tee -a $OUT/v13_kbench_ilv.txt
run() { ( env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v13_tmp.json 2>/dev/null; python -c "import json; d=json.load(open('$OUT/v13_tmp.json')); print('$*', round(d['value'],3), 'img/s', round(d['unet_step_ms_p50'],3), 'ms')"; }
run AE_GEMM_ILV=0
run AE_GEMM_ILV=15
run AE_GEMM_ILV=0
run AE_GEMM_ILV=15
