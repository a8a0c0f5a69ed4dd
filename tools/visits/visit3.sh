#!/bin/bash
# Round-3 visit 3: hybrid loader A/B (AE_GEMM_HYB) on the conv / dense tiles; full suite with the new defaults (producer statistics,
# attention V=3); bench.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( AE_GEMM_HYB=63 AE_GN_COLSTATS=0 timeout 600 python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm" ) > $OUT/v3_pytest_hyb.log 2>&1; echo "pytest HYB=63 rc=$?"; tail -2 $OUT/v3_pytest_hyb.log
( AE_GEMM_HYB=42 AE_GN_COLSTATS=0 timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv3x3" ) > $OUT/v3_pytest_hyb42.log 2>&1; echo "pytest HYB=42 rc=$?"; tail -2 $OUT/v3_pytest_hyb42.log
for h in 0 1 2 4 16; do
  echo "== kbench conv AE_GEMM_HYB=$h"; AE_GEMM_HYB=$h python tools/kbench.py "conv3x3 res" 2>&1 | grep -E "L1|L2" | tee $OUT/v3_kbench_conv_hyb$h.txt
done
for h in 0 8 32; do
  echo "== kbench dense AE_GEMM_HYB=$h"; AE_GEMM_HYB=$h python tools/kbench.py "gemm " 2>&1 | grep -E "proj L2|skip1x1 L2|ff2 L2|qkv L2|qkv L3|proj L3" | tee $OUT/v3_kbench_gemm_hyb$h.txt
done
( timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_hip_fullsize.py::test_masked_edit_5_steps_cfg_full_size_96 ) > $OUT/v3_pytest_full.log 2>&1; echo "pytest full rc=$?"; tail -3 $OUT/v3_pytest_full.log
for cfg in "0" "1" "5" "21"; do
  ( AE_GEMM_HYB=$cfg AE_GN_COLSTATS=0 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v3_bench_hyb$cfg.json 2> $OUT/v3_bench_hyb$cfg.err
  echo "bench HYB=$cfg (COLSTATS=0): $(python -c "import json,sys; d=json.load(open('$OUT/v3_bench_hyb$cfg.json')); print(round(d['value'],3),'img/s  unet p50', round(d['unet_step_ms_p50'],3),'ms')" 2>&1)"
done
( timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v3_bench_default.json 2> $OUT/v3_bench_default.err
echo "bench default: $(python -c "import json,sys; d=json.load(open('$OUT/v3_bench_default.json')); print(round(d['value'],3),'img/s  unet p50', round(d['unet_step_ms_p50'],3),'ms')" 2>&1)"
