#!/bin/bash
# Round-3 visit 2: producer-side GroupNorm statistics (AE_GN_COLSTATS) and the attention variants (AE_ATTN_V) — parity, then A/B.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py tests/test_hip_fullsize.py -m gpu -q -x -p no:cacheprovider -k "not masked_edit" ) > $OUT/v2_pytest_a.log 2>&1; echo "pytest A rc=$?"; tail -4 $OUT/v2_pytest_a.log
for v in 1 2 3; do
  ( AE_ATTN_V=$v timeout 300 python -m pytest tests/test_hip_ops.py -m gpu -q -p no:cacheprovider -k "attention or attn" ) > $OUT/v2_pytest_attn_v$v.log 2>&1; echo "pytest attn V=$v rc=$?"; tail -2 $OUT/v2_pytest_attn_v$v.log
done
( AE_ATTN_V=3 timeout 600 python -m pytest tests/test_hip_fullsize.py -m gpu -q -s -p no:cacheprovider -k "unet_full_size" ) > $OUT/v2_pytest_full_v3.log 2>&1; echo "pytest fullsize V=3 rc=$?"; grep -E "UNet|passed|failed" $OUT/v2_pytest_full_v3.log | tail -4
for v in 0 1 2 3; do
  echo "== kbench attention AE_ATTN_V=$v"; AE_ATTN_V=$v python tools/kbench.py "attn self" 2>&1 | grep -v "^#" | tee -a $OUT/v2_kbench_attn_v$v.txt
done
echo "== kbench dense T320=11 vs default"; python tools/kbench.py "gemm " 2>&1 | grep -E "ff2 L1|skip1x1 960|proj L1|ff2 L2|qkv L3|ff2 L3|skip1x1 L" | tee $OUT/v2_kbench_gemm_default.txt
AE_GEMM_T320=11 python tools/kbench.py "gemm " 2>&1 | grep -E "ff2 L1|skip1x1 960|proj L1|ff2 L2|qkv L3|ff2 L3|skip1x1 L" | tee $OUT/v2_kbench_gemm_t320_11.txt
echo "== kbench groupnorm / conv"; python tools/kbench.py "groupnorm" 2>&1 | grep -v "^#"; python tools/kbench.py "conv3x3 res" 2>&1 | grep -v "^#"
for cfg in "0 0" "1 0" "0 3" "1 3" "1 2" "1 1"; do
  set -- $cfg
  ( AE_GN_COLSTATS=$1 AE_ATTN_V=$2 timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline ) > $OUT/v2_bench_cs$1_v$2.json 2> $OUT/v2_bench_cs$1_v$2.err
  echo "bench COLSTATS=$1 ATTN_V=$2: $(python -c "import json,sys; d=json.load(open('$OUT/v2_bench_cs$1_v$2.json')); print(round(d['value'],3),'img/s  unet p50', round(d['unet_step_ms_p50'],3),'ms')" 2>&1)"
  cp $OUT/kernels_by_shape.json $OUT/v2_kernels_cs$1_v$2.json 2>/dev/null
done
cd /tmp && rm -rf /tmp/kt && ( timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 12 --no-cpu-baseline --no-roofline ) > $OUT/v2_ktrace.log 2>&1; echo "ktrace rc=$?"; cd $R
python tools/trace_gaps.py /tmp/kt $OUT/v2_trace_gaps.json
