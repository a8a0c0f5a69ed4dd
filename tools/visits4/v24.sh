#!/bin/bash
# round 4 visit 24: (1) LayerNorm folded into the consuming GEMM (AE_LN_FOLD, default on) — operator parity, UNet at the bench batch, A/B;
# (2) the slab form of the ping-pong conv loop (AE_GEMM_PP flags 16 / 32, default off) — conv parity, bit-identity against the product loop
# (checksums), A/B in the bench and per shape.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out; mkdir -p $OUT
T0=$(date +%s)
( timeout 500 python -m pytest tests/test_hip_ops.py -q -s -x -p no:cacheprovider -k "layernorm_folded or conv3x3 or gemm_plain or rowpanel" ) > $OUT/v24_ops.log 2>&1; echo "ops rc=$? ($(( $(date +%s) - T0 )) s)"
tail -2 $OUT/v24_ops.log; grep "LN fold" $OUT/v24_ops.log
( AE_GEMM_PP=63 timeout 500 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -q -s -x -p no:cacheprovider -k "conv3x3" ) > $OUT/v24_slab_tests.log 2>&1; echo "slab conv tests rc=$? ($(( $(date +%s) - T0 )) s)"
tail -2 $OUT/v24_slab_tests.log
( AE_GEMM_PP=15 timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v24_cks_pp15.txt 2>&1
( AE_GEMM_PP=63 timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v24_cks_pp63.txt 2>&1
if diff -q $OUT/v24_cks_pp15.txt $OUT/v24_cks_pp63.txt > /dev/null; then echo "checksums: slab loop == product loop ($(grep -c . $OUT/v24_cks_pp63.txt) lines)"; else echo "CHECKSUMS DIFFER"; diff $OUT/v24_cks_pp15.txt $OUT/v24_cks_pp63.txt | head -20; fi
( AE_GEMM_PP=63 timeout 900 python -m pytest tests/test_hip_bench_shapes.py -q -s -x -p no:cacheprovider -k "test_unet_bench_batch_vs_oracle_with_bf16_control and 12" ) > $OUT/v24_unet12.log 2>&1; echo "unet batch 12 (fold + slab) rc=$? ($(( $(date +%s) - T0 )) s)"
grep -E "rel-L2|passed|failed|Error" $OUT/v24_unet12.log | tail -6
run() { echo -n "$1: "; env $1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline', {}); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step; dominant', r.get('kernel'), round(r.get('achieved', 0), 1), round(r.get('frac', 0), 3))"; cp $OUT/kernels_by_shape.json "$OUT/v24_kbs_$(echo $1 | tr ' =' '__').json" 2>/dev/null; }
{ for r in 1 2; do
    run "AE_LN_FOLD=0 AE_GEMM_PP=15"; run "AE_LN_FOLD=1 AE_GEMM_PP=15"; run "AE_LN_FOLD=1 AE_GEMM_PP=31"; run "AE_LN_FOLD=1 AE_GEMM_PP=63"
  done; } | tee $OUT/r04_v24_lnfold_slab_ab.txt
echo "total $(( $(date +%s) - T0 )) s"
