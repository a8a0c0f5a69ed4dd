#!/bin/bash
# round 4 visit 25: LayerNorm fold with the statistics loads batched (one round trip per block instead of 6-15) and, on the 128x128 tiles, folded
# after the main loop; A/B against AE_LN_FOLD=0 and with the slab conv loop (AE_GEMM_PP=63), three alternating rounds.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out; mkdir -p $OUT
T0=$(date +%s)
( timeout 500 python -m pytest tests/test_hip_ops.py -q -s -x -p no:cacheprovider -k "layernorm_folded" ) > $OUT/v25_ops.log 2>&1; echo "ops rc=$? ($(( $(date +%s) - T0 )) s)"
tail -1 $OUT/v25_ops.log; grep "LN fold" $OUT/v25_ops.log | head -3
run() { echo -n "$1: "; env $1 python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline', {}); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step (p50', round(d.get('unet_step_ms_p50', 0), 3), '); dominant', r.get('kernel'), round(r.get('achieved', 0), 1), round(r.get('frac', 0), 3))"; cp $OUT/kernels_by_shape.json "$OUT/v25_kbs_$(echo $1 | tr ' =' '__').json" 2>/dev/null; }
{ for r in 1 2 3; do
    run "AE_LN_FOLD=0 AE_GEMM_PP=15"; run "AE_LN_FOLD=1 AE_GEMM_PP=15"; run "AE_LN_FOLD=1 AE_GEMM_PP=63"
  done; } | tee $OUT/r04_v25_lnfold_slab_ab.txt
echo "total $(( $(date +%s) - T0 )) s"
