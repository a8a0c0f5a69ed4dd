#!/bin/bash
# round 4 visit 30: wave-local synchronisation inside the staged GEMM / conv epilogue (product) against the block barriers (variant build
# -DAE_EPI_BLOCK_SYNC=1): operator tests, bit-identity (checksums of both libraries), alternating A/B of the bench.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out; mkdir -p $OUT
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_hip_ops.py -q -x -p no:cacheprovider ) > $OUT/v30_ops.log 2>&1; echo "ops rc=$? ($(( $(date +%s) - T0 )) s)"; tail -1 $OUT/v30_ops.log
( timeout 300 python -m pytest tests/test_hip_bench_shapes.py -q -x -p no:cacheprovider -k "conv3x3 or rowpanel" ) > $OUT/v30_shapes.log 2>&1; echo "shapes rc=$?"; tail -1 $OUT/v30_shapes.log
( timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v30_cks_product.txt 2>&1
( AE_LIB_PATH=$PWD/anyedit_amd/libanyedit_hip_blocksync.so timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v30_cks_blocksync.txt 2>&1
if diff -q $OUT/v30_cks_product.txt $OUT/v30_cks_blocksync.txt > /dev/null; then echo "checksums: wave-local sync == block barriers ($(grep -c . $OUT/v30_cks_product.txt) lines)"; else echo "CHECKSUMS DIFFER"; diff $OUT/v30_cks_product.txt $OUT/v30_cks_blocksync.txt | head; fi
diff -q $OUT/v30_cks_product.txt $OUT/v24_cks_pp63.txt > /dev/null 2>&1 && echo "and == visit 24's"
bash tools/ab_lib.sh blocksync 3 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import sys, re
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end=': ')
    else:
        m = re.search(r'\"value\": ([0-9.]+).*\"ms_per_step\": ([0-9.]+)', l)
        if m: print(round(float(m.group(1)), 3), 'img/s', round(float(m.group(2)) / 50, 3), 'ms per UNet step')
" | tee $OUT/r04_v30_epi_sync_ab.txt
echo "total $(( $(date +%s) - T0 )) s"
