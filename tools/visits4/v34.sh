#!/bin/bash
# round 4 visit 34: packed fp32 VALU (v_pk_fma_f32) in the GEGLU staging loop of the 192x320 tile (product) against one value per instruction (variant -DAE_GEGLU_PK=0)
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out; mkdir -p $OUT
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_hip_ops.py -q -x -p no:cacheprovider ) > $OUT/v34_tests.log 2>&1; echo "tests rc=$? ($(( $(date +%s) - T0 )) s)"; tail -1 $OUT/v34_tests.log
( timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v34_cks_product.txt 2>&1
( AE_LIB_PATH=$PWD/anyedit_amd/libanyedit_hip_nopk.so timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v34_cks_nopk.txt 2>&1
echo "checksum lines that differ between packed and plain GEGLU staging (none expected: the same operations per value):"; diff $OUT/v34_cks_product.txt $OUT/v34_cks_nopk.txt | head -12
bash tools/ab_lib.sh nopk 3 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import sys, re
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end=': ')
    else:
        m = re.search(r'\"value\": ([0-9.]+).*\"ms_per_step\": ([0-9.]+)', l)
        if m: print(round(float(m.group(1)), 3), 'img/s', round(float(m.group(2)) / 50, 3), 'ms per UNet step')
" | tee $OUT/r04_v34_geglu_pk_ab.txt
{ echo "lab, packed:"; tools/ubench/build/pp_plain x | grep geglu; echo "lab, one value per instruction:"; tools/ubench/build/pp_nopk x | grep geglu; } | tee -a $OUT/r04_v34_geglu_pk_ab.txt
echo "total $(( $(date +%s) - T0 )) s"
