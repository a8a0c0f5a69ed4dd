#!/bin/bash
# round 4 visit 32: GELU / GEGLU epilogue arithmetic written for instruction count (gelu = 0.5 (x + |x| erf(|x| / sqrt 2)), GEGLU's 0.5 in a's bias add)
# against the round-1 form (variant build -DAE_GELU_OLD=1): operator + transformer tests, GEGLU agreement of the two forms, A/B of the bench, lab shapes.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=gpurun_out; mkdir -p $OUT
T0=$(date +%s)
( timeout 600 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py -q -x -p no:cacheprovider ) > $OUT/v32_tests.log 2>&1; echo "tests rc=$? ($(( $(date +%s) - T0 )) s)"; tail -1 $OUT/v32_tests.log
( timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v32_cks_product.txt 2>&1
( AE_LIB_PATH=$PWD/anyedit_amd/libanyedit_hip_gelu_old.so timeout 200 python tools/gemm_conv_checksum.py ) > $OUT/v32_cks_gelu_old.txt 2>&1
echo "checksum lines that differ between the two GELU forms (geglu rows expected, in the last digits):"; diff $OUT/v32_cks_product.txt $OUT/v32_cks_gelu_old.txt | head -12
bash tools/ab_lib.sh gelu_old 3 python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import sys, re
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end=': ')
    else:
        m = re.search(r'\"value\": ([0-9.]+).*\"ms_per_step\": ([0-9.]+)', l)
        if m: print(round(float(m.group(1)), 3), 'img/s', round(float(m.group(2)) / 50, 3), 'ms per UNet step')
" | tee $OUT/r04_v32_gelu_ab.txt
tools/ubench/build/pp_plain x | grep geglu | tee -a $OUT/r04_v32_gelu_ab.txt
echo "total $(( $(date +%s) - T0 )) s"
