#!/bin/bash
# Round 5, visit 14: polynomial erf in the GELU / GEGLU epilogues (AE_GELU_POLY) against the 7.1.26 series (variant library gelu_as): tests, bench A/B, lab shapes.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py -m gpu -q -x -s -p no:cacheprovider -k "gelu or geglu or layernorm_folded or transformer or unet or gemm" ) > $OUT/v14_pytest.log 2>&1; echo "rc=$?"
grep -E "passed|failed|LN fold M=49152 C=320 N=2560|LN fold M=12288 C=640 N=5120|Error" $OUT/v14_pytest.log | tail -6
bash tools/ab_lib.sh gelu_as 3 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import sys, re
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end=': ')
    else:
        m = re.search(r'\"value\": ([0-9.]+).*\"ms_per_step\": ([0-9.]+)', l)
        if m: print(round(float(m.group(1)), 3), 'img/s', round(float(m.group(2)) / 50, 3), 'ms per UNet step')
" | tee $OUT/v14_gelu_poly_ab.txt
tools/ubench/build/pp_plain g | grep geglu | tee -a $OUT/v14_gelu_poly_ab.txt
