#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
V=$R/anyedit_amd/libanyedit_hip_wsgpr.so
( AE_LIB_PATH=$V timeout 12 python tools/gemm_conv_checksum.py 2>/dev/null | grep -v amdgpu ) > $OUT/v52_sum_wsgpr.txt
if cmp -s $OUT/v52_sum_wsgpr.txt tools/visits/v50_checksums_reference.txt; then echo "checksums IDENTICAL to the product build's ($(wc -l < $OUT/v52_sum_wsgpr.txt) lines)"; else echo "checksums DIFFER"; diff $OUT/v52_sum_wsgpr.txt tools/visits/v50_checksums_reference.txt | head -6; fi
for i in 1 2; do
  for w in product wsgpr; do
    if [ $w = wsgpr ]; then export AE_LIB_PATH=$V; else unset AE_LIB_PATH; fi
    timeout 12 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],3), 'img/s', round(d['unet_step_ms'],3), 'ms/UNet step')" | tee -a $OUT/v52_bench_ab.txt
  done
done
