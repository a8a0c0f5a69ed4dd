#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
P=$R/anyedit_amd/libanyedit_hip_prev.so
( AE_LIB_PATH=$P timeout 25 python tools/gemm_conv_checksum.py ) > $OUT/v50_sum_prev.txt 2>&1
( timeout 25 python tools/gemm_conv_checksum.py ) > $OUT/v50_sum_new.txt 2>&1
if cmp -s <(grep -v amdgpu $OUT/v50_sum_prev.txt) <(grep -v amdgpu $OUT/v50_sum_new.txt); then echo "checksums IDENTICAL ($(grep -vc amdgpu $OUT/v50_sum_new.txt) lines)"; else echo "checksums DIFFER"; diff <(grep -v amdgpu $OUT/v50_sum_prev.txt) <(grep -v amdgpu $OUT/v50_sum_new.txt) | head -12; fi
tail -2 $OUT/v50_sum_new.txt | cut -c1-150
for i in 1 2; do
  for w in prev new; do
    if [ $w = prev ]; then export AE_LIB_PATH=$P; else unset AE_LIB_PATH; fi
    timeout 40 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', round(d['value'],3), 'img/s', round(d['unet_step_ms'],3), 'ms/UNet step')" | tee -a $OUT/v50_bench_ab.txt
  done
done
