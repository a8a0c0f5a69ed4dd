#!/bin/bash
# First GPU visit of the next round (≈ 3 GPU-minutes): the A/Bs prepared at the end of round 3.
# Here, before the visit (no GPU needed, ≈ 2 min each):
#   python -m anyedit_amd.build --variant cspec gemm_conv.hip=-DAE_CONV_SPEC=1
#   python -m anyedit_amd.build --variant noslp attention_bwd.hip=-fno-slp-vectorize attention.hip=-fno-slp-vectorize
# Then:  gpurun --timeout 400 -- 'bash tools/next_round_first_visit.sh'
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
C=$R/anyedit_amd/libanyedit_hip_cspec.so; N=$R/anyedit_amd/libanyedit_hip_noslp.so
if [ -f $C ]; then
  # 1. same bits?  (40 checksum lines of the product build, recorded in round 3)
  ( AE_LIB_PATH=$C timeout 30 python tools/gemm_conv_checksum.py 2>/dev/null | grep -v amdgpu ) > $OUT/n1_sum_cspec.txt
  if cmp -s $OUT/n1_sum_cspec.txt tools/visits/v50_checksums_reference.txt; then echo "cspec: checksums IDENTICAL"; else echo "cspec: checksums DIFFER"; diff $OUT/n1_sum_cspec.txt tools/visits/v50_checksums_reference.txt | head -6; fi
  # 2. faster?
  bash tools/ab_lib.sh cspec 2 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('=='): print(l, end='  ')
    elif l.startswith('{'):
        d = json.loads(l); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')
" | tee $OUT/n1_cspec_ab.txt
  # 3. the conv / GEMM GPU tests on the variant
  ( AE_LIB_PATH=$C timeout 150 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm or colstats" ) > $OUT/n1_cspec_pytest.txt 2>&1; echo "cspec pytest rc=$?"; tail -2 $OUT/n1_cspec_pytest.txt | cut -c1-160
fi
if [ -f $N ]; then
  bash tools/ab_lib.sh noslp 2 python tools/bench_train.py --steps 10 --warmup 2 | tee $OUT/n1_noslp_train_ab.txt | cut -c1-300
  ( AE_LIB_PATH=$N timeout 120 python -m pytest tests/test_hip_backward.py -m gpu -q -x -p no:cacheprovider -k "attention" ) > $OUT/n1_noslp_pytest.txt 2>&1; echo "noslp pytest rc=$?"; tail -2 $OUT/n1_noslp_pytest.txt | cut -c1-160
fi
