#!/bin/bash
# round 4 visit 15: ADVICE r3 / VERDICT r3 test items on the GPU + the cost of the MFMA-source fences in every attention instantiation
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for v in product nofence; do
  if [ $v = product ]; then L=""; else L=$R/anyedit_amd/libanyedit_hip_$v.so; fi
  echo "== kbench attn, $v"; AE_LIB_PATH=$L timeout 120 python tools/kbench.py attn 2>&1 | grep -v "amdgpu.ids\|^#"
done | tee $OUT/r04_v15_attn_fence_ab.txt
for v in product nofence product nofence; do
  if [ $v = product ]; then L=""; else L=$R/anyedit_amd/libanyedit_hip_$v.so; fi
  echo -n "$v: "; AE_LIB_PATH=$L python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"
done | tee -a $OUT/r04_v15_attn_fence_ab.txt
( timeout 1200 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py tests/test_hip_unet.py tests/test_hip_sam_anysd.py -m gpu -q -x -p no:cacheprovider -k "attention or ms_deform or bench_plan or launcher or sampler or dpm_solver or edit_pipeline or training_step_gradients" ) > $OUT/r04_v15_pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $OUT/r04_v15_pytest.txt | cut -c1-250
