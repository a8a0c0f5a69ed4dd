#!/bin/bash
# Round 5, visit 2: the software-pipelined self-attention kernel (AE_ATTN_V flag 4) — bit-identity against the two-query-group kernel, time,
# PMC, the UNet step with it; the pooled router-gradient test again (accumulation fixed).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
echo "== attn_pipe_check"
( timeout 600 python tools/attn_pipe_check.py 3 7 ) 2>&1 | grep -v amdgpu.ids | tee $OUT/v2_attn_pipe_check.txt
echo "== pooled router test"
( timeout 600 python -m pytest tests/test_hip_sam_anysd.py -m gpu -q -s -x -p no:cacheprovider -k training_step_gradients ) > $OUT/v2_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|pooled|Error" $OUT/v2_pytest.log | tail -8
echo "== pmc pipe"
AE_ATTN_V=7 bash tools/pmc.sh v2_attn_pipe "attn self N=4096" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU | tail -10
AE_ATTN_V=7 bash tools/pmc.sh v2_attn_pipe2 "attn self N=4096" SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE | tail -10
echo "== bench A/B (alternating)"
for i in 1 2; do
  for v in 3 7; do
    AE_ATTN_V=$v timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AE_ATTN_V=$v', d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50'))"
  done
done | tee $OUT/v2_bench_ab.txt
