#!/bin/bash
# Round 5, visit 1: throttle / DVFS evidence, the hygiene batch on hardware (attention fences in every instantiation, pooled
# router-gradient estimator), attention timings after the fence change, PMC of the self-attention kernel BEFORE this round's kernel work.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
B=$R/tools/ubench/build
echo "== throttle probes"
timeout 120 python tools/throttle_probe.py $OUT/v1_throttle_mfma16_random.json -- $B/mfma_burn 5 0 | cut -c1-1500
timeout 120 python tools/throttle_probe.py $OUT/v1_throttle_mfma32_random.json -- $B/mfma_burn 4 1 | cut -c1-600
timeout 120 python tools/throttle_probe.py $OUT/v1_throttle_mfma16_zeros.json -- $B/mfma_burn 4 0 z | cut -c1-600
AE_LAB_ITERS=20000 timeout 120 python tools/throttle_probe.py $OUT/v1_throttle_convlab.json -- $B/pp_plain c | cut -c1-600
timeout 300 python tools/throttle_probe.py $OUT/v1_throttle_bench.json -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline | cut -c1-1500
for f in mfma16_random mfma32_random mfma16_zeros convlab; do echo "-- $f"; python -c "import json;d=json.load(open('$OUT/v1_throttle_$f.json'));print(d['workload_tail'][-700:])"; done
echo "== tests"
( timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_sam_anysd.py "tests/test_hip_bench_shapes.py::test_training_step_full_size_vs_oracle_autograd_with_control" -m gpu -q -s -x -p no:cacheprovider ) > $OUT/v1_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|pooled|grad gate|grad task_embs|Error|error" $OUT/v1_pytest.log | tail -30
echo "== kbench attn"
( timeout 300 python tools/kbench.py attn ) > $OUT/v1_kbench_attn.txt 2>&1; cat $OUT/v1_kbench_attn.txt | grep -v amdgpu.ids
echo "== pmc attn self (before)"
bash tools/pmc.sh v1_attn_before "attn self N=4096" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU | tail -12
bash tools/pmc.sh v1_attn_before2 "attn self N=4096" SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE | tail -12
