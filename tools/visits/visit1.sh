#!/bin/bash
# Round-3 visit 1: the new bench-shape parity tests + whole -m gpu suite, the bench line with power / clock samples taken beside it,
# the counter list of this rocprofv3, and PMC passes (incl. GRBM_GUI_ACTIVE -> effective clock) over the latency-bound dense GEMMs.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 2400 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=15 ) > $OUT/v1_pytest.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^FAILED|^ERROR|rel-L2|grad |training step|UNet batch|conv3x3 \[|kl-f8|SAM ViT" $OUT/v1_pytest.log | tail -80
# power / clock samples while the bench runs (the driver's smi.*.json are not visible to the builder)
( rocm-smi --showmaxpower --showpower --showclocks 2>&1 | head -60 ) > $OUT/v1_smi_idle.txt
( while true; do date +%s.%N; rocm-smi --showpower --showclocks --showuse --json 2>/dev/null; sleep 0.2; done ) > $OUT/v1_smi_bench.jsonl &
SMI=$!
( timeout 900 python bench.py --steps 6 --warmup 2 ) > $OUT/v1_bench.json 2> $OUT/v1_bench.err; echo "bench rc=$?"
kill $SMI 2>/dev/null
cut -c1-1500 $OUT/v1_bench.json
cp $OUT/kernels_by_shape.json $OUT/v1_kernels_by_shape.json 2>/dev/null
( rocprofv3 -L 2>&1 | grep -E "^\s*(Name|.*SQ_|.*GRBM_|.*TCC_|.*TCP_)" | head -400 ) > $OUT/v1_counters.txt; wc -l $OUT/v1_counters.txt
python tools/kbench.py "gemm " > $OUT/v1_kbench_gemm.txt 2>&1; tail -22 $OUT/v1_kbench_gemm.txt
bash tools/pmc2.sh v1_gemm_a "gemm " SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE > $OUT/v1_pmc_a.out 2>&1; tail -3 $OUT/v1_pmc_a.out
bash tools/pmc2.sh v1_gemm_b "gemm " SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE > $OUT/v1_pmc_b.out 2>&1; tail -3 $OUT/v1_pmc_b.out
bash tools/pmc2.sh v1_clk "conv3x3 res" SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE > $OUT/v1_pmc_clk_conv.out 2>&1; tail -3 $OUT/v1_pmc_clk_conv.out
bash tools/pmc2.sh v1_clk_attn "attn self" SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE > $OUT/v1_pmc_clk_attn.out 2>&1; tail -3 $OUT/v1_pmc_clk_attn.out
ls -la $OUT | tail -30
