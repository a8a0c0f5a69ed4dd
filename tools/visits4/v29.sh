#!/bin/bash
# round 4 visit 29: effective shader clock per kernel symbol inside the UNet evaluation (GRBM_GUI_ACTIVE / duration, MI355X_MICROARCH.md "DVFS give-back"),
# and the matrix-pipe / VALU busy counters of the same pass — which kernels of the step run clock-limited.
set -u
bash tools/pmc2.sh bench_clock "@bench" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA > /dev/null 2>&1; echo "rc=$?"
grep -E "^[a-z]|effective_clock|avg_duration|MFMA_BUSY|SQ_BUSY_CYCLES" gpurun_out/pmc_bench_clock.txt | head -150
