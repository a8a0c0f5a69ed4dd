#!/bin/bash
# round 4 visit 31: what the epilogue costs the short-K launches — the pp lab binaries with and without the epilogue (-DAE_GEMM_LAB_NOEPI: one store keeps the
# accumulators alive), all 192x320 shapes incl. the two GEGLU projections, and the 32x32-level shapes of the 128x128 tile.
set -u
B=tools/ubench/build
{ echo "== with epilogue"; $B/pp_plain x; $B/pp_plain m | tail -5; echo "== without epilogue (AE_GEMM_LAB_NOEPI)"; $B/pp_noepi x; $B/pp_noepi m | tail -5;
  echo "== phase buckets (AE_GEMM_LAB), GEGLU shapes"; $B/pp_lab x | grep -A2 geglu; } 2>&1 | tee gpurun_out/r04_v31_epilogue_share.txt
