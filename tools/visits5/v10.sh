#!/bin/bash
# Round 5, visit 10: where the GEGLU launch's time goes — cycle buckets of one SIMD's two waves (AE_GEMM_LAB build), and the launch without its epilogue.
set -u
B=$PWD/tools/ubench/build; OUT=$PWD/gpurun_out; mkdir -p $OUT
( $B/pp_lab g; $B/pp_noepi g; $B/pp_plain g; $B/pp_lab x | tail -12 ) 2>&1 | tee $OUT/v10_geglu_lab.txt
