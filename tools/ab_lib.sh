#!/bin/bash
# A/B of two builds of the library on ONE GPU box (box-to-box spread is larger than most single changes: DESIGN.md §7a).
#   here (no GPU):  python -m anyedit_amd.build --variant noslp attention_bwd.hip=-fno-slp-vectorize attention.hip=-fno-slp-vectorize
#   on the box:     bash tools/ab_lib.sh noslp 2 python tools/bench_train.py --steps 10 --warmup 2
# runs the command alternately with the product library and with libanyedit_hip_<tag>.so (AE_LIB_PATH), <rounds> times each, and prints
# the last line of every run.
set -u
TAG=$1; ROUNDS=$2; shift 2
VAR=$PWD/anyedit_amd/libanyedit_hip_$TAG.so
[ -f "$VAR" ] || { echo "missing $VAR: build it first (python -m anyedit_amd.build --variant $TAG file.hip=-flag ...)"; exit 2; }
for i in $(seq 1 $ROUNDS); do
  echo "== product library (round $i)"; env -u AE_LIB_PATH "$@" 2>/dev/null | tail -1
  echo "== variant $TAG (round $i)"; AE_LIB_PATH=$VAR "$@" 2>/dev/null | tail -1
done
