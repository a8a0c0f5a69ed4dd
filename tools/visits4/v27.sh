#!/bin/bash
# round 4 visit 27: rocprofv3 kernel stats of the graphed bench with and without the LayerNorm fold (per-symbol durations inside the graph:
# the un-graphed per-op table is polluted by host-side launch latency on the short GEMMs).
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for cfg in "AE_LN_FOLD=0" "AE_LN_FOLD=1"; do
  tag=$(echo $cfg | tr '=' '_')
  cd /tmp && ( env $cfg timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o bench -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline ) > $OUT/v27_rocprof_$tag.log 2>&1; echo "$cfg rocprof rc=$?"; cd $R
  f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/v27_kernel_stats_$tag.csv
  rm -rf $OUT/prof_$tag
done
head -5 $OUT/v27_kernel_stats_AE_LN_FOLD_1.csv | cut -c1-200
