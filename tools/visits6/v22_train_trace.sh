#!/bin/bash
# Training step (configs[3]): kernel trace at the head, to price the glue launches.
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out/v22; mkdir -p $OUT
timeout 300 python tools/bench_train.py --steps 10 --warmup 2 2>/dev/null | tail -1 | cut -c1-300
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o train -- python $R/tools/bench_train.py --steps 10 --warmup 2 > /dev/null 2>&1
cd $R
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/train_kernel_stats.csv
rm -rf $OUT/prof
wc -l $OUT/train_kernel_stats.csv
