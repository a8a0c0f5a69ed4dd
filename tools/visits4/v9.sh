#!/bin/bash
# round 4 visit 9: product build with the ping-pong loop on by default: checksums, full GPU suite, bench A/B against AE_GEMM_PP=0
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 60 python tools/gemm_conv_checksum.py 2>/dev/null | grep -v amdgpu ) > $OUT/r04_v9_sum.txt
if cmp -s $OUT/r04_v9_sum.txt tools/visits/v50_checksums_reference.txt; then echo "checksums IDENTICAL"; else echo "checksums DIFFER"; diff $OUT/r04_v9_sum.txt tools/visits/v50_checksums_reference.txt | head -10; fi
for i in 1 2; do
  for pp in 0 15; do
    echo -n "pp=$pp round $i: "
    AE_GEMM_PP=$pp python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"
  done
done | tee $OUT/r04_v9_bench_pp.txt
( timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ) > $OUT/r04_v9_pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/r04_v9_pytest_gpu.txt | cut -c1-200
