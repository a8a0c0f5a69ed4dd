#!/bin/bash
# round 4 visit 3: first run of the ping-pong main loop (AE_GEMM_PP bit flags): same bits? faster?
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for pp in 0 15; do
  ( AE_GEMM_PP=$pp timeout 60 python tools/gemm_conv_checksum.py 2>/dev/null | grep -v amdgpu ) > $OUT/r04_v3_sum_pp$pp.txt
  if cmp -s $OUT/r04_v3_sum_pp$pp.txt tools/visits/v50_checksums_reference.txt; then echo "pp=$pp: checksums IDENTICAL"; else echo "pp=$pp: checksums DIFFER"; diff $OUT/r04_v3_sum_pp$pp.txt tools/visits/v50_checksums_reference.txt | head -10; fi
done
for pp in 0 15; do
  echo "== kbench AE_GEMM_PP=$pp"
  AE_GEMM_PP=$pp timeout 120 python tools/kbench.py conv3x3 2>&1 | grep -v "^#"
  AE_GEMM_PP=$pp timeout 120 python tools/kbench.py "gemm" 2>&1 | grep -E "ff1|ff2 L1|skip1x1 960"
done | tee $OUT/r04_v3_kbench_pp.txt
for i in 1 2; do
  for pp in 0 15; do
    echo -n "pp=$pp round $i: "
    AE_GEMM_PP=$pp python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"
  done
done | tee $OUT/r04_v3_bench_pp.txt
( AE_GEMM_PP=15 timeout 200 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or gemm or colstats" ) > $OUT/r04_v3_pytest_pp15.txt 2>&1; echo "pp15 pytest rc=$?"; tail -3 $OUT/r04_v3_pytest_pp15.txt | cut -c1-200
