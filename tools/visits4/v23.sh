#!/bin/bash
# round 4 visit 23: the up-sampling convs on the ping-pong loop (AE_GEMM_PP flag 16)
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out
for pp in 15 31; do
  ( AE_GEMM_PP=$pp timeout 60 python tools/gemm_conv_checksum.py 2>/dev/null | grep -v amdgpu ) > $OUT/r04_v23_sum_pp$pp.txt
  if cmp -s $OUT/r04_v23_sum_pp$pp.txt tools/visits/v50_checksums_reference.txt; then echo "pp=$pp: checksums IDENTICAL"; else echo "pp=$pp: checksums DIFFER"; diff $OUT/r04_v23_sum_pp$pp.txt tools/visits/v50_checksums_reference.txt | head; fi
done
{ for pp in 15 31 15 31; do echo "== AE_GEMM_PP=$pp"; AE_GEMM_PP=$pp timeout 100 python tools/kbench.py "conv3x3 up" 2>&1 | grep -v "amdgpu\|^#"; done
  run() { echo -n "$1: "; env $1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['value'], 3), 'img/s', round(d['unet_step_ms'], 3), 'ms per UNet step')"; }
  run "AE_GEMM_PP=15"; run "AE_GEMM_PP=31"; run "AE_GEMM_PP=15"; run "AE_GEMM_PP=31"; } | tee $OUT/r04_v23_ups_pp.txt
( AE_GEMM_PP=31 timeout 300 python -m pytest tests/test_hip_ops.py tests/test_hip_bench_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv" ) > $OUT/r04_v23_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/r04_v23_pytest.txt | tail -1
