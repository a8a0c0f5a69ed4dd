#!/bin/bash
# round 4 visit 19: SAM windowed attention with resident keys (one block of seven waves per window and head)
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 600 python -m pytest tests/test_hip_sam_anysd.py tests/test_hip_ops.py tests/test_hip_bench_shapes.py tests/test_hip_sam_decoder.py -m gpu -q -x -p no:cacheprovider -k "relpos or sam" ) > $OUT/r04_v19_pytest.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/r04_v19_pytest.txt | tail -2
{ for w in 0 2 1 0 2 1; do echo "AE_ATTN_WIN=$w: "; AE_ATTN_WIN=$w timeout 200 python tools/bench_sam.py 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('  encoder p50', round(d['latency_ms_p50'], 3), 'ms;', {k: (v['calls'], round(v['ms'] * 1e3 / v['calls'], 1)) for k, v in d['kernels'].items() if 'attn' in k})"
done; } | tee $OUT/r04_v19_sam_win.txt
