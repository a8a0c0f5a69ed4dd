#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py tests/test_hip_fullsize.py -m gpu -q -x -p no:cacheprovider -k "groupnorm or norm or colstats or unet_full_size_vs or determinism or resblock" ) > $OUT/v26_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|Error|assert" $OUT/v26_pytest.log | tail -5
cd /tmp && ( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof26 -o b -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 6 --no-cpu-baseline --no-roofline ) > $OUT/v26_rocprof.log 2>&1; cd $R
f=$(find $OUT/prof26 -name "*kernel_stats.csv" | head -1); grep -E "gn_finalize|gn_apply|colstats" "$f" | cut -c1-160; rm -rf $OUT/prof26
python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],3), round(d['unet_step_ms_p50'],3))"
