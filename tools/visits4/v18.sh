#!/bin/bash
# round 4 visit 18: per-kernel table of the step with the ping-pong loops on
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_v18_bench.json
cp gpurun_out/kernels_by_shape.json gpurun_out/r04_v18_kernels_by_shape.json
python -c "
import json; d=json.load(open('gpurun_out/r04_v18_bench.json')); print(d['value'], d['unet_step_ms'], d.get('unet_step_ms_p50')); print(json.dumps(d['roofline']))"
