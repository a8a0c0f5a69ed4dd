mkdir -p gpurun_out/v13
( timeout 300 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "groupnorm" 2>&1 | grep -E "passed|failed" | tail -2
bash tools/ab_lib.sh gncs4 3 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | python -c "
import sys,json
for line in sys.stdin:
    line=line.strip()
    if line.startswith('=='): print(line, end=' ')
    elif line.startswith('{'):
        try:
            d=json.loads(line); print(d['unet_step_ms'], d['unet_step_ms_p50'])
        except Exception as e: print('trunc', line[:80])
"
) > gpurun_out/v13/gncs.txt 2>&1
cat gpurun_out/v13/gncs.txt
