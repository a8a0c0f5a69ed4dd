mkdir -p gpurun_out/v4
( timeout 600 python -m pytest tests/test_hip_unet.py tests/test_hip_fullsize.py -x -q -m gpu 2>&1 | tail -5
for i in 1 2 3; do
  echo "== AE_FF_FUSED=0 (round $i)"; AE_FF_FUSED=0 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unet_step_ms'], d['unet_step_ms_p50'])"
  echo "== AE_FF_FUSED=1 (round $i)"; AE_FF_FUSED=1 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unet_step_ms'], d['unet_step_ms_p50'])"
done ) > gpurun_out/v4/ff_ab.txt 2>&1
cat gpurun_out/v4/ff_ab.txt
