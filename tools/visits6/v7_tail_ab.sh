mkdir -p gpurun_out/v7
( timeout 1200 python -m pytest tests/test_hip_unet.py tests/test_hip_fullsize.py tests/test_hip_sam_anysd.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2 3; do
  for v in 0 1; do
  echo "== AE_FF_TAIL=$v (round $i)"; AE_FF_TAIL=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unet_step_ms'], d['unet_step_ms_p50'])"
  done
done ) > gpurun_out/v7/tail_ab.txt 2>&1
grep -v Warn gpurun_out/v7/tail_ab.txt
