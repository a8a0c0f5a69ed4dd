mkdir -p gpurun_out/v11
( timeout 200 python tools/xattn_fused_lab.py --rounds 2 2>/dev/null | tail -1
timeout 1500 python -m pytest tests/test_hip_unet.py tests/test_hip_fullsize.py tests/test_hip_sam_anysd.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -5
for i in 1 2 3; do
  for v in 0 1; do
  echo "== AE_XATTN_FUSED=$v (round $i)"; AE_XATTN_FUSED=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unet_step_ms'], d['unet_step_ms_p50'])"
  done
done ) > gpurun_out/v11/xattn.txt 2>&1
cat gpurun_out/v11/xattn.txt
