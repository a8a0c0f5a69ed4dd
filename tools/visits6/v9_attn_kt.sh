mkdir -p gpurun_out/v9
( timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "feed_forward_fused or pipelined_equals or attention_core or run_to_run" -s 2>&1 | grep -v Warn | tail -25
timeout 200 python tools/ff_fused_lab.py --rounds 2 2>/dev/null | tail -1 | cut -c1-900
for i in 1 2 3; do
  for v in 0 1; do
  echo "== AE_ATTN_KT128=$v (round $i)"; AE_ATTN_KT128=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['unet_step_ms'], d['unet_step_ms_p50'])"
  done
done ) > gpurun_out/v9/attn_kt.txt 2>&1
cat gpurun_out/v9/attn_kt.txt
