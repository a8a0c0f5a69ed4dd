export PYTHONDONTWRITEBYTECODE=1
run() { python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(sys.argv[1], round(d['unet_step_ms'],4), round(d['unet_step_ms_p50'],4), round(d['value'],3))" "$1"; }
for i in 1 2 3; do
  AE_GN_SPLITK=0 run off
  AE_GN_SPLITK=1 run fold256
  AE_GN_SPLITK=1 AE_GN_SPLITK_T=1024 run fold1024
done
