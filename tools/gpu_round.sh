#!/bin/bash
# One GPU-box visit: smoke, parity tests (crash-isolated via xdist), bench, rocprof summary.  Logs -> gpurun_out/
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
OUT=$PWD/gpurun_out
mkdir -p $OUT
echo "== host: $(nproc) cores, $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)" | tee $OUT/host.log
rocminfo 2>/dev/null | grep -m3 -E "gfx|Marketing" | tee -a $OUT/host.log
( timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" ) > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -4 $OUT/smoke.log
( timeout ${PYTEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -n 1 --timeout 600 ${PYTEST_ARGS:-} ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -3
grep -E "^(FAILED|ERROR)" $OUT/pytest_gpu.log | head -60
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  ( timeout 1500 python bench.py --steps ${BENCH_STEPS:-1} --warmup 1 ${BENCH_ARGS:-} ) > $OUT/bench.log 2>&1; echo "bench rc=$?"; tail -3 $OUT/bench.log | cut -c1-3000
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  cd /tmp && ( timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $OLDPWD/bench.py --steps 1 --warmup 1 --ddim-steps ${PROF_DDIM_STEPS:-10} --no-cpu-baseline ) > $OUT/rocprof.log 2>&1; echo "rocprof rc=$?"; cd $OLDPWD
  find $OUT/prof -name "*kernel_stats*" | head -3
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
fi
