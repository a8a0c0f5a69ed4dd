#!/bin/bash
# round 4 visit 33: the full GPU suite at the final commit (device code = 38b9d37's), edit control for real
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
( AE_TEST_EDIT_CONTROL=1 timeout 1800 python -m pytest tests -m gpu -q -s -p no:cacheprovider --durations=8 ) > gpurun_out/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/pytest_gpu_full.log | tail -3; grep -E "grad gate.weight|grad task_embs" gpurun_out/pytest_gpu_full.log
