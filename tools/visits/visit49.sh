#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( echo "== previous build (SLP-vectorised softmax: v_pk_mul_f32 / v_pk_add_f32)"; AE_LIB_PATH=$R/anyedit_amd/libanyedit_hip_prev.so timeout 40 python tools/attn_ab_probe.py; echo "== attention_fast.hip with -fno-slp-vectorize"; timeout 40 python tools/attn_ab_probe.py ) > $OUT/v49_attn_slp.txt 2>&1; grep -v amdgpu.ids $OUT/v49_attn_slp.txt | cut -c1-150
( timeout 60 python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "attention" ) > $OUT/v49_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/v49_pytest.txt | cut -c1-160
