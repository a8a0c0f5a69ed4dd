#!/bin/bash
set -u
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
( echo "== previous build"; AE_LIB_PATH=$R/anyedit_amd/libanyedit_hip_prev.so timeout 60 python tools/gn_slab_probe.py; echo "== this build"; timeout 60 python tools/gn_slab_probe.py ) > $OUT/v46_gn_slab.txt 2>&1; cat $OUT/v46_gn_slab.txt | grep -v amdgpu.ids | cut -c1-160
( timeout 100 python -m pytest tests/test_hip_ops.py tests/test_hip_unet.py -m gpu -q -p no:cacheprovider -k "groupnorm or gn or norm or resblock or unet_tiny" ) > $OUT/v46_pytest.txt 2>&1; echo "pytest rc=$?"; tail -3 $OUT/v46_pytest.txt | cut -c1-160
