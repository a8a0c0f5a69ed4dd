#!/bin/bash
# Lab builds of the fused cross-attention kernel (as tools/ff_variants.sh):  tools/xa_variants.sh tag "-DXA_LAB=1" ...
set -e
cd "$(dirname "$0")/../anyedit_amd"
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  mkdir -p build_lab
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-inline-asm -fno-slp-vectorize $flags -c csrc/xattn_fused.hip -o build_lab/xattn_fused_$tag.o
  objs=$(ls build/*.o | grep -v xattn_fused.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o libanyedit_hip_$tag.so $objs build_lab/xattn_fused_$tag.o
  echo built libanyedit_hip_$tag.so
done
