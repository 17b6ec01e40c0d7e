#!/bin/bash
# diagnosis of the sparse RAG-only output errors of the all-DMA conv_wino4 kernel
set -o pipefail
O=gpurun_out/r04
mkdir -p $O
for v in "" mask stlast; do
  if [ -n "$v" ]; then export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$v.so; fi
  echo "=== variant '${v:-default}'"
  timeout -k 10 120 python scripts/w4_diag.py 3 2 32 64 28 56 2>&1 | grep -v amdgpu.ids | tail -22
  timeout -k 10 120 python scripts/w4_diag.py 3 2 32 64 28 28 2>&1 | grep -v amdgpu.ids | tail -8
done
