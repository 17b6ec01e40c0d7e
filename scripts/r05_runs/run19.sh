#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run19; mkdir -p $O
for v in base xa_nos xa_nopv xa_none; do
  if [ $v = base ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$v.so; fi
  echo "== $v"; python3 scripts/r05_runs/run19_xattn_ablation.py 2>&1 | grep -v amdgpu.ids | tee -a $O/xattn_ablation.txt || exit 1
done
unset IDIFF_LIB
python3 scripts/step_convs.py 2>&1 | grep -v amdgpu.ids | tee $O/step_convs.txt
