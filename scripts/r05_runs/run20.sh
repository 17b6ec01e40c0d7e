#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run20; mkdir -p $O
echo "== splits 64 (r04 rule), lazy rescale"; IDIFF_XATTN_SPLITS=64 python3 scripts/r05_runs/run19_xattn_ablation.py 2>&1 | grep -v amdgpu.ids | tee -a $O/xattn.txt
echo "== splits 32"; python3 scripts/r05_runs/run19_xattn_ablation.py 2>&1 | grep -v amdgpu.ids | tee -a $O/xattn.txt
echo "== splits 32 + two S chains"; IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_xa_s2.so python3 scripts/r05_runs/run19_xattn_ablation.py 2>&1 | grep -v amdgpu.ids | tee -a $O/xattn.txt
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "attn or xattn or smm or chain or scoremap or invarian" 2>&1 | tail -3 | tee $O/tests.txt &&
for v in s64 s32 s2 s64 s32 s2; do
  unset IDIFF_LIB IDIFF_XATTN_SPLITS
  [ $v = s64 ] && export IDIFF_XATTN_SPLITS=64
  [ $v = s2 ] && export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_xa_s2.so
  echo "== $v"; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2> $O/bench_$v.err | tee -a $O/bench_$v.json | cut -c1-200 || exit 1
done
