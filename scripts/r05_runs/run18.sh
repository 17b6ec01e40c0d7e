#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run18; mkdir -p $O
python3 scripts/r05_runs/run18_memproj_lin.py 2>&1 | grep -v amdgpu.ids | tee $O/micro.txt &&
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "memproj or linear or smm or token or chain or scoremap" 2>&1 | tail -3 | tee $O/tests.txt &&
for v in base xattn3 base xattn3; do
  if [ $v = base ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$v.so; fi
  echo "== $v"; python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2> $O/bench_$v.err | tee -a $O/bench_$v.json | cut -c1-200 || exit 1
done
