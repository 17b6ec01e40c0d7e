#!/bin/bash
# ab_kernel_stats.sh OUTDIR LIB...: rocprofv3 kernel stats of a short headline run per library variant (instancediff_amd/variants/
# libidiff_<LIB>.so), then the rows of the kernels named in $KERNELS (regex).  One process per variant, program directly after `--`.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
for lib in "$@"; do
  export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$lib -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/$lib.log 2>&1
  f=$(find $O/$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib: $(tail -1 $O/$lib.log | cut -c1-120)"
  grep -E "${KERNELS:-scoremap4|combine}" $f | cut -c1-160
  cp $f $O/${lib}_kernel_stats.csv; rm -rf $O/$lib
done
