#!/bin/bash
# collect_pmc_table.sh OUTDIR: counter-free trace + FETCH_SIZE pass + WRITE_SIZE pass of the single-stream eager step, then the
# per-kernel HBM table (scripts/pmc_table.py).  The program follows `--` directly (python3, no wrapper), one counter per pass.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pmc_table}; mkdir -p $O
export IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0
ARGS="bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg"
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $ARGS > $O/trace.log 2>&1 && echo "trace done" &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $ARGS > $O/fetch.log 2>&1 && echo "fetch done" &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $ARGS > $O/write.log 2>&1 && echo "write done" &&
python3 scripts/pmc_table.py $O/trace $O/fetch $O/write $O/pmc_kernels.json > $O/pmc_kernels.txt && cat $O/pmc_kernels.txt
# the raw traces are large: keep the summaries only
rm -rf $O/trace $O/fetch $O/write
