#!/bin/bash
# closing check of the round on the final tree: the whole GPU suite in one process, the training iteration's single-stream kernel stats
# (scripts/collect_profiles.sh has produced everything else, the headline line included, in its own call)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r04/final}; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t_gpu_tests_final.log 2>&1; rc=$?; tail -3 $O/t_gpu_tests_final.log; [ $rc -eq 0 ] || exit 1
export IDIFF_TRAIN_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --mode train --batch 32 --steps 3 --warmup 2 > $O/stats1.log 2>&1
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_single_stream.csv; rm -rf $O/stats1
python3 - $O/train_kernel_stats_single_stream.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("single stream: %.1f ms of kernels per iteration, %d launches" % (sum(float(r["TotalDurationNs"]) for r in rows) / 5e6, sum(int(r["Calls"]) for r in rows) / 5))
PY
