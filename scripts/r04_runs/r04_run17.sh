#!/bin/bash
# single-stream kernel stats of the training step after the reduce rewrite
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r04/tp17}; mkdir -p $O
export IDIFF_TRAIN_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --mode train --batch 32 --steps 3 --warmup 2 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_kernel_stats_1s.csv; rm -rf $O/stats
tail -2 $O/stats.log | cut -c1-200
