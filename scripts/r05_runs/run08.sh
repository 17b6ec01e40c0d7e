#!/bin/bash
# r05 run 8: training profile with the compact memory (two streams + single stream), per-kernel roofline line
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run08; mkdir -p $O
bash scripts/train_profile.sh $O/train 32 > $O/train_profile.txt 2>&1; rm -rf $O/train/stats; head -60 $O/train_profile.txt
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/train.json 2> $O/train.err; grep -o '"roofline".*' $O/train.json | cut -c1-3000
