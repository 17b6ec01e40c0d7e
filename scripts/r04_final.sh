#!/bin/bash
# final checkpoint of the round: the whole GPU suite in one process, the headline line, the training profile (two streams / one stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r04/final}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/t_gpu_tests_final.log 2>&1; rc=$?; tail -3 $O/t_gpu_tests_final.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python3 bench.py > $O/a_bench_headline.json 2> $O/a_bench_headline.err || exit 1
cut -c1-200 $O/a_bench_headline.json; grep -o '"train": {[^}]*}' $O/a_bench_headline.json | cut -c1-200
timeout -k 10 300 python3 bench.py --mode train --batch 32 --steps 8 --warmup 2 > $O/e_bench_train_b32.json 2>/dev/null || exit 1
cut -c1-200 $O/e_bench_train_b32.json
bash scripts/train_profile.sh $O/train 32 > $O/train_profile.txt 2>&1; cp $O/train/train_kernel_stats.csv $O/train_kernel_stats.csv; rm -rf $O/train
export IDIFF_TRAIN_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --mode train --batch 32 --steps 3 --warmup 2 > $O/stats1.log 2>&1
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_single_stream.csv; rm -rf $O/stats1
head -12 $O/train_profile.txt
