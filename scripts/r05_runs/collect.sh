#!/bin/bash
# r05 collection, part A (part B = `IDIFF_ROUND=r05 bash scripts/collect_profiles.sh gpurun_out/r05/c1` in a call of its own: PMC table ->
# kernel stats -> secondary lines -> training profile -> MFMA utilisation -> headline last): the whole GPU suite in one process, smoke,
# the training iteration's single-stream stats, the 2-rank rehearsal of the default line (training leg across ranks)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r05/c1}; mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/t_smoke_final.log 2>&1; tail -2 $O/t_smoke_final.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/t_gpu_tests_final.log 2>&1; rc=$?; tail -3 $O/t_gpu_tests_final.log; [ $rc -eq 0 ] || { tail -40 $O/t_gpu_tests_final.log; exit 1; }
export IDIFF_TRAIN_TWO_STREAMS=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 bench.py --mode train --batch 32 --steps 3 --warmup 2 --no-roofline > $O/stats1.log 2>&1
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats_single_stream.csv; rm -rf $O/stats1
python3 - $O/train_kernel_stats_single_stream.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("training, single stream: %.1f ms of kernels per iteration, %d launches" % (sum(float(r["TotalDurationNs"]) for r in rows) / 5e6, sum(int(r["Calls"]) for r in rows) / 5))
PY
unset IDIFF_TRAIN_TWO_STREAMS
bash scripts/train_mfma_util.sh $O/train_mfma_util.txt > $O/train_mfma_util.log 2>&1; echo "train mfma util done"
bash scripts/step_mfma_util.sh $O/step_mfma_util.txt > $O/step_mfma_util.log 2>&1; echo "step mfma util done"
rocprofv3 --kernel-trace --output-format csv -d $O/trace2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/trace2.log 2>&1
python3 scripts/step_overlap.py $O/trace2 > $O/g_two_stream_step_overlap.txt 2>&1; rm -rf $O/trace2; tail -4 $O/g_two_stream_step_overlap.txt
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-train-leg --no-roofline 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'steps/s', d['ms_per_step'], 'ms/step')"; done > $O/a_bench_headline_repeats.txt; cat $O/a_bench_headline_repeats.txt
IDIFF_BENCH_REHEARSAL=1 timeout -k 10 600 python3 bench.py --gpus 2 --no-cpu-baseline --no-roofline > $O/a_bench_gpus2_default_line_rehearsal.json 2> $O/a_bench_gpus2_default_line_rehearsal.err; echo "rehearsal rc=$?"
ls $O
