#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/rNN/ (copied to profiles/rNN/ afterwards):
#   per-kernel HBM table of the steady-state step (separate FETCH_SIZE / WRITE_SIZE passes + a counter-free trace), rocprofv3 kernel
#   stats of the timed configuration (two streams, HIP graph) and of the single-stream eager step, the steady-state step table,
#   secondary bench lines (224, 512, irsde, train) and the training step's kernel stats.
# The headline line (with the CPU baseline, the training leg and the PMC evidence) runs last, once pmc_kernels.json is in place.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r04}; mkdir -p $O
bash scripts/collect_pmc_table.sh $O/pmc > $O/pmc.log 2>&1; cp $O/pmc/pmc_kernels.json $O/pmc/pmc_kernels.txt $O/ 2>/dev/null; echo "pmc table done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_two_stream -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/stats_two_stream.log 2>&1
cp $(find $O/stats_two_stream -name "*kernel_stats.csv" | head -1) $O/stats_two_stream_kernel_stats.csv; rm -rf $O/stats_two_stream
IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single_stream -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-train-leg > $O/stats_single_stream.log 2>&1
python3 scripts/step_trace.py $O/stats_single_stream $O/g_steady_state_step_single_stream.csv > $O/g_steady_state_step_single_stream.txt
cp $(find $O/stats_single_stream -name "*kernel_stats.csv" | head -1) $O/stats_single_stream_kernel_stats.csv; rm -rf $O/stats_single_stream; echo "traces done"
python3 bench.py --size 224 --batch 16 --no-cpu-baseline --no-train-leg > $O/b_bench_224.json 2>/dev/null
python3 bench.py --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline --no-train-leg > $O/c_bench_512_b8.json 2>/dev/null
python3 bench.py --size 512 --batch 8 --steps 5 --warmup 2 --attn f16 --no-cpu-baseline --no-train-leg > $O/c_bench_512_b8_attn_f16.json 2>/dev/null
python3 bench.py --mode irsde > $O/d_bench_irsde.json 2>/dev/null
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/e_bench_train_b32.json 2>/dev/null; echo "bench lines done"
bash scripts/train_profile.sh $O/train 32 > $O/train_profile.txt 2>&1; cp $O/train/train_kernel_stats.csv $O/train_kernel_stats.csv; rm -rf $O/train
bash scripts/collect_mfma_util.sh $O/mfma_util.json > $O/mfma_util.log 2>&1; echo "mfma util done"
# the headline line reads the counter evidence from profiles/rNN/ (hash-gated): place this collection's tables there first (the box's
# copy of the repository is scratch; the same files are copied into the real profiles/rNN/ afterwards)
R=profiles/${IDIFF_ROUND:-r04}
mkdir -p $R && cp $O/pmc_kernels.json $O/mfma_util.json $O/g_steady_state_step_single_stream.csv $R/ 2>/dev/null
python3 bench.py > $O/a_bench_headline.json 2> $O/a_bench_headline.err; echo "headline done"; cut -c1-400 $O/a_bench_headline.json
ls $O
