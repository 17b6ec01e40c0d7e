#!/bin/bash
# Collects the round's measurement artefacts on the GPU box into gpurun_out/r02/ (copied to profiles/r02/ afterwards):
#   bench lines (headline, 224, 512, irsde, train), rocprofv3 kernel stats (timed configuration and single-stream eager),
#   the steady-state step table, and the PMC traffic of the dominant kernel (separate FETCH_SIZE / WRITE_SIZE passes).
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r02}; mkdir -p $O
python3 bench.py > $O/a_bench_headline.json 2> $O/a_bench_headline.err; echo "headline done" 
python3 bench.py --size 224 --batch 16 --no-cpu-baseline > $O/b_bench_224.json 2>/dev/null
python3 bench.py --size 512 --batch 8 --steps 5 --warmup 2 --no-cpu-baseline > $O/c_bench_512_b8.json 2>/dev/null
python3 bench.py --mode irsde > $O/d_bench_irsde.json 2>/dev/null
python3 bench.py --mode train --batch 32 --steps 5 --warmup 2 > $O/e_bench_train_b32.json 2>/dev/null; echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_two_stream -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/stats_two_stream.log 2>&1
IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single_stream -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/stats_single_stream.log 2>&1
python3 scripts/step_trace.py $O/stats_single_stream $O/g_steady_state_step_single_stream.csv > $O/g_steady_state_step_single_stream.txt; echo "traces done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write.log 2>&1
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; echo "pmc done"
find $O -name "*kernel_stats.csv" | head; ls $O
