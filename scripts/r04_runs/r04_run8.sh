#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
IDIFF_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --mode train --grad-wire bf16 --batch 8 --steps 2 --warmup 1 > $O/a_bench_gpus2_train_bf16wire_rehearsal.log 2>&1; echo "rehearsal rc $?"; tail -3 $O/a_bench_gpus2_train_bf16wire_rehearsal.log | cut -c1-600
bash scripts/r04_bench.sh gnfused2 IDIFF_GN_FUSED=1
bash scripts/r04_bench.sh default2
