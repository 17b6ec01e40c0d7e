#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t18_train.log 2>&1; rc=$?; tail -3 $O/t18_train.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e18_train.json 2> $O/e18_train.err && cat $O/e18_train.json | cut -c1-260 &&
timeout -k 10 400 python scripts/train_aten_sources.py > $O/p18_aten.log 2>&1; tail -70 $O/p18_aten.log
