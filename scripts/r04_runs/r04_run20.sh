#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py -x -q -m gpu > $O/t20.log 2>&1; rc=$?; tail -5 $O/t20.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e20_train.json 2> $O/e20_train.err && cat $O/e20_train.json | cut -c1-260
