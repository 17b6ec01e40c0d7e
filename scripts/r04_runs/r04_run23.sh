#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t23.log 2>&1; rc=$?; tail -3 $O/t23.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 400 python scripts/train_bgemm_probe.py > $O/p23_bgemm.log 2>&1; head -16 $O/p23_bgemm.log | tail -14
for v in 1 2; do
timeout -k 10 300 python bench.py --mode train --batch 32 --steps 8 --warmup 2 > $O/e23_train_$v.json 2> $O/e23_train.err || exit 1
echo "run $v $(cut -c1-150 $O/e23_train_$v.json | grep -o '"value": [0-9.]*')"
done
