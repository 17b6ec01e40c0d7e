#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t21.log 2>&1; rc=$?; tail -3 $O/t21.log; [ $rc -eq 0 ] || exit 1
for v in 1 0 1 0; do
IDIFF_LAZY_PACK=$v timeout -k 10 300 python bench.py --mode train --batch 32 --steps 8 --warmup 2 > $O/e21_train_lazy$v.json 2> $O/e21_train.err || exit 1
echo "lazy=$v $(cut -c1-150 $O/e21_train_lazy$v.json | grep -o '"value": [0-9.]*')"
done
