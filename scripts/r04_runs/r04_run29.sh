#!/bin/bash
# register-resident channel LayerNorm (forward + backward dx): tests, train line
O=gpurun_out/r04/r29; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py tests/test_ops_gpu.py tests/test_ops2_gpu.py tests/test_attn_pin_gpu.py -x -q -m gpu > $O/t29.log 2>&1; rc=$?; tail -3 $O/t29.log; [ $rc -eq 0 ] || exit 1
for i in 1 2; do timeout -k 10 300 python bench.py --mode train --batch 32 --steps 8 --warmup 2 2>/dev/null | grep -o '"value": [0-9.]*'; done
