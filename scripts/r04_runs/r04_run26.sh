#!/bin/bash
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_train_gpu.py -x -q -m gpu > $O/t26.log 2>&1; rc=$?; tail -3 $O/t26.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e26_a.json 2>/dev/null && echo "standalone $(grep -o '"value": [0-9.]*' $O/e26_a.json)" &&
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/e26_b.json 2>/dev/null && echo "default line without cpu leg: $(grep -o '"it_per_s": [0-9.]*' $O/e26_b.json)" &&
timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline > $O/e26_c.json 2>/dev/null && echo "default line without cpu leg and roofline pass: $(grep -o '"it_per_s": [0-9.]*' $O/e26_c.json)" &&
timeout -k 10 300 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e26_d.json 2>/dev/null && echo "standalone $(grep -o '"value": [0-9.]*' $O/e26_d.json)"
