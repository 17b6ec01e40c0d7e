#!/bin/bash
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -q -k "wgrad" > $O/t11_wgrad.log 2>&1; rc=$?
grep -E "^FAILED|passed|failed" $O/t11_wgrad.log | tail -20
if [ $rc -ne 0 ]; then grep -E "AssertionError|assert " $O/t11_wgrad.log | head -20; exit 1; fi
timeout -k 10 400 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e11_train.json 2> $O/e11_train.err; python3 -c "
import json; d=json.load(open('$O/e11_train.json')); print('train wgrad4', d['value'], d['ms_per_step'])"
IDIFF_WGRAD4=0 timeout -k 10 400 python bench.py --mode train --batch 32 --steps 6 --warmup 2 > $O/e11_train_w2.json 2> $O/e11_train_w2.err; python3 -c "
import json; d=json.load(open('$O/e11_train_w2.json')); print('train wgrad2', d['value'], d['ms_per_step'])"
