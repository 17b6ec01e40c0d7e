#!/bin/bash
# bench A/B of environment variants: r04_bench.sh TAG [ENV=VAL ...]
O=gpurun_out/r04; mkdir -p $O
TAG=$1; shift
env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-leg > $O/b_$TAG.json 2> $O/b_$TAG.err && python3 -c "
import json; d=json.load(open('$O/b_$TAG.json')); print('$TAG', d['value'], d['ms_per_step'], d['launches_per_step'], [(k['kernel'][:18], k['frac'], k['ms_per_step_single_stream'], k['avg_launch_ms'], k['launches_per_step']) for k in d['roofline']['kernels']])"
