#!/bin/bash
# ScoreMapModule phase of each net on a side stream, overlapping the net's mid blocks (experiment)
for cfg in "" "IDIFF_SMM_SIDE=1" "" "IDIFF_SMM_SIDE=1"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[$cfg] /"
done
