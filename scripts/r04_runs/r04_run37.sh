#!/bin/bash
# stream priorities for the two nets' streams (graph replay and eager), A/B in one call
for cfg in "" "IDIFF_STREAM_PRIO=hl" "IDIFF_STREAM_PRIO=lh" "" "IDIFF_STREAM_PRIO=hl"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/graph [$cfg] /"
done
for cfg in "IDIFF_HIP_GRAPH=0" "IDIFF_HIP_GRAPH=0 IDIFF_STREAM_PRIO=hl" "IDIFF_HIP_GRAPH=0" "IDIFF_HIP_GRAPH=0 IDIFF_STREAM_PRIO=hl"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/eager [$cfg] /"
done
