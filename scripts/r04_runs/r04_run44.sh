#!/bin/bash
# phase offset between the two nets (the noise net starts when the drift net has finished its encoder / its ScoreMapModule phase)
for cfg in "" "IDIFF_NET_OFFSET=enc_done" "IDIFF_NET_OFFSET=smm_done" "" "IDIFF_NET_OFFSET=enc_done" "IDIFF_NET_OFFSET=smm_done"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-roofline --no-train-leg 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | sed "s/^/[$cfg] /"
done
