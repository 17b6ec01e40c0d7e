#!/bin/bash
# run-to-run spread of the headline on one box: five default lines without the CPU / training legs
O=gpurun_out/r04/r35; mkdir -p $O
for i in 1 2 3 4 5; do
  python3 bench.py --no-cpu-baseline --no-train-leg 2>/dev/null > $O/h$i.json || exit 1
  python3 -c "
import json; d=json.load(open('$O/h$i.json')); r=d['roofline']; print('run $i: %.2f steps/s  %.3f ms/step  conv_wino4 frac %.4f (%.4f ms/launch)  conv_wino4h frac %.4f' % (d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'], r['kernels'][1]['frac']))"
done
