#!/usr/bin/env python3
"""Per-shape table of the conv launches of one denoising step (c2: B=16, 256^2), single stream, HIP-event timed."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IDIFF_TWO_STREAMS"] = "0"
from instancediff_amd import ops, pipeline  # noqa: E402
from instancediff_amd.utils.synthetic import make_batch  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model, sde = pipeline.build(phase="test", device=dev, T=1000, seed=0)
    model.set_eval()
    run = bench.StepRunner(model, sde, make_batch(16, 256, seed=1234, mixed=True))
    for _ in range(2):
        run.step()
    torch.cuda.synchronize()
    ops.PROFILE = []
    n = 3
    for _ in range(n):
        run.step()
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for r in ops.PROFILE:
        k = (r['ks'], r['mode'], r['Cin'], r['Cout'], r['Hout'], r['algo'])
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += r['e0'].elapsed_time(r['e1'])
        a[2] += r['flops']
    ops.PROFILE = None
    tot = 0.0
    for k, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        tot += ms / n
        print(f"ks={k[0]} mode={k[1]} Cin={k[2]:4d} Cout={k[3]:4d} H={k[4]:4d} algo={k[5]}  x{c // n:3d}/step  {ms / n:7.3f} ms/step  {ms / c * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s")
    print(f"all convs: {tot:.2f} ms/step")


if __name__ == "__main__":
    main()
