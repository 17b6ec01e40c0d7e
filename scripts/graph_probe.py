#!/usr/bin/env python3
"""Probe: does capturing one denoising step in a HIP graph (torch.cuda.CUDAGraph) change the step time?  (timing only:
the captured step has its scalars baked in, so the replayed trajectory is not a valid sampling chain)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instancediff_amd import pipeline  # noqa: E402
from instancediff_amd.utils.synthetic import make_batch  # noqa: E402
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    model, sde = pipeline.build(phase="test", device=dev, T=1000, seed=0)
    model.set_eval()
    run = bench.StepRunner(model, sde, make_batch(16, 256, seed=1234, mixed=True))
    for _ in range(3):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run.step()
    torch.cuda.synchronize()
    print("eager  : %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run.step()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            run.step()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print("graph  : %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))


if __name__ == "__main__":
    main()
