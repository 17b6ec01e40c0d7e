"""Which lines of the training step still launch ATen kernels?  One profiled iteration (torch.profiler, with_stack), ATen ops that
launched a kernel grouped by the innermost frame inside this package.
    python scripts/train_aten_sources.py [--batch 32]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    from instancediff_amd import pipeline
    from instancediff_amd.utils.synthetic import make_batch
    dev = torch.device("cuda:0")
    model, sde = pipeline.build(phase="train", device=dev, T=100, seed=0, dist=False)
    model.set_train()
    sde.set_seed(1234)
    batch = make_batch(args.batch, args.size, seed=1234, mixed=True)
    for _ in range(2):
        model.feed_data(batch)
        model.optimize_parameters()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        model.feed_data(batch)
        model.optimize_parameters()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or not ev.kernels:
            continue
        where = "?"
        for fr in ev.stack or []:
            if "instancediff_amd" in fr and "site-packages" not in fr:
                where = fr.strip()
                break
        if where == "?" and ev.stack:
            where = " < ".join(x.strip()[-60:] for x in ev.stack[:4])
        k = (ev.name, where, tuple(ev.input_shapes[0]) if ev.input_shapes else ())
        agg[k][0] += 1
        agg[k][1] += sum(kk.duration for kk in ev.kernels)
    # device-to-device copies (hipMemcpyAsync -> __amd_rocclr_copyBuffer) by the op that issued them
    cp = collections.defaultdict(lambda: [0, 0.0])
    for ev in prof.events():
        ks = [k for k in (ev.kernels or []) if "copyBuffer" in k.name or "Memcpy" in k.name or "memcpy" in k.name]
        if ks:
            key = (ev.name, tuple(ev.input_shapes[0]) if ev.input_shapes else ())
            cp[key][0] += len(ks)
            cp[key][1] += sum(k.duration for k in ks)
    print("device-to-device copies by issuing op:")
    for (name, shp), (n, us) in sorted(cp.items(), key=lambda kv: -kv[1][0])[:25]:
        print("%5d x %-28s %9.1f us %s" % (n, name, us, list(shp)))
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    print("%d ATen ops with kernels; by (op, innermost package frame):" % sum(v[0] for v in agg.values()))
    for (name, where, shp), (n, us) in rows[:60]:
        print("%5d x %-14s %9.1f us %-22s %s" % (n, name, us, str(list(shp)), where[-240:]))


if __name__ == "__main__":
    main()
