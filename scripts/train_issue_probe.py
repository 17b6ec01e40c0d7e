"""How long does the host take to ISSUE one training iteration, against how long the GPU takes to run it?
    python scripts/train_issue_probe.py [--batch 32] [--steps 4]
Prints per iteration: host time until forward_backward_inputRes returned (everything enqueued, nothing waited for), host time
until the Adam steps were enqueued, and the time at which the device had finished.  If the first two are close to the third the
iteration is bound by the host's launch rate, not by the kernels."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--steps", type=int, default=4)
    args = ap.parse_args()
    from instancediff_amd import pipeline, train_ops
    from instancediff_amd.utils.synthetic import make_batch
    dev = torch.device("cuda:0")
    model, sde = pipeline.build(phase="train", device=dev, T=100, seed=0, dist=False)
    model.set_train()
    sde.set_seed(1234)
    batch = make_batch(args.batch, args.size, seed=1234, mixed=True)
    torch.manual_seed(99)
    marks = {}
    inner = train_ops.forward_backward_inputRes

    def wrapped(m):
        marks["fb0"] = time.perf_counter()
        r = inner(m)
        marks["fb1"] = time.perf_counter()
        marks["fwd"] = r[1]
        return r

    train_ops.forward_backward_inputRes = wrapped
    bw = torch.autograd.backward

    def bw_marked(*a, **k):
        r = bw(*a, **k)
        marks.setdefault("bw", []).append(time.perf_counter())
        return r

    torch.autograd.backward = bw_marked
    cpu = torch.Tensor.cpu

    def cpu_marked(self, *a, **k):
        if self.numel() == 10 and "rec" not in marks:
            marks["rec"] = time.perf_counter()
        return cpu(self, *a, **k)

    torch.Tensor.cpu = cpu_marked
    for i in range(2 + args.steps):
        torch.cuda.synchronize()
        marks.clear()
        t0 = time.perf_counter()
        model.feed_data(batch)
        model.optimize_parameters()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if i >= 2:
            print("   backward walks returned at %s ms" % ", ".join("%.1f" % ((x - t0) * 1e3) for x in marks.get("bw", [])))
            print("iteration %d: feed_data %.1f ms | forwards issued %.1f | forward+backward issued %.1f | Adam issued %.1f | device done %.1f"
                  % (i - 2, (marks["fb0"] - t0) * 1e3, (marks["fb0"] - t0 + marks["fwd"]) * 1e3, (marks["fb1"] - t0) * 1e3,
                     (marks.get("rec", t1) - t0) * 1e3, (t1 - t0) * 1e3), flush=True)


if __name__ == "__main__":
    main()
