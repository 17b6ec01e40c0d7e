"""Where does the batched-GEMM time of a training iteration go?  One single-stream iteration with train_ops.bgemm wrapped in events,
grouped by (M, N, K, batch, transA, transB).   python scripts/train_bgemm_probe.py [--batch 32]"""
import argparse
import collections
import os
import sys

os.environ["IDIFF_TRAIN_TWO_STREAMS"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    args = ap.parse_args()
    from instancediff_amd import pipeline, train_ops
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
    rec = []
    inner = train_ops.bgemm

    def timed(A, B, M, N, K, lda, ldb, transA, transB, sA, sB, batch, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = inner(A, B, M, N, K, lda, ldb, transA, transB, sA, sB, batch, *a, **k)
        e1.record()
        rec.append(((M, N, K, batch, bool(transA), bool(transB)), e0, e1))
        return r

    train_ops.bgemm = timed
    model.feed_data(batch)
    model.optimize_parameters()
    torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for k, e0, e1 in rec:
        agg[k][0] += 1
        agg[k][1] += e0.elapsed_time(e1) * 1e3
    tot = sum(v[1] for v in agg.values())
    print("%d bgemm calls, %.1f ms (event time, includes launch gaps)" % (len(rec), tot / 1e3))
    print("     M      N      K  batch tA tB   calls   total_us   avg_us   GB moved (A+B+C)  TB/s")
    for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        M, N, K, b, ta, tb = k
        gb = 4.0 * b * (M * K + K * N + M * N) / 1e9
        print("%6d %6d %6d %6d %2d %2d %7d %10.1f %8.1f %10.3f %12.2f" % (M, N, K, b, ta, tb, n, us, us / n, gb, gb / (us / n) * 1e6 / 1e3))


if __name__ == "__main__":
    main()
