#!/usr/bin/env python3
"""Per-kernel HBM traffic of ONE steady-state denoising step: joins two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- separate runs,
one counter each, as MI355X_MICROARCH.md's HBM section prescribes) with the durations of a counter-free kernel trace of the same command.

    IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d T -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline
    IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d F -- python3 bench.py ... (same)
    IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d W -- python3 bench.py ... (same)
    python3 scripts/pmc_table.py T F W out.json > out.txt

The step is cut as in step_trace.py (between the last two drift_step_dev_kernel launches); in each pass kernels are matched by name and
averaged over the step's launches of that name (the launch sequence of a step is deterministic).  Units / gfx950 correction per the
guide: FETCH_SIZE and WRITE_SIZE are KB; FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so read bytes =
2 x FETCH_SIZE (exact for 16-B-per-lane streaming reads, an upper bound for narrow ones); WRITE_SIZE is exact.  Infinity-Cache hits are
counted as traffic.  GB/s = bytes / duration of the counter-free pass; frac = GB/s / 8000 (HBM3E peak; ~6300 achievable)."""
import csv
import glob
import json
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.step_trace import short  # noqa: E402

HBM_PEAK_GBS = 8000.0


def load(d, pattern):
    files = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    if not files:
        raise SystemExit(f"no {pattern} under {d}")
    rows = []
    for path in files:
        with open(path, newline="") as f:
            rows += list(csv.DictReader(f))
    return rows


def step_window(trace_rows):
    """(start, end) timestamps of the last full step and its rows"""
    rows = sorted(trace_rows, key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "drift_step_dev_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit("need at least two denoising steps in the trace")
    return rows[marks[-2] + 1: marks[-1] + 1]


def counter_by_kernel(d, counter):
    """kernel short name -> mean counter value per launch, over the launches of the last step of that pass"""
    step = step_window(load(d, "*kernel_trace.csv"))
    ids = {r["Dispatch_Id"] for r in step}
    acc = {}
    for r in load(d, "*counter_collection.csv"):
        if r["Counter_Name"] == counter and r["Dispatch_Id"] in ids:
            a = acc.setdefault(short(r["Kernel_Name"]), [0.0, set()])
            a[0] += float(r["Counter_Value"])   # one row per (dispatch, counter instance): summed
            a[1].add(r["Dispatch_Id"])
    return {k: v[0] / len(v[1]) for k, v in acc.items()}


def main():
    tdir, fdir, wdir, out = sys.argv[1:5]
    step = step_window(load(tdir, "*kernel_trace.csv"))
    dur = OrderedDict()
    for r in step:
        c = dur.setdefault(short(r["Kernel_Name"]), [0, 0])
        c[0] += 1
        c[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    fetch = counter_by_kernel(fdir, "FETCH_SIZE")
    write = counter_by_kernel(wdir, "WRITE_SIZE")
    tot_ns = sum(v[1] for v in dur.values())
    recs, step_bytes = [], 0.0
    for k, (n, ns) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        fk, wk = fetch.get(k), write.get(k)
        rec = {"kernel": k, "launches_per_step": n, "avg_us": round(ns / n / 1e3, 2), "share_of_step_kernel_time": round(ns / tot_ns, 4)}
        if fk is not None and wk is not None:
            b = (2 * fk + wk) * 1024.0
            step_bytes += b * n
            gbs = b / (ns / n)  # bytes per ns = GB/s
            rec.update(bound="hbm", FETCH_SIZE_KB=round(fk, 1), WRITE_SIZE_KB=round(wk, 1), hbm_bytes_per_launch=int(b), achieved_GBps=round(gbs, 1),
                       peak_GBps=HBM_PEAK_GBS, frac=round(gbs / HBM_PEAK_GBS, 4))
        recs.append(rec)
    from bench import kernel_source_hash
    doc = {"formula": "hbm bytes = 2*FETCH_SIZE + WRITE_SIZE (KB; gfx950: 128-B read requests are tallied at 64 B); GB/s over the duration of a "
                      "counter-free trace of the same single-stream eager step",
           "step_launches": len(step), "step_kernel_ms": round(tot_ns / 1e6, 3), "step_hbm_bytes": int(step_bytes),
           "step_hbm_GBps_over_kernel_time": round(step_bytes / tot_ns, 1), "kernel_source_hash": kernel_source_hash(), "kernels": recs}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(f"steady-state step: {len(step)} launches, {tot_ns / 1e6:.3f} ms of kernel time, {step_bytes / 1e9:.2f} GB of HBM traffic "
          f"({step_bytes / tot_ns:.0f} GB/s over the kernel time)")
    print("  %-66s %4s %9s %6s %10s %10s %9s %6s" % ("kernel", "n", "avg_us", "share", "read_MB", "write_MB", "GB/s", "frac"))
    for r in recs[:40]:
        if "frac" in r:
            print("  %-66s %4d %9.1f %6.3f %10.1f %10.1f %9.0f %6.3f" % (r["kernel"][:66], r["launches_per_step"], r["avg_us"], r["share_of_step_kernel_time"],
                                                                        2 * r["FETCH_SIZE_KB"] / 1024, r["WRITE_SIZE_KB"] / 1024, r["achieved_GBps"], r["frac"]))
        else:
            print("  %-66s %4d %9.1f %6.3f" % (r["kernel"][:66], r["launches_per_step"], r["avg_us"], r["share_of_step_kernel_time"]))


if __name__ == "__main__":
    main()
