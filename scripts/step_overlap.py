#!/usr/bin/env python3
"""Where does the wall time of ONE two-stream denoising step go?  From a rocprofv3 kernel trace of the timed configuration (HIP graph,
two streams): the interval between the last two `drift_step_dev_kernel` launches is cut at every kernel start / end and each piece is
classed by what was running: nothing (dispatch gaps), one kernel, two or more.  For the pieces with exactly one kernel the time is
listed by kernel: that is the time a kernel spent ALONE on the GPU (its partner stream had nothing to overlap with it).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace2 -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline \
        --no-roofline --no-train-leg
    python3 scripts/step_overlap.py gpurun_out/trace2"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("idiff_detail::", "")
    return re.sub(r"\(.*$", "", name)[:60]


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    for path in files:
        with open(path, newline="") as f:
            rows += list(csv.DictReader(f))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "drift_step_dev_kernel" in r["Kernel_Name"]]
    if len(marks) < 3:
        raise SystemExit("need at least three denoising steps in the trace")
    t0 = int(rows[marks[-2]]["End_Timestamp"])
    t1 = int(rows[marks[-1]]["End_Timestamp"])
    step = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and int(r["End_Timestamp"]) <= t1]
    ev = []
    for i, r in enumerate(step):
        ev.append((int(r["Start_Timestamp"]), 1, i))
        ev.append((int(r["End_Timestamp"]), 0, i))
    ev.sort()
    live = set()
    by_n = defaultdict(int)
    alone = defaultdict(int)
    pair = defaultdict(int)
    prev = t0
    for t, kind, i in ev:
        dt = t - prev
        if dt > 0:
            n = len(live)
            by_n[min(n, 3)] += dt
            if n == 1:
                alone[short(step[next(iter(live))]["Kernel_Name"])] += dt
            elif n == 2:
                a, b = sorted(short(step[j]["Kernel_Name"]).split("<")[0] for j in live)
                pair[(a, b)] += dt
        prev = t
        if kind:
            live.add(i)
        else:
            live.discard(i)
    by_n[0] += t1 - prev
    wall = t1 - t0
    ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
    print(f"one step of the timed configuration: {len(step)} launches, wall {wall / 1e6:.3f} ms, sum of kernel durations {ksum / 1e6:.3f} ms")
    for n, label in ((0, "no kernel running (dispatch gaps)"), (1, "exactly one kernel"), (2, "two kernels"), (3, "three or more")):
        print(f"  {label:36s} {by_n[n] / 1e6:8.3f} ms  {100.0 * by_n[n] / wall:5.1f} %")
    print("  time ALONE on the GPU, by kernel:")
    for k, t in sorted(alone.items(), key=lambda kv: -kv[1])[:14]:
        print(f"    {k:62s} {t / 1e6:8.3f} ms")
    print("  time in PAIRS, by the two kernels:")
    for (a, b), t in sorted(pair.items(), key=lambda kv: -kv[1])[:12]:
        print(f"    {a:34s} + {b:34s} {t / 1e6:8.3f} ms")


if __name__ == "__main__":
    main()
