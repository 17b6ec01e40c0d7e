#!/usr/bin/env python3
"""Steady-state view of ONE denoising step from a rocprofv3 kernel trace of bench.py (excludes warm-up, weight preparation, graph
capture): the kernels between the last two `drift_step_dev_kernel` launches.

    IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- \
        python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline
    python3 scripts/step_trace.py gpurun_out/trace [out.csv]

Prints launches per step, kernel time per step by kernel name, the share of the dominant kernel, and every launch that is not one
of this library's kernels (ATen `at::native::*`, `__amd_rocclr_*` copies/fills, rocBLAS `Cijk_*`) -- the steady-state step should
show none."""
import csv
import glob
import os
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("idiff_detail::", "")
    return re.sub(r"\(.*$", "", name)[:70]


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
    if len(marks) < 2:
        raise SystemExit("need at least two denoising steps in the trace")
    step = rows[marks[-2] + 1: marks[-1] + 1]
    # state advance follows the update: count it with the step it closes
    span_ns = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
    by = OrderedDict()
    for r in step:
        k = short(r["Kernel_Name"])
        t = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c = by.setdefault(k, [0, 0])
        c[0] += 1
        c[1] += t
    tot = sum(v[1] for v in by.values())
    foreign = {k: v for k, v in by.items() if k.startswith("at::") or k.startswith("__amd_rocclr") or k.startswith("Cijk_") or "rocclr" in k}
    wino = sum(v[1] for k, v in by.items() if k.startswith("conv_wino"))
    print(f"steady-state step: {len(step)} launches, {tot / 1e6:.3f} ms of kernel time, {span_ns / 1e6:.3f} ms first-start to last-end")
    print(f"  3x3 Winograd convs (conv_wino4_kernel + conv_wino4h_kernel + conv_wino_kernel) {wino / 1e6:.3f} ms ({100.0 * wino / tot:.1f} %), everything else {(tot - wino) / 1e6:.3f} ms")
    print(f"  launches that are not this library's kernels: {sum(v[0] for v in foreign.values())} {dict((k, v[0]) for k, v in foreign.items())}")
    lines = [("kernel", "launches_per_step", "total_us", "avg_us", "share_pct")]
    for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        lines.append((k, n, f"{t / 1e3:.1f}", f"{t / 1e3 / n:.2f}", f"{100.0 * t / tot:.2f}"))
    for ln in lines[:40]:
        print("  %-72s %6s %10s %9s %7s" % ln)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w", newline="") as f:
            w = csv.writer(f)
            w.writerows(lines)
            w.writerow(("TOTAL", len(step), f"{tot / 1e3:.1f}", "", "100"))


if __name__ == "__main__":
    main()
