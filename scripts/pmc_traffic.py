#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes of bench.py into profiles/<round>/pmc_traffic.json -- HBM bytes per launch of the dominant kernel,
the `roofline.traffic` of the bench line.  Run on the GPU box, from the repo root, after

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    python3 scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic.json [kernel-name-substring]

(separate passes: one counter per run, as MI355X_MICROARCH.md's HBM section prescribes).  Units and gfx950 correction per that guide:
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B -> x2; WRITE_SIZE is exact.
The file records a hash of the kernel sources (bench.kernel_source_hash): bench.py refuses to print a traffic figure measured on
other sources."""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_launch(dirname, counter, needle):
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {dirname}")
    tot, n, names = 0.0, 0, set()
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and needle in row["Kernel_Name"]:
                    tot += float(row["Counter_Value"])
                    n += 1
                    names.add(row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0][:80])
    if n == 0:
        raise SystemExit(f"no {counter} rows for kernels matching {needle!r} in {files}")
    return tot / n, n, sorted(names)


def main():
    fdir, wdir, out = sys.argv[1:4]
    needle = sys.argv[4] if len(sys.argv) > 4 else "conv_wino4_kernel"
    from bench import kernel_source_hash
    fetch_kb, nf, names = per_launch(fdir, "FETCH_SIZE", needle)
    write_kb, nw, _ = per_launch(wdir, "WRITE_SIZE", needle)
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    rec = {"kernel": needle, "kernel_instances": names, "launches_averaged": [nf, nw],
           "FETCH_SIZE_KB_per_launch": round(fetch_kb, 2), "WRITE_SIZE_KB_per_launch": round(write_kb, 2),
           "formula": "2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests of wide coalesced reads at 64 B)",
           "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
           "kernel_source_hash": kernel_source_hash(), "commit": commit or os.environ.get("IDIFF_COMMIT"),
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
                      "--no-cpu-baseline --no-roofline (two passes)"}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
