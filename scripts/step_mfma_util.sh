#!/bin/bash
# step_mfma_util.sh OUT.txt: matrix-pipe utilisation of every matrix-core kernel of the SAMPLING step (c2: 256^2, batch 16),
#   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) per kernel name, one rocprofv3 --pmc pass (counters + --kernel-trace only),
#   single stream, eager (the counters are per launch: overlap of the two nets' streams would mix their kernels' busy cycles)
OUT=${1:-gpurun_out/r05/step_mfma_util.txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/smu $(dirname $OUT) && rm -rf gpurun_out/smu/p
export IDIFF_HIP_GRAPH=0 IDIFF_TWO_STREAMS=0
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/smu/p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-train-leg > gpurun_out/smu/p.log 2>&1; tail -1 gpurun_out/smu/p.log | cut -c1-160
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob("gpurun_out/smu/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
            n[k] += 1
rows = []
for k, c in acc.items():
    busy, mf = c.get("SQ_BUSY_CU_CYCLES", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if busy > 0 and mf > 0:
        rows.append((busy, k, n[k], mf / (4.0 * busy)))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
with open(sys.argv[1], "w") as o:
    o.write("matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES) of the sampling step's matrix-core kernels (3 steps incl. warm-up + graph-free setup, one stream, 256^2 batch 16)\n")
    o.write("(busy cycles count every matrix instruction at its own rate: for the bf16x3 1x1 convs and the f16 attention variant the figure is the bf16 / f16 pipe's occupancy)\n")
    o.write("%-62s %8s %10s %12s\n" % ("kernel (by busy CU cycles)", "launches", "mfma_util", "share_busy"))
    for busy, k, cnt, u in rows[:28]:
        o.write("%-62s %8d %10.3f %12.3f\n" % (k, cnt, u, busy / tot))
print(open(sys.argv[1]).read())
PY
rm -rf gpurun_out/smu
