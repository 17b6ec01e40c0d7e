#!/bin/bash
# train_mfma_util.sh OUT.txt: matrix-pipe utilisation of the training iteration's kernels,
#   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) per kernel name, one rocprofv3 --pmc pass (counters + --kernel-trace only)
OUT=${1:-gpurun_out/r04/train_mfma_util.txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/tmu $(dirname $OUT)
export IDIFF_TRAIN_TWO_STREAMS=0
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/tmu/p -- python3 bench.py --mode train --batch 32 --steps 1 --warmup 1 > gpurun_out/tmu/p.log 2>&1; tail -3 gpurun_out/tmu/p.log | cut -c1-200
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob("gpurun_out/tmu/p/**/*counter_collection.csv", recursive=True):
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
with open(sys.argv[1], "w") as o:
    o.write("matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES), two training iterations (one warm-up), one stream, batch 32\n")
    o.write("%-62s %8s %10s\n" % ("kernel (matrix-core kernels only, by busy cycles)", "launches", "mfma_util"))
    for busy, k, cnt, u in rows[:24]:
        o.write("%-62s %8d %10.3f\n" % (k, cnt, u))
print(open(sys.argv[1]).read())
PY
rm -rf gpurun_out/tmu
