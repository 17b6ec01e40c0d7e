#!/bin/bash
# collect_mfma_util.sh OUT.json: matrix-pipe utilisation of conv_wino4_kernel on the step's main layer shapes,
#   SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), one rocprofv3 --pmc pass per shape (counters only: --kernel-trace, no other trace domain)
OUT=${1:-gpurun_out/r04/mfma_util.json}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/mfma_util
python3 - <<'PY' > gpurun_out/mfma_util/hash.txt
import sys; sys.path.insert(0, ".")
import bench; print(bench.kernel_source_hash())
PY
i=0
for F in "L0 64->64 3x3 +stats" "L0 64->64 3x3 +stats+pro" "L0up 144->64 3x3 +stats" "L2up 416->256 3x3 +stats" "L1 64->64 3x3 +stats+pro" "up 128->64 3x3 upsample"; do
  d=gpurun_out/mfma_util/s$i; i=$((i+1))
  rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $d -- python3 scripts/conv_bench.py --only "$F" --rounds 1 --iters 2 --algos 3 > $d.log 2>&1
  echo "$F" > $d.name
done
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = {"kernel": "conv_wino4_kernel", "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)", "kernel_source_hash": open("gpurun_out/mfma_util/hash.txt").read().strip(), "shapes": {}}
for nf in sorted(glob.glob("gpurun_out/mfma_util/s*.name")):
    d = nf[:-5]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_wino4_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    if "SQ_BUSY_CU_CYCLES" in acc and acc["SQ_BUSY_CU_CYCLES"][0] > 0:
        busy, mf = acc["SQ_BUSY_CU_CYCLES"][0], acc["SQ_VALU_MFMA_BUSY_CYCLES"][0]
        out["shapes"][open(nf).read().strip()] = {"mfma_util": round(mf / (4.0 * busy), 4), "launches": acc["SQ_BUSY_CU_CYCLES"][1]}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/mfma_util
