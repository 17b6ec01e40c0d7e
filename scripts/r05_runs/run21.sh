#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run21; mkdir -p $O
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES SQ_BUSY_CU_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"; do
  d=$O/$(echo $c | cut -c1-40 | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 scripts/r05_runs/xattn_l0.py > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  python3 - "$d" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "smm_xattn_grouped" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({kk: round(a[0] / max(a[1], 1)) for kk, a in acc.items()}, flush=True)
PY
done 2>&1 | tee $O/pmc_xattn_l0.txt
