#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_ab
for v in old scal; do
  for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
    d=gpurun_out/pmc_ab/${v}_$(echo $c | tr ' ' '_')
    IDIFF_LIB=$GRAFT_REPO_ROOT/instancediff_amd/variants/libidiff_$v.so rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 scripts/conv_bench.py --only "L2up" --rounds 1 --iters 2 > $d.log 2>&1
    python3 - "$d" "$v" "$c" <<'PY'
import csv, glob, sys, collections
d, v, c = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wino_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(v, {k: round(a[0] / max(a[1], 1)) for k, a in acc.items()}, flush=True)
PY
  done
done
