#!/bin/bash
# pmc_kernel.sh KERNEL_SUBSTRING BENCH_FILTER [LIB]: SQ counters of one kernel on one conv_bench shape (separate --pmc passes)
K=$1; F=$2; LIB=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_k
[ -n "$LIB" ] && export IDIFF_LIB=$GRAFT_REPO_ROOT/$LIB
for c in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  d=gpurun_out/pmc_k/$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 scripts/conv_bench.py --only "$F" --rounds 1 --iters 2 ${ALGOS:+--algos $ALGOS} > $d.log 2>&1
  python3 - "$d" "$K" <<'PY'
import csv, glob, sys, collections
d, k = sys.argv[1:3]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print({kk: round(a[0] / max(a[1], 1)) for kk, a in acc.items()}, flush=True)
PY
done
