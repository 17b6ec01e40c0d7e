#!/bin/bash
# what differs between the shipped conv_wino4_kernel and a build shifted by 12 bytes behind a 64-byte anchor (w4al3: +0.55 ms/step): SQ / SQC counters on the
# two-source 144 -> 64 layer at 256^2 (the shape that moves most), separate --pmc passes (counters + --kernel-trace only)
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05/run31; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQC\?_[A-Z0-9_]*" | sort -u | grep -i "ICACHE\|IFETCH\|INST_LEVEL\|INSTS_SALU\|INSTS_VALU\|BRANCH\|WAIT_INST\|BUSY\|WAVE_CYCLES\|INSTS_SMEM\|DCACHE" | tr '\n' ' ' > $O/counters_available.txt; cat $O/counters_available.txt; echo
for lib in tree w4al3; do
  if [ $lib = tree ]; then unset IDIFF_LIB; else export IDIFF_LIB=$PWD/instancediff_amd/variants/libidiff_$lib.so; fi
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_INSTS_VALU"; do
    d=$O/pmc_$lib; rm -rf $d
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- python3 scripts/conv_bench.py --only "L0up 144" --rounds 1 --iters 3 --algos 3 > $d.log 2>&1 || { tail -3 $d.log; }
    python3 - "$d" "$lib" <<'PY'
import csv, glob, sys, collections
d, lib = sys.argv[1:3]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wino4_kernel" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
print(lib, {kk: round(a[0] / max(a[1], 1)) for kk, a in sorted(acc.items())}, flush=True)
PY
  done
done 2>&1 | tee $O/pmc_placement.txt
