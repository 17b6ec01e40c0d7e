#!/bin/bash
# train_profile.sh OUTDIR [BATCH]: rocprofv3 kernel stats of the training step (bench.py --mode train), summary CSV kept
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/train_prof}; B=${2:-32}; mkdir -p $O
python3 bench.py --mode train --batch $B --steps 4 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err; cat $O/bench_train.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --mode train --batch $B --steps 3 --warmup 2 --no-roofline > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/train_kernel_stats.csv; rm -rf $O/stats
python3 - "$O/train_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6/5:.1f} ms per iteration (5 iterations traced)")
for r in rows[:32]:
    name = re.sub(r"\(.*$", "", r["Name"].replace("void ", "").replace("(anonymous namespace)::", ""))[:70]
    print(f"  {name:70s} {int(r['Calls'])//5:5d}/it {float(r['TotalDurationNs'])/5e6:8.2f} ms/it {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f}%")
PY
