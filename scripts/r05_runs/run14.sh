#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sed -n '/^cat > \/tmp\/inv.py/,/^PY$/p' scripts/r05_runs/run11_invariance.sh | sed '1d;$d' > /tmp/inv.py
for cfg in "IDIFF_TWO_STREAMS=0" "IDIFF_TWO_STREAMS=0 IDIFF_HIP_GRAPH=0" "IDIFF_HIP_GRAPH=0 AMD_SERIALIZE_KERNEL=3" "IDIFF_SELECT_STRIPS=0"; do
  echo "== [$cfg]"; env $cfg python3 /tmp/inv.py 2>&1 | tail -1
done
