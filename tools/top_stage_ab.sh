#!/bin/bash
# per-kernel times of the protein traversal under run-time variants of its top stage, one box: tools/top_stage_ab.sh "<ENV=val ...>" ...
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
O=$GRAFT_REPO_ROOT/gpurun_out/top_ab
mkdir -p $O
i=0
for spec in "$@"; do
  i=$((i+1))
  for kv in $spec; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/run$i -- python3 $B --workload protein --steps 30 --warmup 5 --no-cpu-baseline --no-also > $O/run$i.log 2>&1
  for kv in $spec; do unset "${kv%%=*}"; done
  echo "== $spec"
  for f in $(find $O/run$i -name "*kernel_stats.csv"); do grep "k_traverse" $f | cut -d, -f1-4 | cut -c1-110; done
done
