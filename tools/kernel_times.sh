#!/bin/bash
# per-kernel average times of one bench workload (rocprofv3 --kernel-trace --stats): tools/kernel_times.sh <workload> [ENV=val ...]
set -e -o pipefail
w=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/ktimes_$w
rm -rf $O && mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --steps 60 --warmup 10 --no-cpu-baseline --no-also > $O/bench.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) > 5:
            print("%-90s calls %6s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
grep -o '"ms_per_step": [0-9.]*' $O/bench.log | head -1
