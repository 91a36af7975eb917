#!/bin/bash
# One rocprofv3 --pmc pass over a bench workload (GPU box).  usage: bash tools/pmc_pass.sh <tag> <workload> "<counters>" [extra bench args]
set -e
TAG=$1; W=$2; CTR=$3; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline "$@" > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-30s %.4g per launch (%d launches)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
