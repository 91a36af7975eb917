#!/bin/bash
# kernel timeline of one optimizeAllBranches sweep (device Newton), DNA 50 x 100k
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_bo_${1:-100000}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bo.py <<'PY'
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
pkg = g.load_package(); synth = importlib.import_module("iqtree_amd.synth")
model = synth.gtr_model()
nwk, pat, freq = synth.make_workload(50, int(sys.argv[1]), model, seed=3)
t = pkg.PhyloTree(nwk); t.set_alignment(4, 0, pat, freq); t.set_model(model); t.attach_engine(0)
t.clear_all_partial_lh(); t.compute_likelihood()
t.optimize_all_branches(iterations=1, tolerance=1e-3)   # (one engine submission per sweep: iqhip_optimize_sweep)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/bo.py ${1:-100000} > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for f in glob.glob(out + "/**/*_kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("%-60s calls %5s avg %9.1f us total %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
for f in glob.glob(out + "/**/*_kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 2: len(rows) // 2 + 16]
    t0 = int(rows[0]["Start_Timestamp"]); pe = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print("%-44s start %8.1f dur %7.1f gap %6.1f" % (r["Kernel_Name"][:44], s / 1e3, (e - s) / 1e3, 0 if pe is None else (s - pe) / 1e3))
        pe = e
PY
