#!/bin/bash
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
mkdir -p gpurun_out/r2f
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2f/gpu_tests.log 2>&1   # (a failing suite ends the sweep)
run() { # name env...
  n=$1; shift
  env "$@" python bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --sustain-seconds 0 > gpurun_out/r2f/${W}_$n.json 2> gpurun_out/r2f/${W}_$n.err
}
for W in codon protein; do
  run lt0 IQHIP_LEAF_TABLES=0 IQHIP_LEVELS=1
  run lt0_l3 IQHIP_LEAF_TABLES=0 IQHIP_LEVELS=3
  run l1 IQHIP_LEVELS=1
  run l2 IQHIP_LEVELS=2
  run l3 IQHIP_LEVELS=3
  run l4 IQHIP_LEVELS=4
  for t in 4 6 8 10; do run l3_t$t IQHIP_LEVELS=3 IQHIP_SPLIT=$t; run l5_t$t IQHIP_LEVELS=5 IQHIP_SPLIT=$t; done
done
IQHIP_DEBUG_PLAN=1 python bench.py --workload codon --steps 2 --warmup 1 --no-cpu-baseline --sustain-seconds 0 2>&1 | grep "iqhip\] plan" | sort | uniq -c | head -5
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2f/*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print("%-44s ms/step %.4f kern/trav %.4f launches/trav %.0f lnL %.6f"%(f.split("/")[-1],d["ms_per_step"],r["kernel_ms_per_traversal"],r["launches_per_traversal"],d["lnL"]))
    except Exception as e: print(f, "ERR", e)
PY
