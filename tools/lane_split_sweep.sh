#!/bin/bash
# 4-state traversal: one lane per pattern vs two (IQHIP_LANE_SPLIT), the latter also with three workgroups per CU (IQHIP_LDS_KB=52)
set -e -o pipefail
run() { # patterns env...
  P=$1; shift
  env "$@" python bench.py --workload dna --patterns $P --steps 100 --warmup 20 --no-also --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print($P, '$*', 'kernel_ms', round(d['roofline']['kernel_ms_per_traversal'],4), 'ms/step', round(d['ms_per_step'],4))"
}
for P in 66000 80000 90000 98000 100000; do
  run $P IQHIP_LANE_SPLIT=1
  run $P IQHIP_LANE_SPLIT=2 IQHIP_LDS_KB=52
  run $P IQHIP_LANE_SPLIT=2 IQHIP_LDS_KB=40
done
