#!/bin/bash
# A/B of an environment switch on ONE GPU box: tools/ab_env.sh <workload> VAR=a VAR=b ...
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
w=$1; shift
for rep in 1 2; do
for kv in "$@"; do
  env $kv python bench.py --workload $w --steps 100 --warmup 30 --no-also --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kv', 'kernel_ms_per_traversal', round(d['roofline']['kernel_ms_per_traversal'],4), 'ms/step', round(d['ms_per_step'],4))"
done
done
