#!/bin/bash
# A/B of run-time switches on ONE GPU box: tools/ab_env.sh <workload> "<libdir> [ENV=val ...]" ["<libdir> ..." ...]
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
w=$1; shift
run() { # libdir env...
  d=$1; shift
  env IQHIP_LIB_DIR=$PWD/iq-tree_amd/$d "$@" python bench.py --workload $w --steps 100 --warmup 30 --no-also --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d $*', 'kernel_ms_per_traversal', round(d['roofline']['kernel_ms_per_traversal'],4), 'launches', d['roofline']['launches_per_traversal'], 'ms/step', round(d['ms_per_step'],4))"
}
for rep in 1 2; do
  for spec in "$@"; do run $spec; done
done
