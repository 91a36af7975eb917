#!/bin/bash
# A/B of library builds on ONE GPU box: tools/ab.sh <workload> <dir> [<dir> ...]  (dirs under iq-tree_amd/, e.g. lib lib_alt_base)
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
w=$1; shift
for rep in 1 2; do
for d in "$@"; do
  IQHIP_LIB_DIR=$PWD/iq-tree_amd/$d python bench.py --workload $w --steps 100 --warmup 30 --no-also --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', 'kernel_ms_per_traversal', round(d['roofline']['kernel_ms_per_traversal'],4), 'ms/step', round(d['ms_per_step'],4))"
done
done
