#!/bin/bash
# protein traversal, leaf tables in LDS (IQHIP_LEAF_TABLES=1) vs products on the matrix pipe, with the LDS budget / parking variants
set -e -o pipefail
run() { # libdir env...
  d=$1; shift
  env IQHIP_LIB_DIR=$PWD/iq-tree_amd/$d "$@" python bench.py --workload protein --steps 100 --warmup 30 --no-also --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d $*', 'kernel_ms_per_traversal', round(d['roofline']['kernel_ms_per_traversal'],4), 'launches', d['roofline']['launches_per_traversal'], 'ms/step', round(d['ms_per_step'],4))"
}
for rep in 1 2; do
run lib IQHIP_LEAF_TABLES=0
run lib IQHIP_LEAF_TABLES=1
run lib IQHIP_LEAF_TABLES=1 IQHIP_HOLD_LDS=0
run lib IQHIP_LEAF_TABLES=1 IQHIP_HOLD_LDS=0 IQHIP_MFMA_LDS_KB=50
run lib IQHIP_LEAF_TABLES=1 IQHIP_TOP_CS2=0
run lib_alt_nomem IQHIP_LEAF_TABLES=1
run lib_alt_nostore IQHIP_LEAF_TABLES=1
done
