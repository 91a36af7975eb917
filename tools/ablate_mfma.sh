#!/bin/bash
# timing-only ablations of the matrix-core traversal kernels (tools/build_alt.sh variants); results of the variants are wrong
# usage: tools/ablate_mfma.sh protein|codon  -> kernel ms per traversal for the shipped library and each lib_alt_* present
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
w=${1:-protein}
for d in lib lib_alt_nostore lib_alt_noload lib_alt_nomem lib_alt_nomfma; do
  [ -d iq-tree_amd/$d ] || continue
  IQHIP_LIB_DIR=$PWD/iq-tree_amd/$d python bench.py --workload $w --steps 100 --warmup 30 --no-also --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$d', 'kernel_ms_per_traversal', round(d['roofline']['kernel_ms_per_traversal'],4), 'ms/step', round(d['ms_per_step'],4))"
done
