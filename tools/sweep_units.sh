#!/bin/bash
# unit size / level sweep of the staged plans on one GPU box: tools/sweep_units.sh <workload> "<targets>" "<levels>"
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
W=$1; T=${2:-"6 8 10 12 14 16 20 24"}; L=${3:-"3"}
for l in $L; do for t in $T; do
  IQHIP_LEVELS=$l IQHIP_SPLIT=$t python bench.py --workload $W --steps 60 --warmup 20 --no-also --no-cpu-baseline --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$W levels $l target $t: kernel_ms_per_traversal %.4f launches %.0f ms/step %.4f' % (r['kernel_ms_per_traversal'], r['launches_per_traversal'], d['ms_per_step']))"
done; done
