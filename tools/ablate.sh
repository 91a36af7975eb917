#!/bin/bash
# timing-only ablations of the traversal kernel (results are wrong when IQHIP_ABLATE != 0)
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
for ab in 0 1 2 3; do IQHIP_ABLATE=$ab python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ablate', $ab, 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'ms/step', round(d['ms_per_step'],4))"; done
