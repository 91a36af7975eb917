#!/bin/bash
# Run on the GPU box (gpurun): bench + rocprofv3 kernel trace + PMC traffic passes for the three
# workloads.  Writes everything under gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
# usage: bash tools/profile_round.sh <tag>
set -e
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
python3 $B --workload dna --steps 200 --warmup 20 > $OUT/bench_dna.json 2> $OUT/bench_dna.err
python3 $B --workload protein --steps 100 --warmup 20 --cpu-seconds 10 > $OUT/bench_protein.json 2> $OUT/bench_protein.err
python3 $B --workload codon --steps 200 --warmup 40 --cpu-seconds 10 > $OUT/bench_codon.json 2> $OUT/bench_codon.err
python3 $B > $OUT/bench_default.json 2> $OUT/bench_default.err   # the driver's command: headline + configs 2..4 + branchopt
for w in dna protein codon; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $B --workload $w --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_$w.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$w -- python3 $B --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $OUT/fetch_$w.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write_$w -- python3 $B --workload $w --steps 6 --warmup 2 --no-cpu-baseline > $OUT/write_$w.log 2>&1
done
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py $OUT > $OUT/summary.json
cat $OUT/summary.json
