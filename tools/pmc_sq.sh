#!/bin/bash
# SQ counters of the traversal kernel (one pass, <= 8 SQ counters + GRBM); usage: tools/pmc_codon.sh <workload> <outdir>
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
W=${1:-codon}; OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/pmc_$W}
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc ${PMC:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE} --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --sustain-seconds 0 > $OUT/run.log 2>&1
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
d=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][-50:], r["Grid_Size"])
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "traverse" not in k[0]: continue
    print(k)
    for c,vals in sorted(v.items()):
        t=vals[len(vals)//3:]
        print("   %-32s %.4g"%(c,sum(t)/len(t)))
PY
