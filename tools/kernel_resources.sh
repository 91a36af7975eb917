#!/bin/bash
# per-kernel register / scratch / occupancy summary of one .hip file (compiler view, gfx950)
# usage: tools/kernel_resources.sh iq-tree_amd/csrc/kernels_mfma.hip [extra hipcc flags]
set -e -o pipefail   # stop at the first failing step: a faulting kernel must not be followed by more runs on the box
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
python3 -c '
import re,sys,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"remark: (?:\s*)(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k,v=m.groups()
    if k=="Function Name":
        cur={"name":v}; rows.append(cur)
    elif cur is not None: cur[k]=v
for r in rows:
    try: name=subprocess.check_output(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt",r["name"]]).decode().strip()
    except Exception: name=r["name"]
    name=re.sub(r"\(.*","",name)
    print("%-62s vgpr %-4s agpr %-4s sgpr %-4s scratch %-5s occ %-2s spillV %s" % (name[:62], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill")))
'
