#!/bin/bash
# timing-only variant of the library: tools/build_alt.sh <name> <hipcc flags for kernels_mfma.hip / kernels_valu4.hip ...>
# -> iq-tree_amd/lib_alt_<name>/ (git-ignored), selected at run time with IQHIP_LIB_DIR.  Results of such a build may be wrong.
set -e
name=$1; shift
cd "$(dirname "$0")/.."
d=iq-tree_amd/lib_alt_$name
mkdir -p $d
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-value -Wno-unused-result -Iinclude"
for f in engine kernels_valu4 kernels_mfma kernels_newton kernels_sweep kernels_rell comm sharded; do
  case $f in
    kernels_mfma|kernels_valu4) /opt/rocm/bin/hipcc $FLAGS "$@" -c iq-tree_amd/csrc/$f.hip -o $d/$f.o & ;;
    *) cp iq-tree_amd/lib/$f.o $d/$f.o ;;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libiqhip.so $d/*.o -ldl
g++ -O2 -std=c++17 -fPIC -shared -o $d/libiqhost.so iq-tree_amd/host/phylo_host.cpp iq-tree_amd/host/iqhost_c.cpp iq-tree_amd/host/model_host.cpp iq-tree_amd/host/alignment_host.cpp iq-tree_amd/host/iqmodel_c.cpp -Iinclude -L$d -liqhip -Wl,-rpath,'$ORIGIN'
echo built $d
