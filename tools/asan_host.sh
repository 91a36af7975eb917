#!/bin/bash
# AddressSanitizer + UBSan run of the host-side C++ (mirror, alignment reader, model producers) on the CPU:
# builds an instrumented libiqhost.so next to the regular libiqhip.so and runs the CPU test modules that
# exercise it.  (GPU sanitizers are not available on the pool; the kernels are covered by the parity tests.)
set -e
cd "$(dirname "$0")/.."
D=/tmp/iqhip_asan
mkdir -p $D && cp iq-tree_amd/lib/libiqhip.so $D/
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o $D/libiqhost.so \
    iq-tree_amd/host/phylo_host.cpp iq-tree_amd/host/iqhost_c.cpp iq-tree_amd/host/model_host.cpp \
    iq-tree_amd/host/alignment_host.cpp iq-tree_amd/host/iqmodel_c.cpp -L$D -liqhip -Wl,-rpath,$D
IQHIP_LIB_DIR=$D LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)" \
    ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_producers.py tests/test_host_logic.py tests/test_abi.py -q
