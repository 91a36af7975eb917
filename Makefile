# Build of the MI355X likelihood engine (gfx950 only), its host mirror and the CPU oracle.
#   make            -> everything that ships to the GPU box
#   make ref        -> oracle/_ref (needs /root/reference; container only)
HIPCC      ?= /opt/rocm/bin/hipcc
CXX        ?= g++
CC         ?= gcc
ARCH       := gfx950
PKG        := iq-tree_amd
CSRC       := $(PKG)/csrc
HOST       := $(PKG)/host
LIBDIR     := $(PKG)/lib
HIPFLAGS   := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
HIP_SRCS   := $(CSRC)/engine.hip $(CSRC)/kernels_valu4.hip $(CSRC)/kernels_mfma.hip $(CSRC)/kernels_newton.hip $(CSRC)/kernels_sweep.hip $(CSRC)/kernels_rell.hip $(CSRC)/comm.hip $(CSRC)/sharded.hip
HIP_OBJS   := $(patsubst $(CSRC)/%.hip,$(LIBDIR)/%.o,$(HIP_SRCS))

all: $(LIBDIR)/libiqhip.so $(LIBDIR)/libiqhost.so $(LIBDIR)/iqhip_lnl oracle/liblh_oracle.so

$(LIBDIR):
	mkdir -p $(LIBDIR)

$(LIBDIR)/%.o: $(CSRC)/%.hip $(CSRC)/iqhip_internal.h include/iqhip.h | $(LIBDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIBDIR)/libiqhip.so: $(HIP_OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(HIP_OBJS) -ldl

HOST_SRCS  := $(HOST)/phylo_host.cpp $(HOST)/iqhost_c.cpp $(HOST)/model_host.cpp $(HOST)/alignment_host.cpp $(HOST)/iqmodel_c.cpp
HOST_HDRS  := $(HOST)/phylo_host.h $(HOST)/model_host.h $(HOST)/alignment_host.h include/iqhip.h include/iqhip_adapter.h

$(LIBDIR)/libiqhost.so: $(HOST_SRCS) $(HOST_HDRS) $(LIBDIR)/libiqhip.so
	$(CXX) -O2 -std=c++17 -fPIC -shared -Wall -o $@ $(HOST_SRCS) -L$(LIBDIR) -liqhip -Wl,-rpath,'$$ORIGIN'

# stand-alone driver (the reference's "-s -te -m -blfix -n 0" evaluation) on top of the same libraries
$(LIBDIR)/iqhip_lnl: cli/iqhip_lnl.cpp $(HOST_HDRS) $(LIBDIR)/libiqhost.so
	$(CXX) -O2 -std=c++17 -Wall -o $@ cli/iqhip_lnl.cpp -L$(LIBDIR) -liqhost -liqhip -Wl,-rpath,'$$ORIGIN'

oracle/liblh_oracle.so: oracle/lh_oracle.c
	$(CC) -O3 -mavx -fopenmp -ffp-contract=off -fPIC -shared -Wall -Wextra -o $@ $< -lm

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf $(LIBDIR) oracle/liblh_oracle.so oracle/_ref

.PHONY: all ref clean
