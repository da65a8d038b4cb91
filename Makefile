# Builds the gfx950 shared library (zeth_amd/libzeth_gpu.so) and the CPU oracle (oracle/liboracle.so).
HIPCC ?= hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
CSRC := zeth_amd/csrc
OBJDIR := build
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.cuh) $(wildcard $(CSRC)/*.h) include/zeth_gpu.h

all: zeth_amd/libzeth_gpu.so host dbg

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

zeth_amd/libzeth_gpu.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

# The same sources with -DZG_DEBUG_KNOBS: the tuning / measurement / fault-injection knobs of csrc/knobs.h are compiled in.
# For tools/ and for the one test that injects a failure (select it with ZETH_GPU_LIBRARY); never the shipped library.
DBGDIR := build/dbg
DBGOBJS := $(patsubst $(CSRC)/%.hip,$(DBGDIR)/%.o,$(SRCS))
$(DBGDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(DBGDIR)
	$(HIPCC) $(HIPFLAGS) -DZG_DEBUG_KNOBS -c $< -o $@

$(DBGDIR)/libzeth_gpu.so: $(DBGOBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(DBGOBJS) -o $@

dbg: $(DBGDIR)/libzeth_gpu.so

clean:
	rm -rf $(OBJDIR) zeth_amd/libzeth_gpu.so

.PHONY: all clean host dbg

# C++ host module above the C ABI (include/libzeth_gpu/): the zeth-tool `prove` counterpart and the adapter test driver
HOSTCXX ?= g++
HOSTFLAGS ?= -O2 -std=c++17 -Wall -Wextra -Iinclude
HOSTBINS := build/zeth-gpu-tool build/host_adapter_test
HOSTHDRS := $(wildcard include/libzeth_gpu/*.hpp) include/zeth_gpu.h

build/zeth-gpu-tool: zeth_amd/host/zeth_gpu_tool.cpp $(HOSTHDRS) zeth_amd/libzeth_gpu.so
	@mkdir -p build
	$(HOSTCXX) $(HOSTFLAGS) $< -o $@ -Lzeth_amd -lzeth_gpu -Wl,-rpath,'$$ORIGIN/../zeth_amd' -lpthread

build/host_adapter_test: zeth_amd/host/host_adapter_test.cpp $(HOSTHDRS) zeth_amd/libzeth_gpu.so
	@mkdir -p build
	$(HOSTCXX) $(HOSTFLAGS) $< -o $@ -Lzeth_amd -lzeth_gpu -Wl,-rpath,'$$ORIGIN/../zeth_amd' -lpthread

host: $(HOSTBINS)
