# Builds the gfx950 shared library (zeth_amd/libzeth_gpu.so) and the CPU oracle (oracle/liboracle.so).
HIPCC ?= hipcc
ARCH ?= gfx950
HIPFLAGS ?= -O3 --offload-arch=$(ARCH) -fPIC -std=c++17 -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result
CSRC := zeth_amd/csrc
OBJDIR := build
SRCS := $(wildcard $(CSRC)/*.hip)
OBJS := $(patsubst $(CSRC)/%.hip,$(OBJDIR)/%.o,$(SRCS))
HDRS := $(wildcard $(CSRC)/*.cuh) $(wildcard $(CSRC)/*.h) include/zeth_gpu.h

all: zeth_amd/libzeth_gpu.so

$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

zeth_amd/libzeth_gpu.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -rf $(OBJDIR) zeth_amd/libzeth_gpu.so

.PHONY: all clean
