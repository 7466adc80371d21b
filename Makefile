# Build everything that has to exist before tests / bench run.
#   make            runtime library (HIP, gfx950) + lowering tool + oracle
#   make rt         neptune-pde-solver_amd/lib/libneptune_hip.so
#   make lowering   neptune-pde-solver_amd/bin/neptune-opt + lib/libneptune_lowering.so
#   make oracle     oracle/_build/liboracle.so   (CPU restatement; test infrastructure)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CC       ?= gcc
ARCH     ?= gfx950
PKG      := neptune-pde-solver_amd
CSRC     := $(PKG)/csrc
LIBDIR   := $(PKG)/lib
BINDIR   := $(PKG)/bin

# -ffp-contract=off: the reference evaluates stencil bodies op by op, strict IEEE, no FMA
# (lib/Pipeline/NeptuneIRPassesPipeline.cpp:28-46); bit-exact parity needs the same here.
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function \
            -Wno-unused-but-set-variable -Wno-unused-variable
KERNEL_HDRS := $(wildcard $(CSRC)/kernels/*.hpp) $(wildcard $(CSRC)/runtime/*.hpp) include/neptune_hip.h

.PHONY: all rt lowering oracle clean resources
all: rt lowering oracle

# two translation units: the kernels + runtime (minutes: every tile x every built-in body) and the host-only slab /
# RCCL code (seconds)
rt: $(LIBDIR)/libneptune_hip.so
build/obj/neptune_hip_rt.o: $(CSRC)/runtime/neptune_hip_rt.hip $(KERNEL_HDRS)
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
build/obj/slab_rccl.o: $(CSRC)/runtime/slab_rccl.hip $(CSRC)/runtime/slab_rccl.hpp $(CSRC)/kernels/apply_launch.hpp include/neptune_hip.h
	@mkdir -p build/obj
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIBDIR)/libneptune_hip.so: build/obj/neptune_hip_rt.o build/obj/slab_rccl.o
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared $^ -ldl -o $@

# per-kernel VGPR/SGPR/occupancy table
resources: $(CSRC)/runtime/neptune_hip_rt.hip $(KERNEL_HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c -Rpass-analysis=kernel-resource-usage $< -o build/res.o 2> build/resources.log
	python3 tools/kernel_resources.py build/resources.log

LOWERING_SRCS := $(wildcard $(CSRC)/lowering/*.cpp)
LOWERING_HDRS := $(wildcard $(CSRC)/lowering/*.h)
lowering: $(LIBDIR)/libneptune_lowering.so $(BINDIR)/neptune-opt
$(LIBDIR)/libneptune_lowering.so: $(filter-out %/neptune_opt_main.cpp,$(LOWERING_SRCS)) $(LOWERING_HDRS)
	@mkdir -p $(LIBDIR)
	$(CXX) -O2 -std=c++17 -fPIC -shared -Wall $(filter-out %/neptune_opt_main.cpp,$(LOWERING_SRCS)) -o $@
$(BINDIR)/neptune-opt: $(LOWERING_SRCS) $(LOWERING_HDRS)
	@mkdir -p $(BINDIR)
	$(CXX) -O2 -std=c++17 -Wall $(LOWERING_SRCS) -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIBDIR)/*.so $(BINDIR) oracle/_build
