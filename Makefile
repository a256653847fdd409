# Build recipe for the native pieces.  `make` builds the product library (HIP, gfx950).
# `make hostcheck` builds the CPU debug harness of the per-ray core used by tests only.
ROOT := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
HIPCC ?= hipcc
CXX ?= g++
ARCH ?= gfx950

LIB := $(ROOT)tracer_amd/lib/libtracer_amd.so
SRC := $(ROOT)tracer_amd/csrc/trc_kernels.hip
HDR := $(ROOT)tracer_amd/csrc/trc_core.h $(ROOT)tracer_amd/csrc/trc_bounds.h $(ROOT)tracer_amd/csrc/trc_footprint.h $(ROOT)tracer_amd/csrc/trc_stream.inc $(ROOT)include/tracer_amd.h

HOSTCHECK := $(ROOT)tests/hostcheck/libtrc_hostcheck.so
HOSTCHECK_SRC := $(ROOT)tests/hostcheck/hostcheck.cpp

all: $(LIB)

# -ffp-contract=off: a*b+c is rounded twice, as NumPy rounds it in the reference.  Fused, the same formula rounds differently in
# each kernel it is inlined into, and the forms of the engine end ~1e-7 of the rays on different surfaces;
# it costs 0.8 % on the bench.  fma() where it is written stays an fma.
FPFLAGS := -ffp-contract=off

$(LIB): $(SRC) $(HDR)
	mkdir -p $(dir $(LIB))
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -munsafe-fp-atomics $(FPFLAGS) -fPIC -shared \
		-Wno-unused-result -o $@ $(SRC)

hostcheck: $(HOSTCHECK)

$(HOSTCHECK): $(HOSTCHECK_SRC) $(HDR)
	$(CXX) -O2 -std=c++17 -fPIC -shared $(FPFLAGS) -o $@ $(HOSTCHECK_SRC)

asm: $(SRC) $(HDR)
	mkdir -p $(ROOT)build
	$(HIPCC) -O3 -std=c++17 --offload-arch=$(ARCH) -munsafe-fp-atomics $(FPFLAGS) -S --cuda-device-only \
		-Rpass-analysis=kernel-resource-usage -o $(ROOT)build/trc_kernels.s $(SRC)

clean:
	rm -f $(LIB) $(HOSTCHECK)

.PHONY: all hostcheck asm clean
