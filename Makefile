# Build recipe for the native pieces.  `make` builds the product library (HIP, gfx950).
# `make hostcheck` builds the CPU debug harness of the per-ray core used by tests only.
ROOT := $(dir $(abspath $(lastword $(MAKEFILE_LIST))))
HIPCC ?= hipcc
CXX ?= g++
ARCH ?= gfx950

LIB := $(ROOT)tracer_amd/lib/libtracer_amd.so
CSRC := $(ROOT)tracer_amd/csrc
# two translation units: the engines and the C-ABI (trc_kernels.hip, which includes trc_stream.inc), and the class-split shading
# kernels of the streaming engine (trc_shade.hip).  `make -j2` compiles them side by side.
SRC := $(CSRC)/trc_kernels.hip
SRCS := $(CSRC)/trc_kernels.hip $(CSRC)/trc_shade.hip
OBJDIR := $(ROOT)build/obj
OBJS := $(OBJDIR)/trc_kernels.o $(OBJDIR)/trc_shade.o
HDR := $(CSRC)/trc_core.h $(CSRC)/trc_bounds.h $(CSRC)/trc_footprint.h $(CSRC)/trc_device.h $(CSRC)/trc_stream.inc $(ROOT)include/tracer_amd.h

HOSTCHECK := $(ROOT)tests/hostcheck/libtrc_hostcheck.so
HOSTCHECK_SRC := $(ROOT)tests/hostcheck/hostcheck.cpp

all: $(LIB)

# -ffp-contract=off: a*b+c is rounded twice, as NumPy rounds it in the reference.  Fused, the same formula rounds differently in
# each kernel it is inlined into, and the forms of the engine end ~1e-7 of the rays on different surfaces;
# it costs 0.8 % on the bench.  fma() where it is written stays an fma.
FPFLAGS := -ffp-contract=off

HIPFLAGS := -O3 -std=c++17 --offload-arch=$(ARCH) -munsafe-fp-atomics $(FPFLAGS) -fPIC -Wno-unused-result

# No machine-level loop-invariant code motion.  The kernels are grid-stride loops around chains of float64 library code
# (logarithm, sine / cosine, tangent, arc cosine, square roots and divisions), and the pass moves the materialisation of every
# constant of those chains in front of the loop, where each then holds two registers for the whole kernel: the lean shading
# kernels 170 / 241 registers with it, 94 / 116 without; k_s_shade 243 -> 131, k_s_fresh 168 -> 95, k_s_bounce 128 + 180 bytes of
# scratch -> 119, k_trace_coop 256 + 408 bytes -> 165, k_ord_bounce 166 -> 121 (build/*.resources.txt, `make asm`).
# Measured (profiles/README.md): NSTTF +3 %, dish +4 %, cavity +9 %, the megakernel on the dish +17 %.
SHADE_FLAGS := -mllvm -disable-machine-licm
KERNELS_FLAGS := -mllvm -disable-machine-licm

$(OBJDIR)/trc_shade.o: $(CSRC)/trc_shade.hip $(HDR)
	mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(SHADE_FLAGS) -c -o $@ $<

$(OBJDIR)/trc_kernels.o: $(CSRC)/trc_kernels.hip $(HDR)
	mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(KERNELS_FLAGS) -c -o $@ $<

$(LIB): $(OBJS)
	mkdir -p $(dir $(LIB))
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -o $@ $(OBJS)

hostcheck: $(HOSTCHECK)

$(HOSTCHECK): $(HOSTCHECK_SRC) $(HDR)
	$(CXX) -O2 -std=c++17 -fPIC -shared $(FPFLAGS) -o $@ $(HOSTCHECK_SRC)

# assembly listings with the registers, scratch and occupancy of every kernel (build/*.s, build/*.resources.txt)
asm: $(SRCS) $(HDR)
	mkdir -p $(ROOT)build
	for f in trc_kernels trc_shade; do \
		fl="$(KERNELS_FLAGS)"; [ $$f = trc_shade ] && fl="$(SHADE_FLAGS)"; \
		$(HIPCC) $(HIPFLAGS) $$fl -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
			-o $(ROOT)build/$$f.s $(CSRC)/$$f.hip 2> $(ROOT)build/$$f.resources.txt || exit 1; done

asm-shade: $(CSRC)/trc_shade.hip $(HDR)
	mkdir -p $(ROOT)build
	$(HIPCC) $(HIPFLAGS) $(SHADE_FLAGS) -S --cuda-device-only -Rpass-analysis=kernel-resource-usage \
		-o $(ROOT)build/trc_shade.s $(CSRC)/trc_shade.hip 2> $(ROOT)build/trc_shade.resources.txt

clean:
	rm -f $(LIB) $(HOSTCHECK)

.PHONY: all hostcheck asm asm-shade clean
