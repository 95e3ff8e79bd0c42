# Builds the product library (HIP kernels + host engine + C ABI) for gfx950 only, in-tree.
#   make            -> audiomod_amd/lib/libaudiomod_pv.so
#   make oracle     -> oracle/libpv_oracle.so (test infrastructure)
#   make ref        -> oracle/_ref/* (the real reference; only where /root/reference exists)
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  := gfx950
# -ffp-contract=off is part of the numerical contract (see pv_kernels.hip header): no FMA contraction,
# on the device or in the host planner.  -fno-slp-vectorize: left alone, the compiler pairs neighbouring scalar
# f32 adds / multiplies into v_pk_add_f32 / v_pk_mul_f32, which on gfx950 issue slower than the two scalar
# instructions they replace (measured: analysis kernel -11 %, synthesis -3 %, overlap-add -5 % with the flag);
# the one loop that gains from packed math (the resampler's) asks for it explicitly.  Same arithmetic either way.
CXXFLAGS := -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Iinclude -Iaudiomod_amd/csrc -Wall -Wno-unused-result
SRC := audiomod_amd/csrc/pv_kernels.hip audiomod_amd/csrc/pv_hostio.hip audiomod_amd/csrc/pv_engine.cc audiomod_amd/csrc/pv_plan.cc audiomod_amd/csrc/phasevocoder.cc
HDR := $(wildcard audiomod_amd/csrc/*.h) $(wildcard include/*.h) $(wildcard include/dafx/*.h)
LIB := audiomod_amd/lib/libaudiomod_pv.so

all: $(LIB)

$(LIB): $(SRC) $(HDR)
	@mkdir -p audiomod_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) $(CXXFLAGS) -shared $(SRC) -o $@

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -f oracle/ref.mk

clean:
	rm -rf audiomod_amd/lib
.PHONY: all oracle ref clean

# stand-alone program over the drop-in C++ class (tests/test_shim.py runs it on the GPU box)
SHIM := audiomod_amd/lib/shim_demo
$(SHIM): tools/shim_demo.cc $(LIB) include/dafx/phasevocoder.h include/dafx/modbase.h
	$(HIPCC) -O2 -std=c++17 -Iinclude -Iinclude/dafx tools/shim_demo.cc -Laudiomod_amd/lib -laudiomod_pv -Wl,-rpath,'$$ORIGIN' -o $@
CLI := audiomod_amd/lib/audiomod-pv-exe
$(CLI): audiomod_amd/csrc/audiomod_pv_cli.cc $(LIB) include/dafx/phasevocoder.h include/dafx/modbase.h
	$(HIPCC) -O2 -std=c++17 -Iinclude -Iinclude/dafx audiomod_amd/csrc/audiomod_pv_cli.cc -Laudiomod_amd/lib -laudiomod_pv -Wl,-rpath,'$$ORIGIN' -o $@
all: $(SHIM) $(CLI)

SBENCH := audiomod_amd/lib/stream_bench
$(SBENCH): tools/stream_bench.cc $(LIB) include/dafx/phasevocoder.h
	$(HIPCC) -O2 -std=c++17 -Iinclude -Iinclude/dafx tools/stream_bench.cc -Laudiomod_amd/lib -laudiomod_pv -Wl,-rpath,'$$ORIGIN' -o $@
all: $(SBENCH)
