# Builds libevplp_hip.so (HIP kernels for gfx950 + C-ABI + C++ host side), the evplp-render
# driver, and the CPU oracle (test infrastructure).  No cmake/ninja: plain hipcc.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = evplp_amd/csrc
OUT = evplp_amd/lib
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value -munsafe-fp-atomics -fno-slp-vectorize $(EXTRA_HIPFLAGS)
HOSTFLAGS = -O2 -std=c++17 -fPIC -Wall -Wno-unused-value -ffp-contract=off

HIP_SRCS = $(CSRC)/kernels_trace.hip $(CSRC)/kernels_gather.hip $(CSRC)/kernels_cut.hip $(CSRC)/kernels_splat.hip $(CSRC)/kernels_pt.hip $(CSRC)/bvh_gpu.hip $(CSRC)/selftest.hip
CPP_SRCS = $(CSRC)/context.cpp $(CSRC)/group.cpp $(CSRC)/bvh_build.cpp $(wildcard $(CSRC)/host/*.cpp)
# VARIANT selects a separate object directory and library name (developer builds, e.g. `make stats`)
VARIANT ?=
BUILD = build$(if $(VARIANT),/$(VARIANT),)
LIBSO = $(OUT)/libevplp_hip$(if $(VARIANT),_$(VARIANT),).so
HIP_OBJS = $(patsubst $(CSRC)/%.hip,$(BUILD)/%.o,$(HIP_SRCS))
CPP_OBJS = $(patsubst $(CSRC)/%.cpp,$(BUILD)/%.o,$(filter-out $(CSRC)/host/driver_main.cpp,$(CPP_SRCS)))
HDRS = $(wildcard $(CSRC)/*.h $(CSRC)/*.hpp $(CSRC)/host/*.hpp include/*.h)

all: $(OUT)/libevplp_hip.so $(OUT)/evplp-render oracle nan

$(LIBSO): $(HIP_OBJS) $(CPP_OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -ldl

# diagnostic build with traversal counters (nodes / leaf blocks / triangle pairs per walk): tools/traversal_stats.py
stats:
	$(MAKE) VARIANT=stats EXTRA_HIPFLAGS=-DEVPLP_TRAVERSAL_STATS=1 $(OUT)/libevplp_hip_stats.so

# debug build that counts non-finite partial sums of the gathers and the splat (the reference's ASSERT under DEBUG, all.cuh:10-17):
# tests/test_gpu_debug_nan.py runs it once, outside every timed path
nan:
	$(MAKE) VARIANT=nan EXTRA_HIPFLAGS=-DEVPLP_DEBUG_NAN=1 $(OUT)/libevplp_hip_nan.so

# the feeders (G-buffer, light tracing) are compiled without floating-point contraction: every operation rounds as in the oracle,
# so G-buffers and light-path records can be compared bit for bit; the hot kernels keep contraction (radiance is toleranced)
$(BUILD)/kernels_trace.o: HIPFLAGS += -ffp-contract=off $(TRACE_FLAGS)
$(BUILD)/kernels_gather.o: HIPFLAGS += $(GATHER_FLAGS)
$(BUILD)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(BUILD)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HOSTFLAGS) $(EXTRA_HIPFLAGS) -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $< -o $@

$(OUT)/evplp-render: $(CSRC)/host/driver_main.cpp $(OUT)/libevplp_hip.so
	g++ -O2 -std=c++17 -Iinclude -o $@ $< -L$(OUT) -levplp_hip -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

isa:
	@mkdir -p build/isa
	$(HIPCC) $(HIPFLAGS) -S --cuda-device-only -o build/isa/kernels_gather.s $(CSRC)/kernels_gather.hip

clean:
	rm -rf build $(OUT)/*.so $(OUT)/evplp-render
	$(MAKE) -C oracle clean

.PHONY: all oracle clean isa stats nan
