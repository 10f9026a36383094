# Builds libevplp_hip.so (HIP kernels for gfx950 + C-ABI + C++ host side), the evplp-render
# driver, and the CPU oracle (test infrastructure).  No cmake/ninja: plain hipcc.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH ?= gfx950
CSRC = evplp_amd/csrc
OUT = evplp_amd/lib
HIPFLAGS = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -Wno-unused-value -munsafe-fp-atomics -fno-slp-vectorize $(EXTRA_HIPFLAGS)
HOSTFLAGS = -O2 -std=c++17 -fPIC -Wall -Wno-unused-value -ffp-contract=off

HIP_SRCS = $(CSRC)/kernels_trace.hip $(CSRC)/kernels_gather.hip $(CSRC)/kernels_splat.hip $(CSRC)/kernels_pt.hip
CPP_SRCS = $(CSRC)/context.cpp $(CSRC)/bvh_build.cpp $(wildcard $(CSRC)/host/*.cpp)
HIP_OBJS = $(patsubst $(CSRC)/%.hip,build/%.o,$(HIP_SRCS))
CPP_OBJS = $(patsubst $(CSRC)/%.cpp,build/%.o,$(filter-out $(CSRC)/host/driver_main.cpp,$(CPP_SRCS)))
HDRS = $(wildcard $(CSRC)/*.h $(CSRC)/*.hpp $(CSRC)/host/*.hpp include/*.h)

all: $(OUT)/libevplp_hip.so $(OUT)/evplp-render oracle

$(OUT)/libevplp_hip.so: $(HIP_OBJS) $(CPP_OBJS)
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^

build/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

build/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(dir $@)
	$(HIPCC) $(HOSTFLAGS) -x c++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $< -o $@

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

.PHONY: all oracle clean isa
