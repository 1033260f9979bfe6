# Builds the MI355X library in-tree (the .so files travel to the GPU box with the snapshot):
#   raytracer_amd/lib/libraytracer_amd.so   C-ABI of include/rt_capi.h: host loader (g++) + HIP kernels (hipcc, gfx950)
#   raytracer_amd/bin/rt_render             headless renderer: scene JSON -> BMP
#   oracle/liboracle_restate.so             CPU restatement (test infrastructure)
# -ffp-contract=off everywhere: a fused multiply-add changes hit decisions relative to the reference.
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
ARCH     ?= gfx950
CSRC     := raytracer_amd/csrc
OBJ      := build
LIB      := raytracer_amd/lib/libraytracer_amd.so
CXXFLAGS := -std=c++17 -O2 -ffp-contract=off -fPIC -Wall -Iinclude -I$(CSRC)
HIPFLAGS := --offload-arch=$(ARCH) -std=c++17 -O3 -ffp-contract=off -fPIC -Iinclude -I$(CSRC) -Wno-unused-result

HOST_SRCS := $(CSRC)/host/scene_loader.cpp $(CSRC)/host/host_api.cpp $(CSRC)/host/bmp_io.cpp $(CSRC)/host/raytracer_api.cpp $(CSRC)/host/view.cpp
HIP_SRCS  := $(CSRC)/rt_kernels.hip $(CSRC)/rt_capi_device.hip
HOST_OBJS := $(patsubst $(CSRC)/host/%.cpp,$(OBJ)/host_%.o,$(HOST_SRCS))
HIP_OBJS  := $(patsubst $(CSRC)/%.hip,$(OBJ)/hip_%.o,$(HIP_SRCS))

.PHONY: all lib oracle cli clean prof
all: lib cli oracle
lib: $(LIB)
cli: raytracer_amd/bin/rt_render raytracer_amd/bin/rt_view
oracle:
	$(MAKE) -C oracle restate

$(OBJ)/host_%.o: $(CSRC)/host/%.cpp $(wildcard include/*.h) $(wildcard $(CSRC)/host/*.h)
	@mkdir -p $(OBJ)
	$(CXX) $(CXXFLAGS) -c $< -o $@
$(OBJ)/hip_%.o: $(CSRC)/%.hip $(wildcard include/*.h) $(wildcard $(CSRC)/*.h) $(wildcard $(CSRC)/host/*.h)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@
$(LIB): $(HOST_OBJS) $(HIP_OBJS)
	@mkdir -p $(dir $@)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^
raytracer_amd/bin/rt_render: $(CSRC)/host/render_cli.cpp $(LIB)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -o $@ $< -Lraytracer_amd/lib -lraytracer_amd -Wl,-rpath,'$$ORIGIN/../lib'
raytracer_amd/bin/rt_view: $(CSRC)/host/view_cli.cpp $(LIB)
	@mkdir -p $(dir $@)
	$(CXX) $(CXXFLAGS) -o $@ $< -Lraytracer_amd/lib -lraytracer_amd -Wl,-rpath,'$$ORIGIN/../lib'
# instrumented build (per-phase shader-clock sums in the packet kernels, printed when RT_PROF_PRINT is set); use through RT_LIB_PATH
prof:
	@mkdir -p $(OBJ)/prof
	$(HIPCC) $(HIPFLAGS) -DRT_PROF=1 -c $(CSRC)/rt_kernels.hip -o $(OBJ)/prof/hip_rt_kernels.o
	$(HIPCC) $(HIPFLAGS) -DRT_PROF=1 -c $(CSRC)/rt_capi_device.hip -o $(OBJ)/prof/hip_rt_capi_device.o
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o raytracer_amd/lib/libraytracer_amd_prof.so $(HOST_OBJS) $(OBJ)/prof/hip_rt_kernels.o $(OBJ)/prof/hip_rt_capi_device.o
clean:
	rm -rf $(OBJ) $(LIB) raytracer_amd/bin
	$(MAKE) -C oracle clean
