# Top-level build: everything lands in-tree so that gpurun ships the binaries.
#   raytracer_amd/librt_hip.so   HIP kernels + C ABI (include/rt.h), gfx950 only
#   raytracer_amd/librt_host.so  host helper (include/rt_host.h)
#   oracle/librt_oracle.so       CPU restatement (test infrastructure)
#   oracle/_ref/*                the reference itself, when /root/reference exists
HIPCC ?= /opt/rocm/bin/hipcc
CXX   ?= g++
PKG   := raytracer_amd
HIPFLAGS = --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off \
           -fno-gpu-rdc -Iinclude -Wall -Wno-unused-function

all: host hip oracle

host: $(PKG)/librt_host.so
$(PKG)/librt_host.so: $(PKG)/host/rt_host.cpp $(PKG)/host/rt_scene_store.hpp include/rt.h include/rt_host.h
	$(CXX) -std=c++17 -O2 -fPIC -shared -ffp-contract=off -fno-trapping-math -fno-math-errno \
	    -Iinclude -I$(PKG)/host -o $@ $(PKG)/host/rt_host.cpp

hip: $(PKG)/librt_hip.so $(PKG)/librt_hip_splittest.so
$(PKG)/librt_hip.so: $(wildcard $(PKG)/csrc/*.hip $(PKG)/csrc/*.h) include/rt.h
	$(HIPCC) $(HIPFLAGS) -I$(PKG)/csrc -o $@ $(PKG)/csrc/rt_hip.hip
# test build of the same sources: subtree donation active from the first ray on and no partial refills, so the
# parity tests exercise that path on every ray instead of only at the end of a kernel (tests/test_gpu_parity.py)
$(PKG)/librt_hip_splittest.so: $(wildcard $(PKG)/csrc/*.hip $(PKG)/csrc/*.h) include/rt.h
	$(HIPCC) $(HIPFLAGS) -DRT_SPLIT_ALWAYS -DRT_REFILL_MIN=64 -DRT_REFILL_MIN_ANY=64 -DRT_REFILL_PATIENCE=1000000 -I$(PKG)/csrc -o $@ $(PKG)/csrc/rt_hip.hip

# experiment build: make variant NAME=foo DEFS="-DRT_STACK_LDS=8"  -> raytracer_amd/librt_hip_foo.so (RT_HIP_SO selects it)
variant:
	$(HIPCC) $(HIPFLAGS) $(DEFS) -I$(PKG)/csrc -o $(PKG)/librt_hip_$(NAME).so $(PKG)/csrc/rt_hip.hip

oracle: hip host
	$(MAKE) -C oracle all

clean:
	rm -f $(PKG)/*.so oracle/*.so
.PHONY: all host hip oracle clean
