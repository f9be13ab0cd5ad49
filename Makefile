# Top-level build: `make` = device library + host shell + CLI + oracle.
#   lib   : denseflow_amd/lib/libdfx.so  (hipcc --offload-arch=gfx950; the C ABI of include/dfx.h)
#   host  : build/libzzdenseflow.a + build/denseflow  (C++17 host shell, same target names as the reference's CMake)
#   oracle: oracle/liboracle.so  (TEST INFRASTRUCTURE ONLY)
HIPCC    ?= /opt/rocm/bin/hipcc
CXX      ?= g++
CXXFLAGS ?= -O3 -std=c++17 -Wall -Wextra -fPIC -Iinclude
CSRC     := $(wildcard denseflow_amd/csrc/*.hip denseflow_amd/csrc/*.cpp)
CHDR     := $(wildcard denseflow_amd/csrc/*.h) $(wildcard include/dfx*.h)
LIB      := denseflow_amd/lib/libdfx.so
HOSTSRC  := src/common.cpp src/utils.cpp src/image_io.cpp src/h5mini.cpp src/denseflow_gpu.cpp
HOSTOBJ  := $(patsubst src/%.cpp,build/%.o,$(HOSTSRC))

all: lib host oracle

lib: $(LIB)
$(LIB): $(CSRC) $(CHDR)
	@mkdir -p denseflow_amd/lib
	$(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -o $@ $(CSRC)

host: build/libzzdenseflow.a build/denseflow build/dfx_prof
# the torch-free process rocprofv3 profiles (C ABI only; measurement tooling)
build/dfx_prof: tools/dfx_prof.cpp include/dfx.h $(LIB)
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -o $@ tools/dfx_prof.cpp -Ldenseflow_amd/lib -ldfx -Wl,-rpath,'$$ORIGIN/../denseflow_amd/lib' -Wl,-rpath,/opt/rocm/lib
build/%.o: src/%.cpp $(wildcard include/*.h)
	@mkdir -p build
	$(CXX) $(CXXFLAGS) -c $< -o $@
build/libzzdenseflow.a: $(HOSTOBJ)
	ar rcs $@ $^
build/denseflow: tools/denseflow.cpp build/libzzdenseflow.a $(LIB)
	$(CXX) $(CXXFLAGS) -o $@ tools/denseflow.cpp build/libzzdenseflow.a -Ldenseflow_amd/lib -ldfx -lpthread -lz \
	    -Wl,-rpath,'$$ORIGIN/../denseflow_amd/lib' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle

# CPU suite (oracle, host logic, encoders vs libjpeg / libpng, the host pipeline against the test-only ABI fake, sanitizers);
# `make check-gpu` on an MI355X box: parity through the C ABI
check: lib host oracle
	python -m pytest tests -q -m "not gpu"
check-gpu: lib host oracle
	python -m pytest tests -q -m gpu

clean:
	rm -rf build $(LIB); $(MAKE) -C oracle clean
.PHONY: all lib host oracle clean check check-gpu
