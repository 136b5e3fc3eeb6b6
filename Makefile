# Builds exp_amd/libexp_amd.so (HIP, gfx950) in-tree and the CPU oracle (test infrastructure).
#   make -j8            library + oracle
#   make lib            library only
HIPCC    ?= hipcc
ARCH     ?= gfx950
HIPFLAGS ?= -O3 -std=c++17 --offload-arch=$(ARCH) -fPIC -munsafe-fp-atomics -Wall \
            -Rpass-analysis=kernel-resource-usage
# make EXPERIMENTAL=1: the tuning / A-B switches of the development rounds become environment variables again
# (EXPAMD_EXPT in exp_amd/csrc/common.h); the default build compiles them to their defaults
ifeq ($(EXPERIMENTAL),1)
HIPFLAGS += -DEXP_AMD_EXPERIMENTAL
endif
OBJ      := build/obj
CSRC     := exp_amd/csrc
SPH_LS   := 0 1 2 3 4 5 6 7 8 9 10 11 12

BASE_SRC := $(filter-out $(CSRC)/sph_inst.hip,$(wildcard $(CSRC)/*.hip))
BASE_OBJ := $(patsubst $(CSRC)/%.hip,$(OBJ)/%.o,$(BASE_SRC))
INST_OBJ := $(foreach l,$(SPH_LS),$(OBJ)/sph_inst_L$(l).o)
HDRS     := $(wildcard $(CSRC)/*.h) include/exp_amd.h

all: lib oracle h5 adaptor
lib: exp_amd/libexp_amd.so
oracle:
	$(MAKE) -C oracle

# host-side HDF5 basis-cache shim (no GPU code); skipped where the HDF5 C headers are absent
HDF5_INC ?= /opt/conda/include
HDF5_LIB ?= /opt/conda/lib
h5:
	@if [ -f $(HDF5_INC)/hdf5.h ]; then \
	  gcc -O2 -fPIC -shared -Wall -I$(HDF5_INC) exp_amd/csrc_host/h5cache.c exp_amd/csrc_host/h5part.c -o exp_amd/libexp_amd_h5.so \
	      -L$(HDF5_LIB) -lhdf5 -Wl,-rpath,$(HDF5_LIB); \
	else echo "hdf5.h not found: HDF5 cache shim not built"; fi

$(OBJ)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJ) build/log
	$(HIPCC) $(HIPFLAGS) -c $< -o $@ 2> build/log/$*.log || (grep -A8 -E "error" build/log/$*.log; exit 1)

$(OBJ)/sph_inst_L%.o: $(CSRC)/sph_inst.hip $(HDRS)
	@mkdir -p $(OBJ) build/log
	$(HIPCC) $(HIPFLAGS) -DSPH_L=$* -c $< -o $@ 2> build/log/sph_inst_L$*.log || (grep -A8 -E "error" build/log/sph_inst_L$*.log; exit 1)

exp_amd/libexp_amd.so: $(BASE_OBJ) $(INST_OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $^ -ldl

# the C++ adaptor (include/exp_amd_potaccel.hpp) and its no-Python test, built with g++ against the C ABI
adaptor: build/test_potaccel build/test_potaccel2
build/test_potaccel%: tests/cpp/test_potaccel%.cpp tests/cpp/potaccel_test_util.hpp include/exp_amd_potaccel.hpp include/exp_amd.h exp_amd/libexp_amd.so
	@mkdir -p build
	g++ -std=c++17 -O2 -Wall -Wextra -Iinclude -Itests/cpp $< -o $@ -Lexp_amd -lexp_amd \
	    -Wl,-rpath,'$$ORIGIN/../exp_amd' -Wl,-rpath-link,/opt/rocm/lib
build/test_potaccel: tests/cpp/test_potaccel.cpp tests/cpp/potaccel_test_util.hpp include/exp_amd_potaccel.hpp include/exp_amd.h exp_amd/libexp_amd.so
	@mkdir -p build
	g++ -std=c++17 -O2 -Wall -Wextra -Iinclude -Itests/cpp $< -o $@ -Lexp_amd -lexp_amd \
	    -Wl,-rpath,'$$ORIGIN/../exp_amd' -Wl,-rpath-link,/opt/rocm/lib

clean:
	rm -rf build exp_amd/libexp_amd.so exp_amd/libexp_amd_h5.so
	$(MAKE) -C oracle clean

.PHONY: all lib oracle h5 adaptor clean
