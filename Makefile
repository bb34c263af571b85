# Build libdagr_hip.so (gfx950 only) and the CPU oracle.  No cmake: one hipcc invocation per TU.
HIPCC      ?= hipcc
ARCH       ?= gfx950
# -ffp-contract=off: several integer decisions (voxel ids, denormalised coordinates) hang on
# separately-rounded fp32 multiply/add/divide exactly as torch computes them (SURVEY QUIRK-2).
HIPFLAGS   ?= --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Iinclude
SRC        := $(wildcard dagr_amd/csrc/*.hip)
OBJ        := $(patsubst dagr_amd/csrc/%.hip,build/%.o,$(SRC))
LIB        := dagr_amd/lib/libdagr_hip.so
# the same sources with the measurement knobs compiled in (common.hpp:knob -- A/B switches read from the environment;
# some make results wrong on purpose).  Loaded only by the probes under tools/ (DAGR_HIP_LIB=<this file>), never by default.
OBJ_M      := $(patsubst dagr_amd/csrc/%.hip,build/measure/%.o,$(SRC))
LIB_M      := dagr_amd/lib/libdagr_hip_measure.so

all: $(LIB) $(LIB_M) oracle

measure: $(LIB_M)

$(LIB_M): $(OBJ_M)
	@mkdir -p dagr_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ_M) -L/opt/rocm/lib -lhipblaslt

build/measure/%.o: dagr_amd/csrc/%.hip dagr_amd/csrc/common.hpp dagr_amd/csrc/pool_common.hpp include/dagr_hip.h
	@mkdir -p build/measure
	$(HIPCC) $(HIPFLAGS) -DDAGR_MEASURE -c $< -o $@

$(LIB): $(OBJ)
	@mkdir -p dagr_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ) -L/opt/rocm/lib -lhipblaslt

build/%.o: dagr_amd/csrc/%.hip dagr_amd/csrc/common.hpp dagr_amd/csrc/pool_common.hpp include/dagr_hip.h
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf build $(LIB) $(LIB_M)
	$(MAKE) -C oracle clean

.PHONY: all measure oracle clean
