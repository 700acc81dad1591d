# Builds everything in-tree for gfx950 (MI355X); hipcc cross-compiles without a GPU.
#   tron_amd/lib/libtronhip.so   the C-ABI library (HIP kernels + host orchestration + .ra I/O)
#   tron_amd/bin/tron            the command-line driver (same flags as the reference's tron)
#   oracle/libtron_oracle.so     CPU checker (test infrastructure), oracle/_ref when the reference tree is present
HIPCC     ?= /opt/rocm/bin/hipcc
ROCM      ?= /opt/rocm
ARCH      ?= gfx950
# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(); host tables
# must round exactly like an IEEE host build of the reference.
CXXFLAGS  := -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -Wall -Wno-unused-result -Wno-unused-value -Iinclude
HIPFLAGS  := $(CXXFLAGS) --offload-arch=$(ARCH)
LDFLAGS   := -L$(ROCM)/lib -lrocfft -lamdhip64 -Wl,-rpath,$(ROCM)/lib

SRC := tron_amd/csrc
OBJ := build
LIB := tron_amd/lib/libtronhip.so
BIN := tron_amd/bin/tron

all: $(LIB) $(BIN) oracle

$(OBJ)/%.o: $(SRC)/%.cpp $(wildcard $(SRC)/*.h) $(wildcard include/*.h)
	@mkdir -p $(OBJ)
	$(HIPCC) $(CXXFLAGS) -D__HIP_PLATFORM_AMD__ -c $< -o $@

$(OBJ)/%.o: $(SRC)/%.hip $(wildcard $(SRC)/*.h) $(wildcard include/*.h)
	@mkdir -p $(OBJ)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJ)/tron_kernels.o $(OBJ)/tron_grid_binned.o $(OBJ)/tron_grid_arc.o $(OBJ)/tron_grid_scatter.o $(OBJ)/tron_grid_centre.o $(OBJ)/tron_fft512.o $(OBJ)/tron_degrid_tile.o $(OBJ)/tron_degrid_stream.o $(OBJ)/tron_cgnr.o $(OBJ)/tron_traj_dev.o $(OBJ)/tron_traj.o $(OBJ)/tron_plan.o $(OBJ)/tron_pipeline.o $(OBJ)/tron_hostio.o $(OBJ)/tron_hostmath.o $(OBJ)/rawarray.o
	@mkdir -p tron_amd/lib
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $^ -o $@ $(LDFLAGS) -Wl,--version-script=$(SRC)/exports.map

$(BIN): $(OBJ)/tron_main.o $(LIB)
	@mkdir -p tron_amd/bin
	$(HIPCC) $(OBJ)/tron_main.o -o $@ -Ltron_amd/lib -ltronhip -Wl,-rpath,'$$ORIGIN/../lib' $(LDFLAGS) -pthread

oracle:
	$(MAKE) -C oracle

clean:
	rm -rf $(OBJ) tron_amd/lib tron_amd/bin
	$(MAKE) -C oracle clean

.PHONY: all oracle clean
