# Builds the gfx950 C-ABI library of garage_amd (hand-written HIP kernels).
#   make            -> garage_amd/_C/libgarage_amd.so
# hipcc cross-compiles without a GPU; the .so is git-ignored but travels with
# gpurun snapshots.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := garage_amd/csrc
OUT   := garage_amd/_C
SRCS  := $(CSRC)/gae_scan.hip $(CSRC)/gemm.hip $(CSRC)/losses.hip $(CSRC)/rollout.hip
OBJS  := $(patsubst $(CSRC)/%.hip,$(OUT)/%.o,$(SRCS)) $(OUT)/errors.o $(OUT)/prof.o
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function

all: $(OUT)/libgarage_amd.so

$(OUT)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/prof.h
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OUT)/errors.o: $(CSRC)/errors.cpp
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -x hip -c $< -o $@

$(OUT)/prof.o: $(CSRC)/prof.cpp $(CSRC)/prof.h
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -x hip -c $< -o $@

$(OUT)/libgarage_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -o $@

clean:
	rm -rf $(OUT)/*.o $(OUT)/*.so

.PHONY: all clean
