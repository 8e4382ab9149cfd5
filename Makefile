# Builds the gfx950 C-ABI library of garage_amd (hand-written HIP kernels).
#   make            -> garage_amd/_C/libgarage_amd.so
# hipcc cross-compiles without a GPU; the .so is git-ignored but travels with
# gpurun snapshots.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := garage_amd/csrc
OUT   := garage_amd/_C
HIPS  := gae_scan gemm skinny losses rollout policy_fused small_step fused_train narrow_step lnorm
CPPS  := errors prof update comm rollout_loop
OBJS  := $(patsubst %,$(OUT)/%.o,$(HIPS) $(CPPS))
# -fno-slp-vectorize: hipcc's SLP vectorizer turns pairs of fp32 operations into packed
# VOP3P instructions and, where one operand is the high half of a register pair, sets
# OP_SEL[1] (v_pk_fma_f32 ... op_sel:[0,1,0], v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1]).
# On the MI355X boxes of this pool exactly that form reads a WRONG operand now and then
# while another wave of the SIMD executes v_mfma_f32_32x32x16_bf16
# (tools/mfma_valu_hazard.hip reproduces it in 100 lines).  No
# bf16 MFMA runs in the default exact-fp32 mode, but the opt-in split-operand k-loops
# and any bf16 work of another stream or process do; the library is built without
# packed fp32 altogether (same-box A/B: 108.8 ms per C3 iteration either way), and
# tests/test_isa_hazard_cpu.py scans the generated code for the form.
FLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fno-slp-vectorize $(EXTRA)

all: $(OUT)/libgarage_amd.so

$(OUT)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/prof.h $(CSRC)/small_step.h $(CSRC)/gemm_core.h $(CSRC)/loss_rows.h $(CSRC)/fused_train.h $(CSRC)/rollout_dev.h
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -c $< -o $@

$(OUT)/%.o: $(CSRC)/%.cpp $(CSRC)/prof.h $(CSRC)/small_step.h $(CSRC)/fused_train.h include/garage_amd.h
	@mkdir -p $(OUT)
	$(HIPCC) $(FLAGS) -x hip -c $< -o $@

$(OUT)/libgarage_amd.so: $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OBJS) -ldl -o $@

# developer tool: the matrix pipe's own ceiling (register-only MFMA loops)
mfma-peak: tools/mfma_peak.hip
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result $< -o $(OUT)/mfma_peak

# developer tools: does vector work hide behind MFMAs (fp32: no, bf16: yes), and the
# packed-fp32 / bf16-MFMA hazard reproducer
mfma-tools: tools/mfma_valu_overlap.hip tools/mfma_valu_hazard.hip tools/mfma_rounding.hip
	@mkdir -p $(OUT)
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result -Wno-unused-value tools/mfma_valu_overlap.hip -o $(OUT)/mfma_valu_overlap
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result -Wno-unused-value tools/mfma_valu_hazard.hip -o $(OUT)/mfma_valu_hazard
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result -Wno-unused-value tools/mfma_rounding.hip -o $(OUT)/mfma_rounding

# the library WITH hipcc's SLP vectorizer (packed fp32, the hazardous form included):
# only for tools/hazard_repro_backward.py
slp-variant:
	@mkdir -p $(OUT)/variants/slp
	for f in $(HIPS); do $(HIPCC) $(FLAGS) -fslp-vectorize -c $(CSRC)/$$f.hip -o $(OUT)/variants/slp/$$f.o || exit 1; done
	for f in $(CPPS); do $(HIPCC) $(FLAGS) -fslp-vectorize -x hip -c $(CSRC)/$$f.cpp -o $(OUT)/variants/slp/$$f.o || exit 1; done
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC $(OUT)/variants/slp/*.o -ldl -o $(OUT)/variants/lib_slp.so
	rm -rf $(OUT)/variants/slp

clean:
	rm -rf $(OUT)/*.o $(OUT)/*.so $(OUT)/mfma_peak $(OUT)/mfma_valu_overlap $(OUT)/mfma_valu_hazard $(OUT)/mfma_rounding $(OUT)/variants

.PHONY: all clean mfma-peak mfma-tools slp-variant

# Host-side epoch / rollout loops under AddressSanitizer + UBSan on the CPU, with
# every kernel entry point replaced by a recording fake (tests/host/).
asan-host: $(OUT)/host_asan_test
	$(OUT)/host_asan_test

$(OUT)/host_asan_test: tests/host/update_loop_harness.cpp $(CSRC)/update.cpp $(CSRC)/rollout_loop.cpp $(CSRC)/small_step.h $(CSRC)/fused_train.h include/garage_amd.h
	@mkdir -p $(OUT)
	g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer \
	  -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Wall -Wno-unused-function \
	  tests/host/update_loop_harness.cpp $(CSRC)/update.cpp $(CSRC)/rollout_loop.cpp \
	  -o $@

.PHONY: asan-host
