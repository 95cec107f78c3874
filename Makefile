# Top-level build: everything lands in-tree (git-ignored, but shipped to the GPU box).
#   make lib     -> coati_amd/_build/libcoati_hip.so   (HIP kernels + C ABI, gfx950)
#   make host    -> coati_amd/_build/libcoati_host.so  (C++ host layer: models, seq prep, I/O)
#   make oracle  -> oracle/_build/libcoati_oracle.so   (test infrastructure)
#   make ref     -> oracle/_ref/libcoati_ref.so        (only where /root/reference exists)
HIPCC    ?= /opt/rocm/bin/hipcc
ARCH     ?= gfx950
BUILD     = coati_amd/_build
HIPFLAGS  = --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
            -Wall -Wextra -Wno-unused-parameter -Iinclude

all: lib dist host oracle

lib: $(BUILD)/libcoati_hip.so

HIP_SRC = coati_amd/csrc/abi.hip coati_amd/csrc/plan.hip coati_amd/csrc/pipeline.hip coati_amd/csrc/sample_host.hip coati_amd/csrc/viterbi_ck.hip coati_amd/csrc/viterbi_l1.hip coati_amd/csrc/viterbi_lp.hip coati_amd/csrc/dp_generic.hip coati_amd/csrc/forward_l1.hip coati_amd/csrc/viterbi_k.hip coati_amd/csrc/forward_k.hip \
          coati_amd/csrc/sampleback.hip
$(BUILD)/libcoati_hip.so: $(HIP_SRC) coati_amd/csrc/viterbi_lp_block.inc coati_amd/csrc/abi_internal.hpp coati_amd/csrc/common.hpp coati_amd/csrc/viterbi_cell.hpp coati_amd/csrc/glibc_math.hpp include/coati_hip.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(HIP_SRC)

# multi-GPU layer: RCCL linked directly (include/coati_hip_dist.h)
dist: $(BUILD)/libcoati_hip_dist.so
$(BUILD)/libcoati_hip_dist.so: coati_amd/csrc/dist.hip include/coati_hip_dist.h include/coati_hip.h $(BUILD)/libcoati_hip.so
	$(HIPCC) $(HIPFLAGS) -shared -o $@ coati_amd/csrc/dist.hip -L$(BUILD) -lcoati_hip -L/opt/rocm/lib -lrccl -Wl,-rpath,'$$ORIGIN' -Wl,-rpath,/opt/rocm/lib

# debug variant with per-wave clock stamps in the fill kernel (tools/trace_fill.py)
trace: $(BUILD)/libcoati_hip_trace.so
$(BUILD)/libcoati_hip_trace.so: $(HIP_SRC) coati_amd/csrc/viterbi_lp_block.inc coati_amd/csrc/abi_internal.hpp coati_amd/csrc/common.hpp coati_amd/csrc/viterbi_cell.hpp coati_amd/csrc/glibc_math.hpp include/coati_hip.h
	@mkdir -p $(BUILD)
	$(HIPCC) $(HIPFLAGS) -DCOATI_FILL_TRACE -shared -o $@ $(HIP_SRC)

HOST_SRC = coati_amd/host/model.cc coati_amd/host/seq.cc coati_amd/host/synth.cc coati_amd/host/io.cc \
           coati_amd/host/align.cc coati_amd/host/cli.cc coati_amd/host/format.cc coati_amd/host/tree.cc coati_amd/host/insertions.cc coati_amd/host/msa.cc coati_amd/host/capi.cc
HOST_HDR = $(wildcard coati_amd/host/*.hpp) coati_amd/host/ecm_kosiol2007.inc
CXX      ?= g++
HOSTFLAGS = -std=c++17 -O2 -fPIC -pthread -ffp-contract=off -fno-fast-math -Wall -Wextra -Iinclude -Icoati_amd/host

host: $(BUILD)/libcoati_host.so $(BUILD)/coati-alignpair $(BUILD)/coati-sample $(BUILD)/coati-format $(BUILD)/coati-msa $(BUILD)/coati-genseed $(BUILD)/coati

# the host layer calls the DP through the C ABI of libcoati_hip.so only
$(BUILD)/libcoati_host.so: $(HOST_SRC) $(HOST_HDR) $(BUILD)/libcoati_hip.so
	@mkdir -p $(BUILD)
	$(CXX) $(HOSTFLAGS) -shared -o $@ $(HOST_SRC) -L$(BUILD) -lcoati_hip -Wl,-rpath,'$$ORIGIN' -lm -ldl

$(BUILD)/coati: coati_amd/host/coati_main.cc
	@mkdir -p $(BUILD)
	$(CXX) $(HOSTFLAGS) -o $@ $<

$(BUILD)/coati-%: coati_amd/host/coati_%.cc $(BUILD)/libcoati_host.so
	$(CXX) $(HOSTFLAGS) -o $@ $< -L$(BUILD) -lcoati_host -lcoati_hip -Wl,-rpath,'$$ORIGIN' -lm -ldl

# sanitizer build of the host layer (CPU only; GPU ASan is not available on the pool):
# run the CPU tests / tools/fuzz_host_io.py against it with COATI_HOST_LIB + LD_PRELOAD (see the tool)
asan: $(BUILD)/asan/libcoati_host.so
# ... against a STUB libcoati_hip.so (every entry answers COATI_HIP_ENODEVICE): the sanitizer build maps no HIP runtime
$(BUILD)/asan/libcoati_hip.so: coati_amd/host/hip_stub.cc include/coati_hip.h
	@mkdir -p $(BUILD)/asan
	$(CXX) -std=c++17 -O1 -g -fPIC -Iinclude -shared -o $@ $<
$(BUILD)/asan/libcoati_host.so: $(HOST_SRC) $(HOST_HDR) $(BUILD)/asan/libcoati_hip.so
	@mkdir -p $(BUILD)/asan
	$(CXX) -std=c++17 -O1 -g -fPIC -ffp-contract=off -fno-fast-math -fsanitize=address,undefined -fno-omit-frame-pointer -pthread \
	    -Iinclude -Icoati_amd/host -shared -o $@ $(HOST_SRC) -L$(BUILD)/asan -lcoati_hip -Wl,-rpath,'$$ORIGIN' -lm -ldl

oracle:
	$(MAKE) -C oracle

ref:
	$(MAKE) -C oracle ref

clean:
	rm -rf $(BUILD)
	$(MAKE) -C oracle clean

.PHONY: all lib dist host oracle ref clean asan trace
