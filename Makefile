# Builds libvita_hip.so (gfx950 only) and the oracle's native pieces.  `make` cross-compiles
# without a GPU; the .so is git-ignored but travels to the GPU box with the tree.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := vita_amd/csrc
SRCS  := $(CSRC)/vita_api.hip $(CSRC)/vita_gemm_f32.hip $(CSRC)/vita_gemm_f32_dma.hip $(CSRC)/vita_gemm_f32_t.hip $(CSRC)/vita_gemm_f32_small.hip $(CSRC)/vita_gemm_bf16_tn.hip $(CSRC)/vita_gemm_f32_tn.hip $(CSRC)/vita_elementwise.hip $(CSRC)/vita_train.hip $(CSRC)/vita_loss_step.hip $(CSRC)/vita_api_bf16.hip $(CSRC)/vita_contrastive.hip $(CSRC)/vita_meanflow.hip $(CSRC)/vita_ode_persistent.hip $(CSRC)/vita_ode_resident.hip $(CSRC)/vita_data.hip $(CSRC)/vita_observer.hip
HDRS  := $(CSRC)/vita_common.h $(CSRC)/vita_host.h $(CSRC)/vita_gemm_epilogue.h $(CSRC)/vita_gemm_dma_tile.h $(CSRC)/vita_ode_sync.h $(CSRC)/vita_splitk_fused.h include/vita_hip.h
OBJS  := $(SRCS:.hip=.o)
LIB   := vita_amd/libvita_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function

all: $(LIB)

%.o: %.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(OBJS) $(LIB)



tools/bench_gemm: tools/bench_gemm.cpp $(LIB) include/vita_hip.h
	$(HIPCC) --offload-arch=$(ARCH) -O2 -std=c++17 tools/bench_gemm.cpp -o $@ -Lvita_amd -lvita_hip -Wl,-rpath,'$$ORIGIN/../vita_amd'
.PHONY: all clean
