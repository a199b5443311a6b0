# Builds libbuctd_hip.so (gfx950 only) and the C oracle helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := buctd_amd/csrc
OUT   := buctd_amd/lib/libbuctd_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result
OBJS := $(CSRC)/conv.o $(CSRC)/conv3x3.o $(CSRC)/conv3x3_lean.o $(CSRC)/conv3x3_pers.o $(CSRC)/conv_gather_x6.o $(CSRC)/conv_gather_wgrad.o $(CSRC)/conv3x3_wgrad.o $(CSRC)/matmul.o $(CSRC)/bn.o $(CSRC)/elementwise.o $(CSRC)/attention.o $(CSRC)/attn_smallqk.o $(CSRC)/attn_mha.o $(CSRC)/attn_mha_train.o $(CSRC)/gemm_x6.o $(CSRC)/block.o \
        $(CSRC)/loss_decode.o $(CSRC)/nms.o $(CSRC)/sample.o $(CSRC)/synth.o $(CSRC)/error.o

all: $(OUT)

$(CSRC)/%.o: $(CSRC)/%.hip $(CSRC)/common.h $(CSRC)/gemm_core.h $(CSRC)/c3_common.h $(CSRC)/c3_lean.h $(CSRC)/c3_pers.h $(CSRC)/bn_acc.h include/buctd_hip.h
	$(HIPCC) $(HIPFLAGS) -x hip -c $< -o $@

$(CSRC)/error.o: $(CSRC)/error.cpp
	$(HIPCC) -O2 -std=c++17 -fPIC -c $< -o $@

$(OUT): $(OBJS)
	@mkdir -p buctd_amd/lib
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS)

clean:
	rm -f $(CSRC)/*.o $(OUT)

.PHONY: all clean
