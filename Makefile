# Builds the product library (HIP kernels + C ABI) for gfx950 and the test-only CPU checker.
#   make lib      sequencealigner_amd/libsa_hip.so
#   make oracle   oracle/libnw_oracle.so  (+ oracle/_ref when the reference sources are present)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := sequencealigner_amd/csrc
LIB     := sequencealigner_amd/libsa_hip.so
HIPFLAGS := --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function

.PHONY: all lib oracle clean
all: lib oracle

lib: $(LIB)
$(LIB): $(CSRC)/sa_kernels.hip $(CSRC)/sa_api.cpp $(CSRC)/sa_device.h include/sa_hip.h
	$(HIPCC) $(HIPFLAGS) -shared -o $@ $(CSRC)/sa_kernels.hip $(CSRC)/sa_api.cpp

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(LIB)
	$(MAKE) -C oracle clean
