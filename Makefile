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

MAIN    := sequencealigner_amd/sa_main
TOOLS   := sequencealigner_amd/sa_tune sequencealigner_amd/sa_dataset sequencealigner_amd/sa_pk_max3_check

lib: $(LIB) $(MAIN) $(TOOLS)
sequencealigner_amd/sa_tune: $(CSRC)/sa_tune.cpp $(LIB)
	$(HIPCC) -O2 -std=c++17 -Wall -o $@ $(CSRC)/sa_tune.cpp -Lsequencealigner_amd -lsa_hip -Wl,-rpath,'$$ORIGIN'
sequencealigner_amd/sa_pk_max3_check: $(CSRC)/sa_pk_max3_check.hip
	$(HIPCC) --offload-arch=$(ARCH) -O3 -Wno-unused-result -o $@ $(CSRC)/sa_pk_max3_check.hip
sequencealigner_amd/sa_dataset: $(CSRC)/sa_dataset.c
	gcc -O2 -Wall -Wextra -o $@ $(CSRC)/sa_dataset.c
$(MAIN): $(CSRC)/sa_main.cpp $(LIB)
	$(HIPCC) -O2 -std=c++17 -Wall -o $@ $(CSRC)/sa_main.cpp -Lsequencealigner_amd -lsa_hip -Wl,-rpath,'$$ORIGIN'
# one object per source, so that a change to one kernel recompiles one file
OBJDIR  := build/lib
LIBSRC  := sa_kernels.hip sa_kernels_pair.hip sa_kernels_long.hip sa_kernels_long_pair.hip sa_kernels_long_ckpt.hip \
           sa_kernels_coop.hip sa_format.hip sa_noalloc.hip sa_compact.hip sa_api.cpp sa_host.cpp
LIBOBJ  := $(addprefix $(OBJDIR)/,$(addsuffix .o,$(LIBSRC)))
LIBHDR  := $(CSRC)/sa_kernel_common.h $(CSRC)/sa_traceback.h $(CSRC)/sa_pair_common.h $(CSRC)/sa_parallel.h $(CSRC)/sa_device.h \
           include/sa_hip.h include/sa_host.h
$(OBJDIR)/%.o: $(CSRC)/% $(LIBHDR)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) -c -o $@ $<
$(LIB): $(LIBOBJ)
	$(HIPCC) --offload-arch=$(ARCH) -fPIC -shared -pthread -o $@ $(LIBOBJ)

oracle:
	$(MAKE) -C oracle all

clean:
	rm -f $(LIB) $(MAIN) $(TOOLS)
	$(MAKE) -C oracle clean
