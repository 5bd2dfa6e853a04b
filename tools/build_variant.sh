#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: builds abtest/libsa_<name>.so from the tree with extra -D flags (for tools/ab.sh)
name=$1; shift
C=sequencealigner_amd/csrc
mkdir -p abtest
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -Iinclude "$@" -shared -pthread -o abtest/libsa_$name.so \
  $C/sa_kernels.hip $C/sa_kernels_pair.hip $C/sa_kernels_long.hip $C/sa_kernels_long_pair.hip $C/sa_kernels_coop.hip $C/sa_format.hip $C/sa_noalloc.hip $C/sa_api.cpp $C/sa_host.cpp
