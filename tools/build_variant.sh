#!/bin/bash
# tools/build_variant.sh <name> <source file under csrc> [extra hipcc flags]: abtest/libsa_<name>.so = the tree's library
# with that one source recompiled with the extra flags (for A/B timing through SA_HIP_LIBRARY); `make lib` first.
name=$1; src=$2; shift 2
mkdir -p abtest build/var_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c -o build/var_$name/$src.o sequencealigner_amd/csrc/$src || exit 1
objs=$(ls build/lib/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread -o abtest/libsa_$name.so $objs build/var_$name/$src.o
