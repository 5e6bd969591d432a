#!/bin/bash
# A/B builds of one translation unit: tools/dev/variants.sh <TU> <MACRO> <v1> <v2> ... -> cloudini_amd/lib/variants/libcloudini_hip_<MACRO>_<v>.so
# (the other objects come from build/hip: run cloudini_amd/build.py first). A run script copies a variant over
# cloudini_amd/lib/libcloudini_hip.so on the GPU box before it measures.
R=/root/repo
TU=$1; MACRO=$2; shift 2
mkdir -p $R/cloudini_amd/lib/variants
for v in "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -D$MACRO=$v -c \
      -I$R/include -I$R/cloudini_amd/csrc $R/cloudini_amd/csrc/$TU.hip -o /tmp/var_${TU}_$v.o &&
    objs=""; for o in stage1_kernels stage1_decode viz_kernels lz4_kernels hip_abi; do
      if [ $o = $TU ]; then objs="$objs /tmp/var_${TU}_$v.o"; else objs="$objs $R/build/hip/$o.o"; fi; done
    hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/cloudini_amd/lib/variants/libcloudini_hip_${MACRO}_$v.so && echo built $v ) &
done
wait
