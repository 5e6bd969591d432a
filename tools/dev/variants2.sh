#!/bin/bash
# A/B builds of one translation unit with arbitrary flag sets (always the DEVELOPMENT flavour, -DCLDN_DEV: the CLDN_HIP_*
# environment switches, the ablation hooks and the superseded kernel generations exist only there): tools/dev/variants2.sh <TU> name1:"-DX=1 -DY=0" name2:"..."
#   -> cloudini_amd/lib/variants/libcloudini_hip_<name>.so (the other objects come from build/hip: run cloudini_amd/build.py first)
# tools/dev/ab_variants.sh runs the default bench once per variant on the GPU box (same box, interleaved repeats).
R=/root/repo
TU=$1; shift
mkdir -p $R/cloudini_amd/lib/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -DCLDN_DEV $flags -c \
      -I$R/include -I$R/cloudini_amd/csrc $R/cloudini_amd/csrc/$TU.hip -o /tmp/var_${TU}_$name.o &&
    objs=""; for o in stage1_kernels stage1_decode viz_kernels lz4_kernels hip_abi; do
      if [ $o = $TU ]; then objs="$objs /tmp/var_${TU}_$name.o"; else objs="$objs $R/build/hip/$o.o"; fi; done
    hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/cloudini_amd/lib/variants/libcloudini_hip_$name.so && echo built $name ) &
done
wait
