#!/bin/bash
# compile one HIP translation unit (default: the decode TU) with the build's flags, keep the ISA, print the register
# figures of the kernels whose mangled name matches $2 (default: k_decode_points_w)
R=/root/repo
TU=${1:-stage1_decode}
PAT=${2:-k_decode_points_w}
mkdir -p $R/build/hip
cd $R/build/hip || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -c \
  -I$R/include -I$R/cloudini_amd/csrc $R/cloudini_amd/csrc/$TU.hip -o $R/build/hip/$TU.o -MD -MF $R/build/hip/$TU.o.d -save-temps=obj 2>&1 | grep -v "^$" | head -40
S=$R/build/hip/$TU-hip-amdgcn-amd-amdhsa-gfx950.s
grep -n "\.name:.*$PAT" $S | while IFS=: read l rest; do
  sed -n "$((l-25)),$((l+12))p" $S | grep -E "vgpr_count|sgpr_count|private_segment_fixed|spill" | tr -s ' ' | tr '\n' ' '
  echo "$rest" | sed 's/.*\(k_[a-z_0-9]*[A-Za-z0-9]*\)EvNS.*/\1/' | cut -c1-60
done
