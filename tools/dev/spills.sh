#!/bin/bash
# register spills of the kernels of a compiled object: tools/dev/spills.sh <file.o> <mangled-name filter>
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $1 $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | awk '/\.name:/ {n=$2} /\.private_segment_fixed_size:/ {p=$2} /\.sgpr_count:/ {s=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {v=$2} /\.vgpr_spill_count:/ {print n, "vgpr", v, "sgpr", s, "sgpr_spill", ss, "vgpr_spill", $2, "scratch", p}' | grep -E "${2:-.}"
rm -rf $T
