#!/bin/bash
# kernel resources (VGPRs, SGPRs, scratch, LDS) of a compiled .o / .so: tools/dev/kres.sh <file> [name filter]
F=$1; PAT=${2:-.}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $F $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | awk '/\.name:/ {n=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {g=$2} /\.vgpr_spill_count:/ {print n, "vgpr", v, "sgpr", s, "scratch", p, "lds", g, "vspill", $2}' | grep -E "$PAT" | c++filt | cut -c1-200
rm -rf $T
