#!/bin/bash
# device ISA of one kernel of the decode TU: tools/dev/kisa.sh "<flags>" <mangled-name regex> <out.s>
R=/root/repo
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function $1 --cuda-device-only -S -I$R/include -I$R/cloudini_amd/csrc $R/cloudini_amd/csrc/${TU:-stage1_decode}.hip -o /tmp/kisa_full.s 2>&1 | grep -i "error" | head
S=/tmp/kisa_full.s
start=$(grep -n "^$2.*:" $S | head -1 | cut -d: -f1)
end=$(awk -v s=$start 'NR>s && /^\.Lfunc_end/ {print NR; exit}' $S)
sed -n "${start},${end}p" $S > $3
echo "$3: $(wc -l < $3) lines, VALU $(grep -c '^\s*v_' $3), SALU $(grep -c '^\s*s_' $3)"
