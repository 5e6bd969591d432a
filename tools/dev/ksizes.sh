#!/bin/bash
# device code bytes per kernel family of the built objects (build/hip/*.o): tools/dev/ksizes.sh [top n]
T=$(mktemp -d)
for o in /root/repo/build/hip/*.o; do
  objcopy -O binary --only-section=.hip_fatbin $o $T/fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co 2>/dev/null
  /opt/rocm/lib/llvm/bin/llvm-readelf -s --wide $T/k.co 2>/dev/null | awk '$4 == "FUNC" && $3 > 0 {print $3, $8}'
done | python3 -c "
import sys, subprocess, re
rows = [(int(l.split()[0]), l.split()[1]) for l in sys.stdin if len(l.split()) == 2]
names = subprocess.run(['c++filt'], input='\n'.join(r[1] for r in rows), capture_output=True, text=True).stdout.split('\n')
agg = {}
for (sz, _), n in zip(rows, names):
    fam = re.sub(r'<.*', '', re.sub(r'\(.*', '', n)).replace('void ', '')
    agg.setdefault(fam, [0, 0]); agg[fam][0] += sz; agg[fam][1] += 1
print('kernels', len(rows), 'device code bytes', sum(r[0] for r in rows))
for k, v in sorted(agg.items(), key=lambda x: -x[1][0])[:int('${1:-40}')]: print('%9d %3d %s' % (v[0], v[1], k))
"
rm -rf $T
