#!/usr/bin/env python3
"""One case of tests/test_gpu_fuzz.py by seed, with a byte-level diagnosis when the GPU stream differs from the oracle's:
tools/dev/fuzz_case.py <seed>"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as T
from cloudini_amd import native
from oracle import binding
seed = int(sys.argv[1])
info, data = T._random_case(seed)
n = data.size // info.point_step
print("n", n, "step", info.point_step, "enc", int(info.encoding_opt), "v", info.version, [(f.name, int(f.type), f.offset, f.resolution) for f in info.fields])
o = binding.Oracle()
want, wm = o.encode_stage1(info, data, return_modes=True)
plan = native.Plan(info)
codec = native.Codec(plan)
streams, sizes, modes = codec.encode_host([data])
got = streams[0]
print("want", want.size, "got", got.size, "modes", list(wm), list(modes[0]), "chunk sizes", list(sizes)[:8])
m = min(want.size, got.size)
d = np.nonzero(want[:m] != got[:m])[0]
print("first diff at", int(d[0]) if d.size else None, "diffs", d.size)
# chunk framing of both
def chunks(s):
    out, p = [], 0
    while p + 4 <= s.size:
        z = int(s[p]) | int(s[p+1]) << 8 | int(s[p+2]) << 16 | int(s[p+3]) << 24
        out.append((p, z)); p += 4 + z
    return out
print("want chunks", chunks(want)[:6]); print("got chunks", chunks(got)[:6])
if d.size:
    i = int(d[0]); print("want", want[max(0,i-8):i+24].tolist()); print("got ", got[max(0,i-8):i+24].tolist())
wc, gc = chunks(want), chunks(got)
for k in range(min(len(wc), len(gc), 2)):
    a = want[wc[k][0] + 4: wc[k][0] + 4 + wc[k][1]]; b = got[gc[k][0] + 4: gc[k][0] + 4 + gc[k][1]]
    mm = min(a.size, b.size); dd = np.nonzero(a[:mm] != b[:mm])[0]
    j = int(dd[0]) if dd.size else mm
    print(f"chunk {k}: payloads {a.size} / {b.size}, first differing payload byte {j}")
    print("  want", a[max(0, j - 16): j + 32].tolist()); print("  got ", b[max(0, j - 16): j + 32].tolist())
# the u64 column around the point where it goes wrong is easier to see through the decoder: decode both with the oracle
try:
    dw = o.decode_stage1(info, want, n, fill=0)
    dg = o.decode_stage1(info, got, n, fill=0)
    bad = np.nonzero(dw.reshape(n, -1) != dg.reshape(n, -1))
    print("oracle decodes the GPU stream; differing points:", np.unique(bad[0])[:10], "bytes", np.unique(bad[1])[:16])
except Exception as e:
    print("oracle rejects the GPU stream:", e)
