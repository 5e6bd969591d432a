import sys; sys.path.insert(0,'.')
import numpy as np
from cloudini_amd import native, synth
from oracle.binding import Oracle
o=Oracle()
for n in (1,2,3,62,63,64,65,130):
    info,data=synth.lidar_xyz(n)
    c=native.Codec(native.Plan(info))
    s,cs,m=c.encode_host([data])
    w=o.encode_stage1(info,data)
    ok=np.array_equal(s[0],w)
    print(n, ok, len(s[0]), len(w))
    if not ok:
        print(" got ", s[0][:48].tobytes().hex()); print(" want", w[:48].tobytes().hex())
    c.close()
