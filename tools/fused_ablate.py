#!/usr/bin/env python3
"""Where the single-pass kernel spends its time (profiling only; ablated runs produce wrong bytes).
CLDN_HIP_ABLATE bits: 1 = no statistics pass, 2 = no inter-piece protocol (private output ranges), 4 = no column
stores. One process per setting (the library reads the variable once)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from cloudini_amd import native, synth
wl, clouds = sys.argv[1], int(sys.argv[2])
if wl == "c2": info, data = synth.lidar_xyzi(1_000_000)
elif wl == "c5": info, data = synth.lidar_xyz(1_000_000)
else: info, data = synth.velodyne_xyzir(130048)
n = len(data) // info.point_step
dev = torch.device("cuda", 0)
d_in = torch.from_numpy(np.concatenate([data] * clouds)).to(dev)
plan = native.Plan(info)
codec = native.Codec(plan, device=0)
codec.pipeline(int(sys.argv[3]))
cap = plan.stage1_bound(n) * clouds
d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
d_off = torch.zeros(clouds + 1, dtype=torch.int64, device=dev)
cp = np.full(clouds, n, dtype=np.uint64)
for _ in range(3):
    codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
codec.synchronize()
codec.enable_timing(10)
for _ in range(10):
    codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
codec.synchronize()
k = [codec.kernel_ms(s) for s in range(10)]
print({key: round(float(np.median([x[key] for x in k])), 4) for key in k[0]})
''' % ROOT
for wl, clouds in (("c2", 32), ("c2", 1), ("c5", 32), ("c5", 1)):
    for a in (0, 2, 8, 16, 32, 64, 8 + 64, 8 + 16 + 32 + 64, 16 + 32 + 64, 32 + 64):
        env = dict(os.environ, CLDN_HIP_ABLATE=str(a))
        r = subprocess.run([sys.executable, "-c", CHILD, wl, str(clouds), "3"], env=env, capture_output=True, text=True)
        print(wl, clouds, "ablate", a, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
