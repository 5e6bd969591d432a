#!/usr/bin/env python3
"""32 x 1 M XYZI, device resident: stage 1 alone, + device LZ4, + device LZ4 FAST (ms per call, hash of the output)."""
import os, sys, time, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth
dev = torch.device("cuda", 0)
info, data = synth.lidar_xyzi(1_000_000)
pts = data.size // info.point_step
plan = native.Plan(info)
n_clouds = int(os.environ.get("CLOUDS", "32"))
codec = native.Codec(plan)
d_in = torch.from_numpy(np.concatenate([data] * n_clouds)).to(dev)
cp = np.full(n_clouds, pts, dtype=np.uint64)
out = []
for stage2 in (0, 1, 2):
    codec.set_stage2(stage2)
    cap = plan.stage2_bound(pts, stage2) * n_clouds
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
    for _ in range(3):
        codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
    codec.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(5):
            codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
        codec.synchronize()
        ts.append((time.perf_counter() - t0) / 5)
    codec.status()
    total = int(d_off.cpu()[-1])
    h = hashlib.md5(d_out[:total].cpu().numpy().tobytes()).hexdigest()[:8]
    out.append(f"s2={stage2} {np.median(ts)*1e3:.3f} ms {total} B {h}")
print(" | ".join(out))
