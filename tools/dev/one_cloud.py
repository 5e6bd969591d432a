#!/usr/bin/env python3
"""One cloud per call, device resident: encode and decode loops (for kernel traces of the small-call shapes)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "xyzi"
info, data = {"xyzi": lambda: synth.lidar_xyzi(1_000_000), "velo": lambda: synth.velodyne_xyzir(130048), "rgba": lambda: synth.depthcam_xyzrgba(1280, 800)}[which]()
n = data.size // info.point_step
plan = native.Plan(info)
codec = native.Codec(plan, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
d_points = torch.from_numpy(data).to(dev)
cp = np.full(1, n, dtype=np.uint64)
cap = plan.stage1_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
d_off = torch.zeros(2, dtype=torch.int64, device=dev)
n_chunks = (n + 32767) // 32768
d_sizes = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
def enc():
    codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)
for _ in range(5): enc()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): enc()
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / 50
offs = d_off.cpu().numpy().astype(np.uint64)
d_dec = torch.zeros(d_points.numel(), dtype=torch.uint8, device=dev)
def dec():
    codec.decode_device(d_out.data_ptr(), offs, cp, d_dec.data_ptr(), d_dec.numel(), d_sizes.data_ptr())
for _ in range(5): dec()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): dec()
torch.cuda.synchronize()
td = (time.perf_counter() - t0) / 50
codec.status()
print(f"{which}: {n} points, encode {te*1e3:.4f} ms, decode {td*1e3:.4f} ms, equal {bool(torch.equal(d_dec, d_points)) if which != 'x' else None}")
