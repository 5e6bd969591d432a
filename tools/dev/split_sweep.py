#!/usr/bin/env python3
"""Device-resident decode of n x 1 M-point XYZI clouds (and 130 k-point Velodyne clouds): ms per call; run with
CLDN_HIP_NO_SPLIT_DECODE=1 / CLDN_HIP_SPLIT_PARTS=n to compare launch shapes."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth
dev = torch.device("cuda", 0)
out = []
for name, gen, counts in (("xyzi1M", lambda: synth.lidar_xyzi(1_000_000), (1, 2, 4, 8, 16)), ("xyz1M", lambda: synth.lidar_xyz(1_000_000), (1, 4)),
                          ("velo130k", lambda: synth.velodyne_xyzir(130048), (1, 8, 32))):
    info, data = gen()
    n = data.size // info.point_step
    plan = native.Plan(info)
    for n_clouds in counts:
        codec = native.Codec(plan, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
        d_points = torch.from_numpy(np.concatenate([data] * n_clouds)).to(dev)
        cp = np.full(n_clouds, n, dtype=np.uint64)
        cap = plan.stage1_bound(n) * n_clouds
        d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
        d_off = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
        n_chunks = n_clouds * ((n + 32767) // 32768)
        d_sizes = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
        codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)
        torch.cuda.synchronize()
        offs = d_off.cpu().numpy().astype(np.uint64)
        d_dec = torch.zeros(d_points.numel(), dtype=torch.uint8, device=dev)
        for _ in range(5):
            codec.decode_device(d_out.data_ptr(), offs, cp, d_dec.data_ptr(), d_dec.numel(), d_sizes.data_ptr())
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(20):
                codec.decode_device(d_out.data_ptr(), offs, cp, d_dec.data_ptr(), d_dec.numel(), d_sizes.data_ptr())
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 20)
        codec.status()
        out.append(f"{name}x{n_clouds}({n_chunks}ch) {np.median(ts)*1e3:.3f}")
        codec.close()
print(" | ".join(out))
