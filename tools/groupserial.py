#!/usr/bin/env python3
"""Group-serial stage 1: the C2 batch (32 x 1 M XYZI) encoded as G back-to-back calls of 32/G
clouds on one stream.  A call addresses its slot workspace by the chunk's index within the call, so G calls reuse the
first 1/G of the workspace: with G >= 4 the slots a group writes (<= 60 MB) and k_finish reads back should sit in the
256 MB Infinity Cache.  Prints ms per 32-cloud batch for each G."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth

dev = torch.device("cuda", 0)
N_CLOUDS, N = 32, 1_000_000
info, _ = synth.lidar_xyzi(N)
datas = [synth.lidar_xyzi(N, seed=42 + k)[1] for k in range(4)]
plan = native.Plan(info)
codec = native.Codec(plan, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
host = np.concatenate([datas[k % 4] for k in range(N_CLOUDS)])
d_points = torch.from_numpy(host).to(dev)
bound = plan.stage1_bound(N)
d_out = torch.empty(bound * N_CLOUDS, dtype=torch.uint8, device=dev)
d_off = torch.zeros(N_CLOUDS + 8, dtype=torch.int64, device=dev)
step = info.point_step


def batch(groups: int):
    per = N_CLOUDS // groups
    cp = np.full(per, N, dtype=np.uint64)
    for g in range(groups):
        codec.encode_device(d_points.data_ptr() + g * per * N * step, cp, d_out.data_ptr() + g * per * bound, per * bound,
                            d_off.data_ptr(), 0, 0)


for groups in (1, 2, 4, 8, 16, 32):
    for it in range(10):
        batch(groups)
    torch.cuda.synchronize()
    ts = []
    for blk in range(7):
        t0 = time.perf_counter()
        for it in range(10):
            batch(groups)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 10)
    codec.status()
    dt = float(np.median(ts))
    print(f"groups {groups:2d} ({N_CLOUDS // groups:2d} clouds per call): {dt * 1e3:.3f} ms per 32 M points "
          f"(min {min(ts) * 1e3:.3f}) -> {N_CLOUDS * N / dt / 1e6:.0f} Mpoints/s", flush=True)
codec.close()
