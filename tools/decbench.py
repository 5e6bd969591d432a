#!/usr/bin/env python3
"""Decode timing (device resident): BASELINE configs C2-C5 (argument: prefix of the case name, e.g. c3)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth

dev = torch.device("cuda", 0)
only = sys.argv[1] if len(sys.argv) > 1 else ""
cases = (("c5 xyz 10M", lambda: synth.lidar_xyz(10_000_000), 1), ("c2 xyzi 32x1M", lambda: synth.lidar_xyzi(1_000_000), 32),
         ("c3 xyzrgba 16x1M", lambda: synth.depthcam_xyzrgba(1280, 800), 16), ("c4 velodyne 256x130k", lambda: synth.velodyne_xyzir(130048), 256),
         ("s3 xyzrgba 1x1M", lambda: synth.depthcam_xyzrgba(1280, 800), 1), ("s4 velodyne 1x130k", lambda: synth.velodyne_xyzir(130048), 1))
for name, make, n_clouds in cases:
    if only and not name.startswith(only):
        continue
    info, data = make()
    distinct = int(os.environ.get("DECBENCH_DISTINCT", "1"))  # the cases' generators take a seed: tile several clouds
    datas = [data]
    if distinct > 1 and n_clouds > 1:
        gen = {"c2": synth.lidar_xyzi, "c4": synth.velodyne_xyzir}.get(name[:2])
        if gen is not None:
            datas = [gen(data.size // info.point_step, seed=42 + k)[1] for k in range(distinct)]
    plan = native.Plan(info)
    codec = native.Codec(plan, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
    if os.environ.get("DECBENCH_FILL_ZERO", "0") != "0":  # CLDN_HIP_FILL_ZERO: bytes no field covers may be written as 0
        codec.set_decode_fill(True)
    if os.environ.get("DECBENCH_CLOUDS") and n_clouds > 1:
        n_clouds = int(os.environ["DECBENCH_CLOUDS"])
    step = info.point_step
    n = data.size // step
    host = np.concatenate([datas[k % len(datas)] for k in range(n_clouds)])
    d_points = torch.from_numpy(host).to(dev)
    cloud_points = np.full(n_clouds, n, dtype=np.uint64)
    cap = plan.stage1_bound(n) * n_clouds
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
    n_chunks = int(sum((int(x) + 32767) // 32768 for x in cloud_points))
    d_sizes = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    codec.encode_device(d_points.data_ptr(), cloud_points, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)
    sized = d_sizes.data_ptr() if os.environ.get("DECBENCH_SIZED", "1") != "0" else 0
    torch.cuda.synchronize()
    offs = d_off.cpu().numpy().astype(np.uint64)
    d_dec = torch.zeros(host.size, dtype=torch.uint8, device=dev)
    for it in range(5):
        codec.decode_device(d_out.data_ptr(), offs, cloud_points, d_dec.data_ptr(), host.size, sized)
    torch.cuda.synchronize()
    ts = []
    for blk in range(7):
        t0 = time.perf_counter()
        reps = 10
        for it in range(reps):
            codec.decode_device(d_out.data_ptr(), offs, cloud_points, d_dec.data_ptr(), host.size, sized)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    dt = float(np.median(ts))
    codec.status()
    print(f"{name}: decode median {dt*1e3:.3f} ms (min {min(ts)*1e3:.3f}, max {max(ts)*1e3:.3f}) -> {n*n_clouds/dt/1e6:.0f} Mpoints/s, stream {int(offs[-1])/1e6:.1f} MB, stats {codec.decode_stats()}")
    codec.close()
