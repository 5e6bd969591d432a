#!/usr/bin/env python3
"""Encode rate (device resident) for a few real-world point layouts that are not BASELINE configs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cases
from cloudini_amd import native, synth
from cloudini_amd.schema import FieldType as F

dev = torch.device("cuda", 0)
n = 1_000_000
_, xyz = synth.lidar_xyz(n, seed=3)
p = xyz.view(np.float32).reshape(n, 3)
rs = np.random.RandomState(4)
inten = (rs.randint(0, 256, n)).astype(np.float32)
layouts = {
    "pcl_xyzi_step32 (x y z pad intensity@16)": ([("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001),
                                                   ("intensity", 16, F.FLOAT32, 0.001)], 32,
                                                  {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": inten}),
    "ouster_step48": ([("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001),
                       ("intensity", 16, F.FLOAT32, 0.001), ("t", 20, F.UINT32, None), ("reflectivity", 24, F.UINT16, None),
                       ("ring", 26, F.UINT16, None), ("ambient", 28, F.UINT16, None), ("range", 32, F.UINT32, None)], 48,
                      {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": inten,
                       "t": (np.arange(n) * 97).astype(np.uint32), "reflectivity": rs.randint(0, 256, n).astype(np.uint16),
                       "ring": (np.arange(n) % 64).astype(np.uint16), "ambient": rs.randint(0, 2000, n).astype(np.uint16),
                       "range": (np.linalg.norm(p, axis=1) * 1000).astype(np.uint32)}),
    "xyz_rgb_f32packed_step16 (rgb as FLOAT32 without resolution -> Copy)": (
        [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001), ("rgb", 12, F.FLOAT32, None)], 16,
        {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "rgb": rs.randint(0, 1 << 24, n).astype(np.uint32).view(np.float32)}),
    "dds_sample_layout_step26 (XYZI f32 + ring u16 + f64 stamp, Gorilla)": (
        [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001), ("intensity", 12, F.FLOAT32, 0.001),
         ("ring", 16, F.UINT16, None), ("timestamp", 18, F.FLOAT64, None)], 26,
        {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": inten, "ring": (np.arange(n) % 64).astype(np.uint16),
         "timestamp": 1.7e9 + np.arange(n) * 1e-5}),
    "dds_sample_layout with 1 us stamps (after applyVizLossyPreprocessing)": (
        [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001), ("intensity", 12, F.FLOAT32, 0.001),
         ("ring", 16, F.UINT16, None), ("timestamp", 18, F.FLOAT64, 1e-6)], 26,
        {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": inten, "ring": (np.arange(n) % 64).astype(np.uint16),
         "timestamp": 1.7e9 + np.arange(n) * 1e-5}),
}
from cloudini_amd.schema import EncodingOptions
layouts["xyzi_f32_lossless (EncodingOptions::LOSSLESS: XOR-coded floats)"] = (
    [("x", 0, F.FLOAT32, None), ("y", 4, F.FLOAT32, None), ("z", 8, F.FLOAT32, None), ("intensity", 12, F.FLOAT32, None)], 16,
    {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": inten})
for name, (fields, step, cols) in layouts.items():
    if os.environ.get("SCHEMABENCH_ONLY") and os.environ["SCHEMABENCH_ONLY"] not in name:
        continue
    info = cases.make_info(fields, step, n, enc=EncodingOptions.LOSSLESS) if "lossless" in name else cases.make_info(fields, step, n)
    data = cases.pack(info, cols, n)
    n_clouds = int(os.environ.get("SCHEMABENCH_CLOUDS", "16"))
    d_points = torch.from_numpy(np.concatenate([data] * n_clouds)).to(dev)
    plan = native.Plan(info)
    codec = native.Codec(plan, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
    cap = plan.stage1_bound(n) * n_clouds
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
    cp = np.full(n_clouds, n, dtype=np.uint64)
    for _ in range(3):
        codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    codec.enable_timing(10)
    t0 = time.perf_counter()
    for _ in range(10):
        codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    km = codec.kernel_ms(0)
    print(f"{name}: {n_clouds*n/dt/1e6:.0f} Mpoints/s ({dt*1e3:.3f} ms; regular {km['regular']:.3f}, sections {km['sections']:.3f}, compact {km['compact']:.3f}), {int(d_off.cpu()[-1])/(n_clouds*n):.2f} B/pt")
    # and back (device resident, the walk over the chunk prefixes included)
    offs = d_off.cpu().numpy().astype(np.uint64)
    d_dec = torch.zeros(d_points.numel(), dtype=torch.uint8, device=dev)
    for _ in range(3):
        codec.decode_device(d_out.data_ptr(), offs, cp, d_dec.data_ptr(), d_dec.numel(), 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        codec.decode_device(d_out.data_ptr(), offs, cp, d_dec.data_ptr(), d_dec.numel(), 0)
    torch.cuda.synchronize()
    ddt = (time.perf_counter() - t0) / 10
    codec.status()
    ok = bool(torch.equal(d_dec, d_points)) if all(f[3] is None for f in fields) else None
    print(f"    decode {n_clouds*n/ddt/1e6:.0f} Mpoints/s ({ddt*1e3:.3f} ms), chunks (parallel regular, parallel sections, serial, serial sections) = {codec.decode_stats()}")
    codec.close()
