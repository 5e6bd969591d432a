#!/usr/bin/env python3
"""applyVizLossyPreprocessing: device-resident rate of cldn_hip_viz_preprocess against the reference on one host core."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth
from cloudini_amd.schema import PointField

dev = torch.device("cuda", 0)
for n, res in ((1_000_000, 0.001), (1_000_000, 0.05), (10_000_000, 0.01)):
    info, data = synth.lidar_xyzi(n, seed=5)
    info = info.copy(fields=[PointField(f.name, f.offset, f.type, res) if i < 3 else f for i, f in enumerate(info.fields)])
    codec = native.Codec(native.Plan(info), device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
    d_in = torch.from_numpy(data).to(dev)
    d_out = torch.empty(data.size, dtype=torch.uint8, device=dev)
    for _ in range(2):
        kept = codec.viz_preprocess_device(d_in.data_ptr(), n, 16, 0, res, d_out.data_ptr(), data.size)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        kept = codec.viz_preprocess_device(d_in.data_ptr(), n, 16, 0, res, d_out.data_ptr(), data.size)
    dt = (time.perf_counter() - t0) / reps
    line = f"{n} pts @ {res}: kept {kept} ({100*kept/n:.1f} %), {dt*1e3:.3f} ms per call = {n/dt/1e6:.0f} Mpoints/s"
    try:
        from oracle.binding import RefLib
        ref = RefLib()
        t0 = time.perf_counter()
        out, _, _, _ = ref.viz_preprocess(info, data)
        tr = time.perf_counter() - t0
        assert len(out) // 16 == kept
        line += f"; reference on one core {tr*1e3:.1f} ms = {n/tr/1e6:.1f} Mpoints/s"
    except (OSError, FileNotFoundError):
        pass
    print(line)
    codec.close()
