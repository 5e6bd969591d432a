#!/usr/bin/env python3
"""PCIe-inclusive rate: host buffers in, host buffers out (cldn_hip_encode_stage1 with HOST tags, pageable numpy
memory), 1 M-pt XYZI clouds. Never bench.py's `value`; quoted in DESIGN.md section 5."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cloudini_amd import native, synth

info, data = synth.lidar_xyzi(1_000_000)
codec = native.Codec(native.Plan(info))
for n_clouds in (1, 8, 32):
    clouds = [data] * n_clouds
    for _ in range(2):
        codec.encode_host(clouds)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        streams, _, _ = codec.encode_host(clouds)
    dt = (time.perf_counter() - t0) / reps
    print(f"host->host {n_clouds} x 1M XYZI: {dt*1e3:.2f} ms per call, {n_clouds/dt:.0f} Mpoints/s, "
          f"{n_clouds*16/dt/1e3:.1f} GB/s of input")
codec.close()
