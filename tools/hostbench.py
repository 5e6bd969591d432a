#!/usr/bin/env python3
"""PCIe-inclusive rates (never bench.py's `value`; quoted in DESIGN.md section 5), 1 M-pt XYZI clouds:
  a) cldn_hip_encode_stage1 with HOST tags on pageable numpy memory,
  b) the same call on pinned host memory (torch pin_memory),
  c) Cloudini::PointcloudEncoder::encode of the host mirror end to end with NONE / LZ4 / ZSTD (stage 2 on host threads),
     next to the compiled reference on one core when oracle/_ref is around."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import api, native, synth
from cloudini_amd.schema import CompressionOption

info, data = synth.lidar_xyzi(1_000_000)
codec = native.Codec(native.Plan(info))
for n_clouds in (1, 8):
    clouds = [data] * n_clouds
    for _ in range(2):
        codec.encode_host(clouds)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        codec.encode_host(clouds)
    dt = (time.perf_counter() - t0) / reps
    print(f"a) pageable host->host {n_clouds} x 1M XYZI: {dt*1e3:.2f} ms per call, {n_clouds/dt:.0f} Mpoints/s")

# b) pinned buffers straight through the C ABI
L = native.lib()
for n_clouds in (1, 8):
    src = torch.from_numpy(np.concatenate([data] * n_clouds)).pin_memory()
    cap = codec.plan.stage1_bound(1_000_000) * n_clouds
    dst = torch.empty(cap, dtype=torch.uint8).pin_memory()
    cp = np.full(n_clouds, 1_000_000, dtype=np.uint64)
    offs = np.zeros(n_clouds + 1, dtype=np.uint64)
    def call():
        rc = L.cldn_hip_encode_stage1(codec._h, C.c_void_p(src.data_ptr()), 0, cp.ctypes.data_as(C.POINTER(C.c_uint64)), n_clouds,
                                      C.c_void_p(dst.data_ptr()), cap, 0, offs.ctypes.data_as(C.c_void_p), None, None)
        assert rc == 0
    for _ in range(2):
        call()
    t0 = time.perf_counter()
    for _ in range(5):
        call()
    dt = (time.perf_counter() - t0) / 5
    print(f"b) pinned host->host {n_clouds} x 1M XYZI: {dt*1e3:.2f} ms per call, {n_clouds/dt:.0f} Mpoints/s "
          f"({n_clouds*16/dt/1e3:.1f} GB/s in, {int(offs[-1])/dt/1e9:.1f} GB/s out)")
codec.close()

# c) the host mirror end to end
try:
    from oracle.binding import RefLib
    ref = RefLib()
except (OSError, FileNotFoundError):
    ref = None
for comp in (CompressionOption.NONE, CompressionOption.LZ4, CompressionOption.ZSTD):
    inf = info.copy(compression_opt=comp, use_threads=True)
    enc = api.PointcloudEncoder(inf)
    for _ in range(2):
        out = enc.encode(data)
    t0 = time.perf_counter()
    for _ in range(5):
        out = enc.encode(data)
    dt = (time.perf_counter() - t0) / 5
    line = f"c) PointcloudEncoder::encode {comp.name}: {dt*1e3:.2f} ms per 1M-pt cloud = {1/dt:.0f} Mpoints/s, {len(out)/1e6:.2f} MB"
    if ref is not None:
        _size, t = ref.bench_encode(inf, data, reps=5, threads=1)
        line += f"; reference {float(np.median(t))*1e3:.2f} ms"
    print(line)

# d) Cloudini::PointcloudDecoder::decode of the host mirror, full streams
for comp in (CompressionOption.NONE, CompressionOption.LZ4, CompressionOption.ZSTD):
    inf = info.copy(compression_opt=comp, use_threads=True)
    stream = api.PointcloudEncoder(inf).encode(data)
    dec = api.PointcloudDecoder()
    for _ in range(2):
        out, _i = dec.decode_stream(stream)
    t0 = time.perf_counter()
    for _ in range(5):
        out, _i = dec.decode_stream(stream)
    dt = (time.perf_counter() - t0) / 5
    line = f"d) PointcloudDecoder::decode {comp.name}: {dt*1e3:.2f} ms per 1M-pt cloud = {1/dt:.0f} Mpoints/s"
    if ref is not None:
        _n, t = ref.bench_decode(stream, data.size, reps=5)
        line += f"; reference {float(np.median(t))*1e3:.2f} ms"
    print(line)
