#!/usr/bin/env python3
"""Would a captured HIP graph per call shape pay for one-cloud calls? The encode / decode call of one 1 M-point cloud is
captured on a side stream with torch.cuda.CUDAGraph (relaxed capture mode) and replayed; compared with plain calls, both as a
tight loop (throughput) and with a synchronisation per call (latency)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "xyzi"
info, data = {"xyzi": lambda: synth.lidar_xyzi(1_000_000), "velo": lambda: synth.velodyne_xyzir(130048)}[which]()
n = data.size // info.point_step
plan = native.Plan(info)
side = torch.cuda.Stream(dev)
codec = native.Codec(plan, device=0, stream=side.cuda_stream)
d_points = torch.from_numpy(data).to(dev)
cp = np.full(1, n, dtype=np.uint64)
cap = plan.stage1_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
d_off = torch.zeros(2, dtype=torch.int64, device=dev)
n_chunks = (n + 32767) // 32768
d_sizes = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
d_dec = torch.zeros(d_points.numel(), dtype=torch.uint8, device=dev)
def enc():
    codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)
for _ in range(5): enc()
torch.cuda.synchronize()
offs = d_off.cpu().numpy().astype(np.uint64)
def dec():
    codec.decode_device(d_out.data_ptr(), offs, cp, d_dec.data_ptr(), d_dec.numel(), d_sizes.data_ptr())
for _ in range(5): dec()
torch.cuda.synchronize()

def timed(fn, reps=200, sync_each=False):
    ts = []
    for blk in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
            if sync_each: torch.cuda.synchronize()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    return float(np.median(ts)) * 1e3

for name, fn in (("encode", enc), ("decode", dec)):
    plain_loop, plain_sync = timed(fn), timed(fn, sync_each=True)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
            fn()
        graph_loop, graph_sync = timed(g.replay), timed(g.replay, sync_each=True)
        print(f"{which} {name}: plain loop {plain_loop:.4f} ms, per-call sync {plain_sync:.4f} | graph loop {graph_loop:.4f} ms, per-call sync {graph_sync:.4f}")
    except Exception as e:
        print(f"{which} {name}: plain loop {plain_loop:.4f} ms, per-call sync {plain_sync:.4f} | capture failed: {str(e)[:200]}")
codec.status()
print("equal", bool(torch.equal(d_dec, d_points)))
