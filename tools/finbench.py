#!/usr/bin/env python3
"""C2 encode step (32 x 1 M XYZI, framed) with per-kernel device times, no checks: for A/B runs of switches whose output is
deliberately wrong (CLDN_HIP_FINISH_ABLATE) or already covered by the tests (CLDN_HIP_FINISH_COPY)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth

dev = torch.device("cuda", 0)
N_CLOUDS, N = int(os.environ.get("FINBENCH_CLOUDS", "32")), 1_000_000
info, _ = synth.lidar_xyzi(N)
datas = [synth.lidar_xyzi(N, seed=42 + k)[1] for k in range(4)]
plan = native.Plan(info)
codec = native.Codec(plan, device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
host = np.concatenate([datas[k % 4] for k in range(N_CLOUDS)])
d_points = torch.from_numpy(host).to(dev)
bound = plan.stage1_bound(N)
d_out = torch.empty(bound * N_CLOUDS, dtype=torch.uint8, device=dev)
d_off = torch.zeros(N_CLOUDS + 8, dtype=torch.int64, device=dev)
cp = np.full(N_CLOUDS, N, dtype=np.uint64)
STEPS = 20
codec.enable_timing(STEPS)
for it in range(20):
    codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), bound * N_CLOUDS, d_off.data_ptr(), 0, 0)
torch.cuda.synchronize()
ts, ks = [], []
for blk in range(5):
    t0 = time.perf_counter()
    for it in range(STEPS):
        codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), bound * N_CLOUDS, d_off.data_ptr(), 0, 0)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / STEPS)
    ks += [codec.kernel_ms(s) for s in range(STEPS)]
med = lambda k: float(np.median([x[k] for x in ks]))
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("CLDN_HIP_"))
print(f"[{tag}]".ljust(40), f"step {np.median(ts) * 1e3:.4f} ms (min {min(ts) * 1e3:.4f}); regular {med('regular'):.4f} sections {med('sections'):.4f} "
      f"finish {med('compact'):.4f} all {med('total'):.4f}", flush=True)
codec.close()
