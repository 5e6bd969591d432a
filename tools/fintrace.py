#!/usr/bin/env python3
"""Phase timeline of k_finish's workgroups on the C2 batch (profiling hook CLDN_HIP_FINISH_TRACE, wall_clock64 = 100 MHz)."""
import sys, os, ctypes as C
import numpy as np
os.environ["CLDN_HIP_FINISH_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth

dev = torch.device("cuda", 0)
N_CLOUDS, N = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 1_000_000
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
n_chunks = N_CLOUDS * ((N + 32767) // 32768)
for it in range(20):
    codec.encode_device(d_points.data_ptr(), cp, d_out.data_ptr(), bound * N_CLOUDS, d_off.data_ptr(), 0, 0)
torch.cuda.synchronize()
tr = np.zeros((n_chunks, 16), dtype=np.uint64)
rc = native.lib().cldn_hip_debug_finish_trace(tr.ctypes.data_as(C.c_void_p), C.c_uint32(n_chunks))
assert rc == 0, rc
t = (tr.astype(np.int64) - int(tr[:, 0].min())) / 100.0  # us since the first workgroup's start


def line(name, v):
    print(f"{name:34s} mean {v.mean():7.2f}  p10 {np.percentile(v, 10):7.2f}  p50 {np.percentile(v, 50):7.2f}  p90 {np.percentile(v, 90):7.2f}  max {v.max():7.2f}")


even = np.arange(n_chunks) % 2 == 0
line("start (us after the first)", t[:, 0])
line("  table cleared after", t[:, 8] - t[:, 0])
line("  seed took", t[:, 9] - t[:, 8])
line("  pass 1 took", t[:, 1] - t[:, 9])
line("build done at", t[:, 1])
line("  build duration", t[:, 1] - t[:, 0])
line("position known at", t[:, 2])
line("  wait for earlier records", t[:, 2] - t[:, 1])
line("item table done at", t[:, 7])
line("header written at", t[:, 3])
line("odd: copy first, duration", (t[:, 4] - t[:, 3])[~even])
line("section duration (even)", (t[:, 5] - t[:, 4])[even])
line("section duration (odd)", (t[:, 5] - t[:, 4])[~even])
line("even: copy second, duration", (t[:, 6] - t[:, 5])[even])
line("end at", t[:, 6])
full = np.array([(N - k * 32768) >= 32768 for k in range((N + 32767) // 32768)] * N_CLOUDS)
line("  build duration, ragged chunks", (t[:, 1] - t[:, 0])[~full])
for c in [k for k in (0, 1, 2, 3, 500, 501, 990, 991) if k < n_chunks]:
    print(c, np.round(t[c, :8], 2))
codec.close()
