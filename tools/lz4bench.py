#!/usr/bin/env python3
"""Device-side LZ4 (cloudini_amd/csrc/lz4_kernels.hip): device-resident encode with stage 2 on / off, 1 and 32 clouds of
1 M XYZI points; sizes next to the system liblz4 on the same payloads."""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cloudini_amd import native, synth

dev = torch.device("cuda", 0)
for wl, gen in (("c2 xyzi", lambda: synth.lidar_xyzi(1_000_000)), ("c3 depth rgba", lambda: synth.depthcam_xyzrgba(1280, 800))):
    info, data = gen()
    pts = data.size // info.point_step
    plan = native.Plan(info)
    for n_clouds in (1, 32 if wl.startswith("c2") else 16):
        codec = native.Codec(plan)
        d_in = torch.from_numpy(np.concatenate([data] * n_clouds)).to(dev)
        cp = np.full(n_clouds, pts, dtype=np.uint64)
        res = {}
        for stage2 in (0, 1, 2):
            codec.set_stage2(stage2)
            cap = plan.stage2_bound(pts, stage2) * n_clouds
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_off = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
            for _ in range(3):
                codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
            codec.synchronize()
            t0 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
            codec.synchronize()
            codec.status()
            res[stage2] = ((time.perf_counter() - t0) / reps, int(d_off.cpu()[-1]))
        (t0_, b0) = res[0]
        for mode, tag in ((1, "device LZ4"), (2, "device LZ4 FAST")):
            (t1_, b1) = res[mode]
            print(f"{wl} x{n_clouds}: stage 1 only {t0_*1e3:.3f} ms ({b0/n_clouds/pts:.3f} B/pt); + {tag} {t1_*1e3:.3f} ms "
                  f"({b1/n_clouds/pts:.3f} B/pt, ratio {b1/b0:.4f}) -> LZ4 part {1e3*(t1_-t0_):.3f} ms, {b0/(t1_-t0_)/1e9:.1f} GB/s of payload")
        codec.close()
