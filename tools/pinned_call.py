import ctypes as C, numpy as np, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cloudini_amd import native, synth
info, cloud = synth.lidar_xyzi(1_000_000)
plan = native.Plan(info); codec = native.Codec(plan); L = native.lib()
cap = plan.stage1_bound(1_000_000); cp = np.array([1_000_000], dtype=np.uint64); offs = np.zeros(2, dtype=np.uint64)
src = torch.from_numpy(cloud).pin_memory(); dst = torch.empty(cap, dtype=torch.uint8).pin_memory()
for _ in range(6):
    rc = L.cldn_hip_encode_stage1(codec._h, C.c_void_p(src.data_ptr()), 0, cp.ctypes.data_as(C.POINTER(C.c_uint64)), 1, C.c_void_p(dst.data_ptr()), cap, 0, offs.ctypes.data_as(C.c_void_p), None, None)
    assert rc == 0
