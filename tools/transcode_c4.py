#!/usr/bin/env python3
"""BASELINE configs[3] end to end: 256 Velodyne-style XYZI+ring clouds of 130048 points (18-byte points) as CDR
PointCloud2 files -> cloudini_batch_transcode (ZSTD second stage on the host pool) -> CompressedPointCloud2 files,
next to the reference's per-message converter (oracle/_ref, one thread) on a sample of the same messages."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cloudini_amd import synth
_cdr_pointcloud2 = synth.cdr_pointcloud2

n_msgs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = sys.argv[2] if len(sys.argv) > 2 else None
distinct = [synth.velodyne_xyzir(130048, seed=42 + k) for k in range(4)]
msgs = [_cdr_pointcloud2(distinct[k % 4][0], distinct[k % 4][1], stamp=(1700000000, k)) for k in range(n_msgs)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    src, dst = os.path.join(tmp, "in"), os.path.join(tmp, "out")
    os.makedirs(src)
    for k, m in enumerate(msgs):
        m.tofile(os.path.join(src, f"msg_{k:05d}.bin"))
    exe = os.path.join(ROOT, "cloudini_amd", "lib", "cloudini_batch_transcode")
    env = dict(os.environ)
    if threads:
        env["CLOUDINI_AMD_STAGE2_THREADS"] = threads
    for rep in range(2):  # the second run has warm files and a warm GPU context
        t0 = time.perf_counter()
        r = subprocess.run([exe, src, dst, "--resolution", "0.001", "--compression", "zstd", "--batch", "32"], capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        st = json.loads(r.stdout.strip().splitlines()[-1])
        st["process_wall_s"] = round(wall, 3)
        st["stage2_threads"] = threads or "default (4)"
        print(json.dumps(st))
    # the way back: the compressed files through the decode direction
    back = os.path.join(tmp, "back")
    for rep in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([exe, dst, back, "--decode", "--batch", "32"], capture_output=True, text=True, env=env)
        wall = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        st = json.loads(r.stdout.strip().splitlines()[-1])
        st["process_wall_s"] = round(wall, 3)
        st["direction"] = "decode"
        print(json.dumps(st))
    try:
        from oracle.binding import RefLib
        ref = RefLib()
        t0 = time.perf_counter()
        for k in range(4):
            packed = np.fromfile(os.path.join(dst, f"msg_{k:05d}.bin"), dtype=np.uint8)
            want = ref.ros_decompress(packed, msgs[k].size + 4096)
            got = np.fromfile(os.path.join(back, f"msg_{k:05d}.bin"), dtype=np.uint8)
            assert np.array_equal(got, want), k
        per = (time.perf_counter() - t0) / 4
        print(json.dumps({"reference_decode_per_message_ms": round(per * 1e3, 2),
                          "reference_decode_Mpoints_per_s_1_thread": round(130048 / per / 1e6, 1)}))
    except (OSError, FileNotFoundError):
        pass
    try:
        from oracle.binding import RefLib
        ref = RefLib()
        t0 = time.perf_counter()
        sample = 8
        for k in range(sample):
            want = ref.ros_compress(msgs[k], 0.001, 2)
            got = np.fromfile(os.path.join(dst, f"msg_{k:05d}.bin"), dtype=np.uint8)
            assert np.array_equal(got, want), k
        per = (time.perf_counter() - t0) / sample
        print(json.dumps({"reference_per_message_ms": round(per * 1e3, 2), "reference_Mpoints_per_s_1_thread": round(130048 / per / 1e6, 1),
                          "checked_messages_equal": sample}))
    except (OSError, FileNotFoundError):
        pass
