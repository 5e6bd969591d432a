#!/usr/bin/env python3
"""Where the batch transcoder's wall time goes (CLDN_HOST_TIMING: source / GPU stages / stage 2 / sink to stderr):
C4 messages in /dev/shm, in process, ZSTD; argument: stage-2 threads (default: the library's)."""
import os, sys, tempfile, time
os.environ["CLDN_HOST_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cloudini_amd import api, synth
from cloudini_amd.schema import CompressionOption
if len(sys.argv) > 1:
    api.set_stage2_threads(int(sys.argv[1]))
distinct = [synth.velodyne_xyzir(130048, seed=42 + k) for k in range(4)]
msgs = [synth.cdr_pointcloud2(distinct[k % 4][0], distinct[k % 4][1], stamp=(1700000000, k)) for k in range(256)]
with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
    src, dst = os.path.join(tmp, "in"), os.path.join(tmp, "out")
    os.makedirs(src)
    for k, m in enumerate(msgs):
        m.tofile(os.path.join(src, f"msg_{k:05d}.bin"))
    for rep in range(3):
        t0 = time.perf_counter()
        st = api.transcode_directory(src, dst, resolution=0.001, compression_opt=int(CompressionOption.ZSTD), batch_messages=32)
        print(f"rep {rep}: {time.perf_counter() - t0:.3f} s, {st['points'] / (time.perf_counter() - t0) / 1e6:.0f} Mpoints/s, stage-2 threads {api.stage2_threads()}", flush=True)
    back = os.path.join(tmp, "back")
    for rep in range(3):
        t0 = time.perf_counter()
        st = api.decode_directory(dst, back, batch_messages=32)
        print(f"decode rep {rep}: {time.perf_counter() - t0:.3f} s, {st['points'] / (time.perf_counter() - t0) / 1e6:.0f} Mpoints/s", flush=True)
