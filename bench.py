#!/usr/bin/env python3
"""bench.py -- stage-1 encode throughput of the HIP path on BASELINE.json's headline configuration.

A "step" = one cldn_hip_encode_stage1 call over one batch of synthetic clouds that is already resident in
HBM (inputs and outputs are device buffers; nothing crosses PCIe inside the timed region). The default
workload is BASELINE.json configs[1]: 1M-point XYZI (float32 XYZ + uint16 intensity) clouds at 1 mm,
stage 1 only ("pre-ZSTD"), `--clouds` of them per step and per GPU.

Multi-GPU, one process per GPU over RCCL:
  --shard clouds (default)  weak scaling: every rank encodes its own batch of `--clouds` whole clouds; clouds are
                            independent, so there is no data-path collective (RCCL carries the barrier, the
                            per-cloud size all-gather and the max-reduction of the elapsed time).
  --shard chunks            strong scaling: ONE cloud (`--clouds 1`, e.g. --workload c5) is cut into contiguous ranges
                            of whole 32768-point chunks; per step rank 0 probes the adaptive-int modes on the
                            cloud's head, broadcasts those bytes, every rank encodes its range with the modes forced,
                            and an all-gather of the byte counts gives each rank its offset in the stream.
`python bench.py --gpus N` launches its own N ranks (torch.distributed.run, 127.0.0.1) when it is not already
running under a launcher; under a launcher WORLD_SIZE must equal --gpus. A box with fewer than N GPUs fails loudly.

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the field meanings):
  value            whole-job Mpoints/s over all ranks (median of the --repeats timed blocks of --steps steps each)
  repeats          median/min/max ms_per_step over --repeats blocks of --steps steps
  roofline         dominant kernel: algorithmic bytes per launch / its average duration measured with HIP events on
                   the codec's stream inside the timed region
  cpu_baseline     the real reference (oracle/_ref, kind "reference") or the C port (kind "port") timed on the
                   host cores on a bounded sample of the same workload (rank 0, N=1 only)
  e2e              PCIe-inclusive and single-cloud figures (never `value`), each next to the reference on the host
  configs          encode + decode (ms, Mpoints/s, bit_exact) of BASELINE configs C1, C3, C4, C5 through the same calls (N=1 only)
  bit_exact        the streams of the timed batch equal the checker's (compiled reference, else the C port), byte for byte
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def make_workload(name: str, points: int, n_distinct: int):
    from cloudini_amd import synth
    clouds = []
    info = None
    for k in range(n_distinct):
        if name == "c2":
            info, data = synth.lidar_xyzi(points, seed=42 + k)
        elif name == "c1" or name == "c5":
            info, data = synth.lidar_xyz(points, seed=42 + k)
        elif name == "c3":
            info, data = synth.depthcam_xyzrgba(1280, 800, seed=42 + k)
        elif name == "c4":
            info, data = synth.velodyne_xyzir(130048, seed=42 + k)
        else:
            raise SystemExit(f"unknown workload {name}")
        clouds.append(data)
    return info, clouds


WORKLOAD_DESC = {
    "c2": "BASELINE configs[1]: {clouds} x {points}-pt XYZI (f32 XYZ + u16 intensity, point_step 16) @1mm, "
          "stage-1 encode (pre-LZ4/ZSTD), V5 wire",
    "c1": "BASELINE configs[0] shape: {clouds} x {points}-pt XYZ f32 @1mm, stage-1 encode",
    "c5": "BASELINE configs[4] encode half: {clouds} x {points}-pt XYZ f32 @1mm, stage-1 encode",
    "c3": "BASELINE configs[2]: {clouds} x 1024000-pt organised XYZRGBA (32-byte stride) @0.1mm, stage-1 encode",
    "c4": "BASELINE configs[3]: {clouds} x 130048-pt XYZI(f32)+ring(u16), 18-byte points, stage-1 encode",
}


def host_cpus():
    """(usable cores, visible hardware threads, cgroup quota or None): containers often grant fewer CPUs than they show."""
    ncores = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2 cpu.max = "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except (OSError, ValueError):
        pass
    return (min(ncores, quota) if quota is not None else ncores), ncores, quota


def cpu_baseline(info, cloud, budget_s: float):
    """Reference (or port) stage-1 encode of ONE cloud of the workload on the host: single thread, encoder
    constructed outside the timed region, pre-sized output (mcap_codec_benchmark.cpp:447-457 bracket)."""
    from oracle import binding
    pts = len(cloud) // info.point_step
    try:
        ref = binding.RefLib()
        _size, t = ref.bench_encode(info, cloud, reps=3, threads=1)
        per = float(np.median(t))
        reps = int(max(5, min(2000, budget_s / max(per, 1e-6))))
        _size, t = ref.bench_encode(info, cloud, reps=reps, threads=1)
        best, med = float(t.min()), float(np.median(t))
        out = {"value": pts / med / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "reference",
               "sample": f"{reps} x encode() of one {pts}-pt cloud of the workload, 1 thread, median "
                         f"(best {pts / best / 1e6:.1f} Mpoints/s)"}
        ncores, visible, quota = host_cpus()
        if ncores > 1:
            reps_mt = max(3, reps // 4)
            _size, tm = ref.bench_encode(info, cloud, reps=reps_mt, threads=ncores)
            agg = float(np.sum(pts / np.median(tm, axis=1))) / 1e6
            out["all_cores"] = {"value": agg, "cores": ncores, "visible_threads": visible, "cgroup_cpu_quota": quota,
                                "sample": f"{ncores} independent encoders x {reps_mt} reps, sum of per-thread medians"}
        return out
    except (FileNotFoundError, OSError):
        orc = binding.Oracle()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget_s or reps < 3:
            orc.encode_stage1(info, cloud)
            reps += 1
        per = (time.perf_counter() - t0) / reps
        return {"value": pts / per / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "port",
                "sample": f"{reps} x orc_encode_stage1 of one {pts}-pt cloud of the workload, 1 thread, mean"}


def _stats_ms(times):
    t = np.asarray(times, dtype=np.float64) * 1e3
    return {"median_ms": float(np.median(t)), "min_ms": float(t.min()), "max_ms": float(t.max()), "n": int(t.size)}


def e2e_figures(info, cloud, dev, budget_s: float):
    """End-to-end figures for ONE cloud of the workload per call, every call synchronous (never `value`):
      device_resident   cldn_hip_encode_stage1, input and output in HBM
      device_resident_decode  cldn_hip_decode_stage1 of that stream, input and output in HBM
      pinned_host       the same call with HOST tags on pinned memory (H2D + kernels + D2H)
      pageable_host     ... on pageable memory
      host_mirror_decode_*  Cloudini::PointcloudDecoder::decode of the stream the encoder leg produced, next to the
                        reference's decoder
      host_mirror_*     Cloudini::PointcloudEncoder (libcloudini_amd.so) constructed fresh for every call, full stream
                        with header into a pre-sized buffer (mcap_codec_benchmark.cpp:447-457 bracket, construction
                        included), next to the compiled reference under the same bracket on the host cores."""
    import ctypes as C
    import torch
    from cloudini_amd import api, native
    from cloudini_amd.schema import CompressionOption

    step = info.point_step
    pts = len(cloud) // step
    out = {"points_per_call": pts, "note": "one cloud per call, synchronous; median of the calls that fit the budget"}
    per_leg = max(0.3, budget_s / 10.0)

    def timed(fn, warm=2):
        for _ in range(warm):
            fn()
        ts = []
        t_end = time.perf_counter() + per_leg
        while len(ts) < 5 or (time.perf_counter() < t_end and len(ts) < 200):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        s = _stats_ms(ts)
        s["Mpoints_per_s"] = pts / (s["median_ms"] * 1e-3) / 1e6
        return s

    plan = native.Plan(info)
    codec = native.Codec(plan, device=dev.index)
    L = native.lib()
    cap = plan.stage1_bound(pts)
    cp = np.array([pts], dtype=np.uint64)

    d_in = torch.from_numpy(cloud).to(dev)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(2, dtype=torch.int64, device=dev)

    def dev_call():
        codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
        codec.synchronize()
    out["device_resident"] = timed(dev_call)
    # ... and back: the stream that call left in HBM decoded into HBM (chunk sizes from the encoder), one cloud per call
    n_chunks_1 = (pts + 32767) // 32768
    d_sizes = torch.zeros(max(1, n_chunks_1), dtype=torch.int32, device=dev)
    codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)
    codec.synchronize()
    offs_dev = d_off.cpu().numpy().astype(np.uint64)
    d_dec = torch.zeros(d_in.numel(), dtype=torch.uint8, device=dev)

    def dev_decode_call():
        codec.decode_device(d_out.data_ptr(), offs_dev, cp, d_dec.data_ptr(), d_dec.numel(), d_sizes.data_ptr())
        codec.synchronize()
    out["device_resident_decode"] = timed(dev_decode_call)

    # first call of a FRESH codec (the ROS plugins construct an encoder / a decoder per message,
    # cloudini_publisher_plugin.cpp:53-55): workspaces cold, no mode / Palette hint from an earlier call. Codec creation
    # (cldn_hip_codec_create: stream, status words) is outside the bracket, everything the call allocates is inside.
    def first_call(direction):
        ts = []
        for _ in range(3):
            c2 = native.Codec(plan, device=dev.index)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            if direction == "encode":
                c2.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr())
            else:
                c2.decode_device(d_out.data_ptr(), offs_dev, cp, d_dec.data_ptr(), d_dec.numel(), d_sizes.data_ptr())
            c2.synchronize()
            ts.append(time.perf_counter() - t0)
            c2.close()
        return _stats_ms(ts)
    out["first_call"] = {"encode": first_call("encode"), "decode": first_call("decode"),
                         "note": "fresh codec per call: device workspaces are allocated inside the bracket (hipMalloc), no hints"}
    out["device_resident_decode"]["bit_exact_round_trip"] = bool(torch.equal(d_dec, d_in)) if not any(f.resolution for f in info.fields) else None
    out["device_resident_decode"]["note"] = "batches of at most 64 chunks take the SPLIT launches of the point decoder (round 5)"

    offs = np.zeros(2, dtype=np.uint64)
    src_pin = torch.from_numpy(cloud).pin_memory()
    dst_pin = torch.empty(cap, dtype=torch.uint8).pin_memory()

    def pinned_call():
        rc = L.cldn_hip_encode_stage1(codec._h, C.c_void_p(src_pin.data_ptr()), 0, cp.ctypes.data_as(C.POINTER(C.c_uint64)), 1,
                                      C.c_void_p(dst_pin.data_ptr()), cap, 0, offs.ctypes.data_as(C.c_void_p), None, None)
        assert rc == 0
    out["pinned_host"] = timed(pinned_call)

    dst_pg = np.empty(cap, dtype=np.uint8)

    def pageable_call():
        rc = L.cldn_hip_encode_stage1(codec._h, cloud.ctypes.data_as(C.c_void_p), 0, cp.ctypes.data_as(C.POINTER(C.c_uint64)), 1,
                                      dst_pg.ctypes.data_as(C.c_void_p), cap, 0, offs.ctypes.data_as(C.c_void_p), None, None)
        assert rc == 0
    out["pageable_host"] = timed(pageable_call)
    codec.close()

    try:
        from oracle.binding import RefLib
        ref = RefLib()
    except (OSError, FileNotFoundError):
        ref = None
    pool_threads = api.stage2_threads() if hasattr(api, "stage2_threads") else None
    for comp in (CompressionOption.NONE, CompressionOption.LZ4):
        inf = info.copy(compression_opt=comp, use_threads=True)
        ci, _keep = api._c_info(inf)
        hl = api.lib()
        hcap = api.MaxCompressedSize(inf, pts, True)
        hout = np.empty(hcap, dtype=np.uint8)
        size_box = [0]

        def mirror_call():
            n = hl.cldn_amd_encode(C.byref(ci), api._ptr(cloud), cloud.size, api._ptr(hout), hcap, 1)
            assert n > 0
            size_box[0] = n
        leg = timed(mirror_call)
        leg["bytes"] = int(size_box[0])
        leg["stage2_threads"] = pool_threads if comp != CompressionOption.NONE else 0
        leg["bracket"] = "fresh PointcloudEncoder per call (construction inside the timed region), pre-sized output, host buffers"
        if ref is not None:
            reps = int(max(3, min(30, per_leg / 0.03)))
            _size, t = ref.bench_encode(inf, cloud, reps=reps, threads=1)
            leg["reference"] = {"median_ms": float(np.median(t)) * 1e3, "min_ms": float(t.min()) * 1e3, "n": reps,
                                "threads": "1 encode thread + the reference's own stage-2 worker (use_threads=true)",
                                "bracket": "encoder constructed outside the timed region, pre-sized output"}
        out["host_mirror_" + comp.name] = leg
        # the way back: PointcloudDecoder::decode of that very stream (header parsed once outside, like the subscriber
        # plugin does), host buffers, next to the reference's decoder
        stream = hout[: size_box[0]].copy()
        hdr_len = int(np.nonzero(stream == 0)[0][0]) + 1
        body = stream[hdr_len:]
        dec_out = np.empty(cloud.size, dtype=np.uint8)

        def mirror_decode():
            n = hl.cldn_amd_decode_noheader(C.byref(ci), api._ptr(body), body.size, api._ptr(dec_out), dec_out.size)
            assert n == cloud.size
        dleg = timed(mirror_decode)
        dleg["stage2_threads"] = pool_threads if comp != CompressionOption.NONE else 0
        dleg["bracket"] = "fresh PointcloudDecoder per call, header parsed outside, pre-sized output, host buffers"
        if ref is not None:
            reps = int(max(3, min(30, per_leg / 0.03)))
            _dec, t = ref.bench_decode(stream, cloud.size, reps=reps)
            dleg["reference"] = {"median_ms": float(np.median(t)) * 1e3, "min_ms": float(t.min()) * 1e3, "n": reps,
                                 "threads": "1 decode thread (the reference's decoder is single-threaded)"}
        out["host_mirror_decode_" + comp.name] = dleg
        if comp == CompressionOption.NONE:
            # the same decode into a buffer the caller has just zero-initialised (decode(info, data, std::vector&) on an
            # empty vector, convertCompressedCloudToPointCloud2): nothing of it travels to the GPU first
            def mirror_decode_zeroed():
                n = hl.cldn_amd_decode_noheader_zeroed(C.byref(ci), api._ptr(body), body.size, api._ptr(dec_out), dec_out.size)
                assert n == cloud.size
            zleg = timed(mirror_decode_zeroed)
            zleg["bracket"] = "as host_mirror_decode_NONE, output declared all-zero on entry (PointcloudDecoder::decodeInto)"
            out["host_mirror_decode_NONE_zeroed_output"] = zleg
        if comp == CompressionOption.LZ4 and hasattr(api, "set_stage2_threads"):
            # the same call with the stage-2 knob at 16 threads (the box grants 16 CPUs): from 8 threads on encode() cuts
            # the cloud into two chunk groups and compresses the first while the GPU encodes the second
            api.set_stage2_threads(16)
            try:
                leg16 = timed(mirror_call)
            finally:
                api.set_stage2_threads(pool_threads or 4)
            leg16["bytes"] = int(size_box[0])
            leg16["stage2_threads"] = 16
            leg16["bracket"] = leg["bracket"] + "; CLOUDINI_AMD_STAGE2_THREADS=16, chunk-group pipeline (2 groups)"
            out["host_mirror_LZ4_16_threads"] = leg16
        if comp == CompressionOption.LZ4 and hasattr(api, "set_device_lz4"):
            # stage 2 on the GPU as well (valid LZ4 blocks, not lz4's own bytes); the stream is
            # checked through the reference's decoder once, outside the timed calls
            api.set_device_lz4(True)
            try:
                legd = timed(mirror_call)
                dev_stream = hout[: size_box[0]].copy()
            finally:
                api.set_device_lz4(False)
            legd["bytes"] = int(size_box[0])
            legd["stage2_threads"] = 0
            legd["bracket"] = leg["bracket"] + "; CLOUDINI_AMD_DEVICE_LZ4=1: LZ4 blocks written by the GPU (lz4_kernels.hip)"
            if ref is not None:
                back, _y = ref.decode(dev_stream, cloud.size)
                want, _y = ref.decode(stream, cloud.size)
                legd["decodes_through_reference_to_the_same_points"] = bool(np.array_equal(back, want))
            out["host_mirror_LZ4_device"] = legd
            api.set_device_lz4(2)  # the FAST parameters (CLDN_HIP_STAGE2_LZ4_FAST: 4 KiB windows, blocks ~3 % larger)
            try:
                legf = timed(mirror_call)
                fast_stream = hout[: size_box[0]].copy()
            finally:
                api.set_device_lz4(False)
            legf["bytes"] = int(size_box[0])
            legf["stage2_threads"] = 0
            legf["bracket"] = leg["bracket"] + "; CLOUDINI_AMD_DEVICE_LZ4=2: LZ4 blocks written by the GPU, 4 KiB windows"
            if ref is not None:
                back, _y = ref.decode(fast_stream, cloud.size)
                legf["decodes_through_reference_to_the_same_points"] = bool(np.array_equal(back, want))
            out["host_mirror_LZ4_device_fast"] = legf
    return out


def transcode_figures(n_msgs: int):
    """BASELINE configs[3] through the batch transcoder (the loop it replaces:
    tools/src/mcap_converter.cpp:170-222): n_msgs CDR sensor_msgs/PointCloud2 messages of 130048 XYZI+ring points in a
    directory -> CompressedPointCloud2 messages, ZSTD second stage on the host pool. In process, second of two runs."""
    import tempfile
    from cloudini_amd import api, synth
    from cloudini_amd.schema import CompressionOption
    distinct = [synth.velodyne_xyzir(130048, seed=42 + k) for k in range(4)]
    msgs = [synth.cdr_pointcloud2(distinct[k % 4][0], distinct[k % 4][1], stamp=(1700000000, k)) for k in range(n_msgs)]
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        src, dst = os.path.join(tmp, "in"), os.path.join(tmp, "out")
        os.makedirs(src)
        for k, m in enumerate(msgs):
            m.tofile(os.path.join(src, f"msg_{k:05d}.bin"))
        st = None
        for _rep in range(2):
            t0 = time.perf_counter()
            st = api.transcode_directory(src, dst, resolution=0.001, compression_opt=int(CompressionOption.ZSTD), batch_messages=32)
            wall = time.perf_counter() - t0
        out = {"workload": f"BASELINE configs[3]: {n_msgs} x 130048-pt XYZI(f32)+ring(u16) CDR PointCloud2 messages, 1 mm, ZSTD",
               "Mpoints_per_s": st["points"] / wall / 1e6, "seconds": wall, "messages": int(st["messages"]),
               "gpu_batches": int(st["gpu_batches"]), "gpu_stage_busy_fraction": st["seconds_gpu"] / max(wall, 1e-9),
               "stage2_threads": api.stage2_threads(), "output_bytes_per_point": st["output_bytes"] / max(1.0, st["points"])}
        # the same run with the stage-2 pool sized to the CPUs the box grants (minus the reader, GPU and writer threads):
        # what a batch tool would set; the default above is the library's conservative min(4, cores)
        usable = host_cpus()[0]
        if usable - 3 > api.stage2_threads():
            before = api.stage2_threads()
            api.set_stage2_threads(usable - 3)
            try:
                st2 = None
                for _rep in range(2):
                    t0 = time.perf_counter()
                    st2 = api.transcode_directory(src, dst, resolution=0.001, compression_opt=int(CompressionOption.ZSTD), batch_messages=32)
                    wall2 = time.perf_counter() - t0
                out["all_granted_cpus"] = {"stage2_threads": api.stage2_threads(), "Mpoints_per_s": st2["points"] / wall2 / 1e6,
                                           "seconds": wall2, "gpu_stage_busy_fraction": st2["seconds_gpu"] / max(wall2, 1e-9)}
            finally:
                api.set_stage2_threads(before)
        try:
            from oracle.binding import RefLib
            ref = RefLib()
            t0 = time.perf_counter()
            ok = True
            sample = min(4, n_msgs)
            for k in range(sample):
                want = ref.ros_compress(msgs[k], 0.001, int(CompressionOption.ZSTD))
                ok = ok and bool(np.array_equal(np.fromfile(os.path.join(dst, f"msg_{k:05d}.bin"), dtype=np.uint8), want))
            per = (time.perf_counter() - t0) / sample
            out["reference_converter_Mpoints_per_s_1_thread"] = 130048 / per / 1e6
            out["sample_messages_equal_reference"] = ok
        except (OSError, FileNotFoundError):
            pass
        return out


def leg_roofline(alg_bytes: float, kernel: str, kernel_ms: float, whole_ms: float, what: str):
    """Round 5: every leg of the line carries its own fractions. algorithmic_bytes = the leg's useful HBM bytes per step
    (input read + output written, nothing counted twice); `frac` = those bytes over the HIP-event time of
    the leg's dominant kernel (a lower bound of what that kernel moves: it is charged with the whole leg's bytes) against the
    8 TB/s peak; `whole_leg_frac` = the same bytes over the leg's wall time per step."""
    k = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms and kernel_ms > 0 else None
    w = alg_bytes / (whole_ms * 1e-3) / 1e9 if whole_ms and whole_ms > 0 else None
    return {"bound": "hbm", "algorithmic_bytes": alg_bytes, "bytes_counted": what, "kernel": kernel, "kernel_ms": kernel_ms,
            "kernel_GBps": k, "frac": None if k is None else k / HBM_PEAK_GBPS, "whole_leg_ms": whole_ms, "whole_leg_GBps": w,
            "whole_leg_frac": None if w is None else w / HBM_PEAK_GBPS, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}


def store_pattern_calibration(n_points: int):
    """Round 6: what the decode leg's STORE PATTERN alone costs on this box (tools/hbm_calib.hip, mode points14: plain grid-stride
    kernels, no arithmetic). A decoder that leaves the bytes no field covers alone (KEEP, the reference's behaviour) writes 12 + 2
    of every 16-byte XYZI point: every 32-byte sector of the output is written partially and the memory side has to merge it.
    `keep_ms` is that pattern with the encoded stream's bytes read alongside, `full_ms` the same points as whole 16-byte stores
    (CLDN_HIP_FILL_ZERO). decode.roofline.kernel_ms is to be read against keep_ms, decode.fill_zero_ms_per_step against full_ms."""
    exe = os.path.join(ROOT, "cloudini_amd", "lib", "hbm_calib")
    if not os.path.exists(exe):
        return None
    try:
        r = subprocess.run([exe, "1", "points14", str(int(n_points))], capture_output=True, text=True, timeout=120)
        best = {}
        for line in r.stdout.splitlines():
            for key, tag in (("12+2, +read", "keep_ms"), ("16, +read", "full_ms"), ("12+2 of 16", "keep_store_only_ms"), ("16 of 16", "full_store_only_ms")):
                if line.startswith(key):
                    ms = float(line.split("best")[1].split("ms")[0])
                    best[tag] = min(best.get(tag, 1e9), ms)
        if not best:
            return None
        best["what"] = ("tools/hbm_calib.hip points14: 12 + 2 of every 16 bytes written (keep) / whole 16-byte stores (full), "
                        f"{int(n_points)} points, with / without 6.4 bytes per point read alongside; best of 9 runs and 3 grid sizes")
        return best
    except Exception as exc:  # a calibration that fails must never cost the line
        return {"error": repr(exc)}


def decode_kernel_name(info) -> str:
    """The kernel stage1_launch_decode gives the regular streams of this schema (stage1_decode.hip)."""
    lead = 0
    for f in info.fields:
        if int(f.type) == 7 and f.resolution is not None and int(info.encoding_opt) == 1:
            lead += 1
        else:
            break
    rest = info.fields[lead:] if lead in (3, 4) else info.fields
    v5 = int(info.version) >= 5 and int(info.encoding_opt) == 1
    if lead in (3, 4) and all(int(f.type) in (3, 4, 5, 6, 9, 10) for f in rest) and (v5 or not rest):
        return "k_decode_points_w"
    if int(info.encoding_opt) != 1 and all(int(f.type) in (1, 2, 7, 8) for f in info.fields) and not (
            int(info.encoding_opt) == 2 and int(info.version) >= 4 and any(int(f.type) == 8 for f in info.fields)):
        return "k_decode_fixed"
    return "k_decode_stream_w"


def config_figures(dev, steps: int):
    """extra (never `value`): the other BASELINE configs through the same two device-resident calls -- encode to the framed
    streams, decode with the encoder's chunk sizes -- so that every config has a number in the bench line. Each config's
    first cloud is compared with the reference (the C oracle where oracle/_ref is absent) and the decode with the
    reference's decode of that stream, outside the timed loops."""
    import torch
    from cloudini_amd import native
    checker, checker_kind = None, None
    try:
        from oracle.binding import RefLib
        checker, checker_kind = RefLib(), "reference"
    except (OSError, FileNotFoundError, ImportError):
        try:
            from oracle.binding import Oracle
            checker, checker_kind = Oracle(), "port"
        except (OSError, FileNotFoundError, ImportError):
            pass
    out = {}
    stream = torch.cuda.current_stream(dev)
    for name, clouds, points in (("c1", 256, 65536), ("c3", 16, 1024000), ("c4", 256, 130048), ("c5", 1, 10_000_000)):
        try:
            info, distinct = make_workload(name, points, min(4, clouds))
            step = info.point_step
            n = len(distinct[0]) // step
            plan = native.Plan(info)
            codec = native.Codec(plan, device=dev.index or 0, stream=stream.cuda_stream)
            host = np.concatenate([distinct[k % len(distinct)] for k in range(clouds)])
            cloud_points = np.full(clouds, n, dtype=np.uint64)
            d_points = torch.from_numpy(host).to(dev)
            cap = plan.stage1_bound(n) * clouds
            d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_off = torch.zeros(clouds + 1, dtype=torch.int64, device=dev)
            n_chunks = clouds * ((n + 32767) // 32768)
            d_sizes = torch.zeros(max(1, n_chunks), dtype=torch.int32, device=dev)

            def enc():
                codec.encode_device(d_points.data_ptr(), cloud_points, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)

            def timed(fn):
                for _ in range(3):
                    fn()
                blocks = []
                for _ in range(3):
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(steps):
                        fn()
                    torch.cuda.synchronize(dev)
                    blocks.append((time.perf_counter() - t0) / steps)
                return float(np.median(blocks)) * 1e3

            enc_ms = timed(enc)
            codec.status()
            offs = d_off.cpu().numpy().astype(np.uint64)
            d_dec = torch.zeros(host.size, dtype=torch.uint8, device=dev)
            dec_ms = timed(lambda: codec.decode_device(d_out.data_ptr(), offs, cloud_points, d_dec.data_ptr(), host.size, d_sizes.data_ptr()))
            codec.status()
            dec_stats = codec.decode_stats()
            codec.set_decode_fill(True)
            dec_zero_ms = timed(lambda: codec.decode_device(d_out.data_ptr(), offs, cloud_points, d_dec.data_ptr(), host.size, d_sizes.data_ptr()))
            codec.set_decode_fill(False)
            codec.enable_timing(4)  # HIP events around the kernels: behind the timed loops (the records cost a few per cent)
            for _ in range(4):
                enc()
            enc_k = [codec.kernel_ms(s_) for s_ in range(4)]
            dec_k = []
            for _ in range(4):  # (the last one leaves the checked bytes: default mode)
                codec.decode_device(d_out.data_ptr(), offs, cloud_points, d_dec.data_ptr(), host.size, d_sizes.data_ptr())
                dec_k.append(codec.decode_ms())
            codec.status()
            entry = {"workload": WORKLOAD_DESC[name].format(clouds=clouds, points=n),
                     "decode_fill_zero_ms": dec_zero_ms,
                     "decode_chunks_parallel_regular/parallel_sections/serial/serial_sections": list(dec_stats),
                     "encode_ms": enc_ms, "encode_Mpoints_per_s": clouds * n / (enc_ms * 1e-3) / 1e6,
                     "decode_ms": dec_ms, "decode_Mpoints_per_s": clouds * n / (dec_ms * 1e-3) / 1e6,
                     "stage1_bytes_per_point": int(offs[-1]) / (clouds * n)}
            alg = float(clouds * n * step + int(offs[-1]))
            pipeline = codec.pipeline(0, d_points.data_ptr())
            entry["encode_roofline"] = leg_roofline(alg, "k_encode_fused" if pipeline >= 2 else "k_encode_regular / k_encode_floatn",
                                                    float(np.mean([k["regular"] for k in enc_k])), enc_ms,
                                                    "point bytes read + framed stream bytes written")
            entry["encode_device_ms"] = {"regular": float(np.mean([k["regular"] for k in enc_k])),
                                         "sections": float(np.mean([k["sections"] for k in enc_k])),
                                         "k_finish": float(np.mean([k["compact"] for k in enc_k]))}
            entry["decode_roofline"] = leg_roofline(alg, decode_kernel_name(info), float(np.median([k["regular_kernel"] for k in dec_k])),
                                                    dec_ms, "stream bytes read + point bytes written")
            if checker is not None:
                got = d_out[int(offs[0]):int(offs[1])].cpu().numpy()
                want = checker.encode_stage1(info, distinct[0])
                dec_got = d_dec[: n * step].cpu().numpy()
                dec_want = (checker.decode_noheader(info, want, 0) if checker_kind == "reference" else checker.decode_stage1(info, want, n, 0))[: n * step]
                entry["bit_exact"] = bool(np.array_equal(got, want)) and bool(np.array_equal(dec_got, dec_want))
                entry["checker"] = checker_kind
            out[name] = entry
            codec.close()
            del d_points, d_out, d_dec
        except Exception as exc:  # an extra must not take the line down
            out[name] = {"error": repr(exc)}
    return out


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args) -> int:
    """`python bench.py --gpus N` outside a launcher: start N ranks, one per GPU, and pass their exit code on."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} requested but this box has {have} GPU(s); refusing to report n_gpus={args.gpus} "
              "from fewer devices", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=50)  # (a step is 0.3 ms: 3 steps left the first timed block 5 % behind the others)
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps steps (the first one is `value`)")
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOAD_DESC))
    ap.add_argument("--shard", default="clouds", choices=("clouds", "chunks"))
    ap.add_argument("--clouds", type=int, default=None, help="clouds per step per GPU (default 32; 1 with --shard chunks)")
    ap.add_argument("--points", type=int, default=None, help="points per cloud (default 1M; 10M with --shard chunks)")
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic clouds generated (tiled to --clouds)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    ap.add_argument("--e2e-seconds", type=float, default=6.0, help="budget of the e2e legs (0 = skip; N=1 only)")
    ap.add_argument("--transcode-messages", type=int, default=256,
                    help="messages of the batch-transcoder leg (BASELINE configs[3], ZSTD; 0 = skip; N=1 only)")
    ap.add_argument("--config-legs", type=int, default=1,
                    help="1: the `configs` object (encode + decode of the other BASELINE configs, N=1 only); 0 = skip")
    ap.add_argument("--no-verify", dest="verify", action="store_false",
                    help="skip the bit-exactness check of the timed batch's streams (outside the timed region)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.clouds is None:
        args.clouds = 1 if args.shard == "chunks" else 32
    if args.points is None:
        args.points = 10_000_000 if args.shard == "chunks" else 1_000_000
    if args.shard == "chunks" and args.clouds != 1:
        raise SystemExit("--shard chunks splits ONE cloud over the ranks: use --clouds 1")

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} rank(s); "
                         "start one rank per GPU (--nproc-per-node must equal --gpus)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU (local rank {local_rank}, {torch.cuda.device_count()} device(s))")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run also a single rank takes the RCCL path
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == world

    from cloudini_amd import native, sharding

    info, distinct = make_workload(args.workload, args.points, max(1, min(args.distinct, args.clouds)))
    step = info.point_step
    pts_per_cloud = len(distinct[0]) // step
    plan = native.Plan(info)
    stream = torch.cuda.current_stream(dev)
    codec = native.Codec(plan, device=local_rank, stream=stream.cuda_stream)
    na = plan.adaptive_fields

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.shard == "clouds":
        n_clouds = args.clouds
        host = np.concatenate([distinct[(k + rank) % len(distinct)] for k in range(n_clouds)])
        cloud_points = np.full(n_clouds, pts_per_cloud, dtype=np.uint64)
        points_local = n_clouds * pts_per_cloud
        points_job = points_local * world
        scaling = "weak"
    else:
        # one cloud, contiguous ranges of whole chunks per rank (sharding.shard_chunks); every rank only uploads its range
        n_clouds = 1
        p0, cnt = sharding.shard_chunks(pts_per_cloud, world, rank)
        host = distinct[0][p0 * step:(p0 + cnt) * step]
        head = np.ascontiguousarray(distinct[0][: min(pts_per_cloud, sharding.PROBE_POINTS) * step])
        cloud_points = np.array([cnt], dtype=np.uint64)
        points_local = cnt
        points_job = pts_per_cloud
        scaling = "strong"
    d_points = torch.from_numpy(np.ascontiguousarray(host)).to(dev) if host.size else torch.empty(1, dtype=torch.uint8, device=dev)
    cap = max(1, plan.stage1_bound(int(cloud_points[0])) * n_clouds)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_offsets = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
    n_chunks = int(sum((int(n) + 32767) // 32768 for n in cloud_points))
    d_chunk_sizes = torch.zeros(max(1, n_chunks), dtype=torch.int32, device=dev)
    d_modes = torch.zeros(max(1, n_clouds * max(1, na)), dtype=torch.uint8, device=dev)

    exchanged = {}
    if args.shard == "clouds":
        def one_step():
            codec.encode_device(d_points.data_ptr(), cloud_points, d_out.data_ptr(), cap, d_offsets.data_ptr(),
                                d_chunk_sizes.data_ptr(), d_modes.data_ptr())
    else:
        # the full exchange protocol of DESIGN.md section 6 every step: probe on the head (rank 0), broadcast of the
        # mode bytes, encode of the range with the modes forced, all-gather of the byte counts
        head_codec = native.Codec(plan, device=local_rank, stream=stream.cuda_stream) if rank == 0 and na else None
        d_head = torch.from_numpy(head).to(dev) if head_codec is not None else None
        head_pts = head.size // step
        head_cap = plan.stage1_bound(head_pts)
        d_head_out = torch.empty(head_cap, dtype=torch.uint8, device=dev) if head_codec is not None else None
        d_head_modes = torch.zeros(max(1, na), dtype=torch.uint8, device=dev)

        def one_step():
            if na:
                if head_codec is not None:
                    head_codec.encode_device(d_head.data_ptr(), np.array([head_pts], dtype=np.uint64), d_head_out.data_ptr(),
                                             head_cap, 0, 0, d_head_modes.data_ptr())
                if dist is not None and world > 1:
                    dist.broadcast(d_head_modes, src=0)        # the path's one-to-all exchange (few bytes over RCCL)
                codec.force_modes(d_head_modes.cpu().numpy()[:na])
            if cnt:
                codec.encode_device(d_points.data_ptr(), cloud_points, d_out.data_ptr(), cap, d_offsets.data_ptr(),
                                    d_chunk_sizes.data_ptr(), d_modes.data_ptr())
            if dist is not None and world > 1:
                mine = d_offsets[1:2].clone() if cnt else torch.zeros(1, dtype=torch.int64, device=dev)
                gathered = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(gathered, mine)                # byte counts -> every rank's offset in the stream
                exchanged["sizes"] = gathered

    for _ in range(args.warmup):
        one_step()
    codec.status()  # raises on device-side errors
    codec.enable_timing(max(1, args.steps))

    block_times = []
    kms, kms_blocks = None, []
    for rep in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        block_times.append(elapsed)
        if points_local:  # (the slots hold the block's last --steps calls)
            kms_blocks.append([codec.kernel_ms(s) for s in range(args.steps)])
    codec.status()
    # value / ms_per_step: the MEDIAN block of --steps steps (round 4; the first block is reported next to it); round 5: the
    # per-kernel HIP-event times are those of the same block
    elapsed = float(np.median(block_times))
    if kms_blocks:
        kms = kms_blocks[int(np.argsort(block_times)[len(block_times) // 2])]

    if kms:
        regular_ms = float(np.mean([k["regular"] for k in kms]))
        sections_ms = float(np.mean([k["sections"] for k in kms]))
        compact_ms = float(np.mean([k["compact"] for k in kms]))
        device_ms = float(np.mean([k["total"] for k in kms]))
    else:
        regular_ms = sections_ms = compact_ms = device_ms = float("nan")

    offsets = d_offsets.cpu().numpy()
    total_out = int(offsets[-1])
    out_bpp = total_out / max(1, points_local)

    # job-wide output size through the size exchange of sharding.py (RCCL all-gather; also checks the exchange itself)
    job_bytes = total_out
    if dist is not None and world > 1:
        if args.shard == "clouds":
            local_sizes = [int(offsets[k + 1] - offsets[k]) for k in range(n_clouds)]
            t = torch.tensor(local_sizes, dtype=torch.int64, device=dev)
            gathered = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            job_bytes = int(sum(int(g.sum().item()) for g in gathered))
        else:
            job_bytes = int(sum(int(g.item()) for g in exchanged["sizes"]))

    # extra (not `value`): decode of the streams just produced, device resident, same synchronisation bracket
    decode = None
    if points_local:  # (every rank: the job-wide decode figure of a multi-GPU run is the slowest rank's)
        d_dec = torch.empty(max(1, host.size), dtype=torch.uint8, device=dev)
        so = offsets.astype(np.uint64)
        dec_steps = max(1, args.steps // 2)
        codec.enable_timing(0)  # (the event records between the kernels cost the decode legs 4-17 %: timed without them)
        for _ in range(min(2, max(1, args.warmup))):
            codec.decode_device(d_out.data_ptr(), so, cloud_points, d_dec.data_ptr(), host.size, d_chunk_sizes.data_ptr())
        dec_blocks = []
        for _rep in range(max(1, min(3, args.repeats))):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(dec_steps):
                codec.decode_device(d_out.data_ptr(), so, cloud_points, d_dec.data_ptr(), host.size, d_chunk_sizes.data_ptr())
            torch.cuda.synchronize(dev)
            dec_blocks.append((time.perf_counter() - t1) / dec_steps)
        codec.status()
        dec_stats = codec.decode_stats()
        # the same decode for a caller that does not need the buffer's content (CLDN_HIP_FILL_ZERO: a freshly created
        # output, like the reference's decode into an empty vector): whole-point stores for the padded point
        zero_blocks = []
        codec.set_decode_fill(True)
        for _rep in range(max(1, min(3, args.repeats))):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(dec_steps):
                codec.decode_device(d_out.data_ptr(), so, cloud_points, d_dec.data_ptr(), host.size, d_chunk_sizes.data_ptr())
            torch.cuda.synchronize(dev)
            zero_blocks.append((time.perf_counter() - t1) / dec_steps)
        codec.set_decode_fill(False)
        codec.status()
        # HIP-event time of the kernel that decodes the regular streams (cldn_hip_codec_decode_ms), a few more calls
        dec_kernel_ms = []
        codec.enable_timing(1)
        for _ in range(5):
            codec.decode_device(d_out.data_ptr(), so, cloud_points, d_dec.data_ptr(), host.size, d_chunk_sizes.data_ptr())
            dec_kernel_ms.append(codec.decode_ms())
        codec.status()
        del d_dec
        dec_ms = float(np.median(dec_blocks)) * 1e3
        decode = {"value": points_local / (dec_ms * 1e-3) / 1e6,
                  "unit": "Mpoints/s (rank 0, stage-1 decode of this rank's streams, device resident, chunk sizes from the encoder)",
                  "ms_per_step": dec_ms, "ms_per_step_min": float(min(dec_blocks)) * 1e3,
                  "ms_per_step_max": float(max(dec_blocks)) * 1e3,
                  "HBM_GBps": (total_out + points_local * step) / (dec_ms * 1e-3) / 1e9,
                  "fill_zero_ms_per_step": float(np.median(zero_blocks)) * 1e3,
                  "fill_zero_Mpoints_per_s": points_local / float(np.median(zero_blocks)) / 1e6,
                  "chunks_parallel_regular/parallel_sections/serial/serial_sections": list(dec_stats)}
        decode["roofline"] = leg_roofline(total_out + points_local * step, decode_kernel_name(info),
                                          float(np.median([k["regular_kernel"] for k in dec_kernel_ms])), dec_ms,
                                          "stream bytes read + point bytes written")
        if rank == 0 and args.workload == "c2":
            decode["store_pattern"] = store_pattern_calibration(points_local)
    # extra (not `value`): stage 1 WITHOUT the framing -- cldn_hip_encode_stage1_chunks leaves every chunk's payload as one
    # run of its slot (the reference's own stage-1 / stage-2 boundary is a buffer per chunk, src/cloudini.cpp:590-614);
    # a device-side stage 2 or any other consumer on the GPU starts from there. Same steps, same bracket; the table is
    # framed afterwards (outside the timed region) and compared with the streams of the timed batch.
    chunk_table = None
    if rank == 0 and points_local and args.shard == "clouds":
        try:
            ct_codec = native.Codec(plan, device=local_rank, stream=stream.cuda_stream)
            for _ in range(max(1, args.warmup)):
                ct_codec.encode_chunks_device(d_points.data_ptr(), cloud_points)
            ct_codec.status()
            ct_codec.enable_timing(max(1, args.steps))
            ct_blocks = []
            for _rep in range(max(1, min(3, args.repeats))):
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    table = ct_codec.encode_chunks_device(d_points.data_ptr(), cloud_points)
                torch.cuda.synchronize(dev)
                ct_blocks.append((time.perf_counter() - t1) / args.steps)
            ct_codec.status()
            ct_k = [ct_codec.kernel_ms(s_) for s_ in range(args.steps)]
            d_out2 = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_off2 = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
            ct_codec.frame_chunks_device(d_out2.data_ptr(), cap, d_off2.data_ptr())
            torch.cuda.synchronize(dev)
            ct_codec.status()
            same = bool(torch.equal(d_off2, d_offsets)) and bool(torch.equal(d_out2[:total_out], d_out[:total_out]))
            import ctypes as _C
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            _C.CDLL("libamdhip64.so").hipMemcpy(_C.c_void_p(flag.data_ptr()), _C.c_void_p(table.not_contiguous), _C.c_size_t(4), 3)
            ct_ms = float(np.median(ct_blocks)) * 1e3
            ct_dev = float(np.mean([k["total"] for k in ct_k]))
            chunk_table = {"value": points_local / (ct_ms * 1e-3) / 1e6, "unit": "Mpoints/s (rank 0, stage 1 to the chunk table: no framing)",
                           "ms_per_step": ct_ms, "ms_per_step_min": float(min(ct_blocks)) * 1e3,
                           "device_ms_per_step": {"regular": float(np.mean([k["regular"] for k in ct_k])),
                                                  "sections": float(np.mean([k["sections"] for k in ct_k])), "all_kernels": ct_dev},
                           "whole_stage1_GBps": points_local * (step + out_bpp) / (ct_dev * 1e-3) / 1e9,
                           "payloads_contiguous": int(flag.item()) == 0,
                           "framed_afterwards_equals_the_timed_batch": same,
                           "roofline": leg_roofline(points_local * (step + out_bpp), "k_encode_fused (intra-chunk placement)",
                                                    float(np.mean([k["regular"] for k in ct_k])), ct_ms,
                                                    "point bytes read + payload bytes written")}
            del d_out2
            ct_codec.close()
        except Exception as exc:  # never costs the headline line
            chunk_table = {"error": repr(exc)}

    # "plus bit-exactness flag vs oracle" -- the streams the TIMED batch left in HBM, compared
    # outside the timed region with the compiled reference (oracle/_ref) or, where that is absent, the C port: every
    # distinct cloud against the checker, every tiled copy against its first occurrence
    bit_exact = None
    if points_local and args.shard == "clouds" and args.verify:  # (every rank checks its own streams)
        from oracle import binding
        try:
            checker, checker_kind = binding.RefLib(), "reference"
        except (OSError, FileNotFoundError):
            checker, checker_kind = binding.Oracle(), "port"
        ok, compared = True, 0
        first_of = {}
        for k in range(n_clouds):
            j = (k + rank) % len(distinct)
            got = d_out[int(offsets[k]):int(offsets[k + 1])]
            if j not in first_of:
                first_of[j] = got
                want = checker.encode_stage1(info, distinct[j])
                ok = ok and got.numel() == len(want) and bool(np.array_equal(got.cpu().numpy(), want))
                compared += 1
            else:
                ok = ok and got.numel() == first_of[j].numel() and bool(torch.equal(got, first_of[j]))
        bit_exact = {"ok": bool(ok), "checker": checker_kind, "clouds_vs_checker": compared, "clouds_vs_first_copy": n_clouds - compared}
    # job-wide figures of this rank count (round 4: the same object for every --workload and --gpus, so that the first run on
    # a multi-GPU node is a measurement): the slowest rank's decode, every rank's bit-exactness
    job = None
    if dist is not None and world > 1:
        t = torch.tensor([decode["ms_per_step"] if decode else 0.0, 1.0 if (bit_exact is None or bit_exact["ok"]) else 0.0],
                         dtype=torch.float64, device=dev)
        tmax, tmin = t.clone(), t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        job = {"decode_ms_per_step_max_over_ranks": float(tmax[0].item()), "bit_exact_all_ranks": bool(tmin[1].item() > 0.5)}
    elif decode is not None:
        job = {"decode_ms_per_step_max_over_ranks": decode["ms_per_step"], "bit_exact_all_ranks": bit_exact["ok"] if bit_exact else None}
    if dist is not None:
        dist.barrier()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mpts = points_job * args.steps / elapsed / 1e6
        n_ch_cloud = (pts_per_cloud + 32767) // 32768
        # regular-stream bytes per point: payload minus sections. Measured from the stream size when there are no
        # adaptive fields; otherwise from the float-only schema of the same buffer (V5 == V4 bytes for float-only
        # clouds, test_field_encoders.cpp:695-769): the HIP codec's own stream of that schema IS the regular stream.
        reg_bpp = out_bpp - 4.0 * n_chunks / max(1, points_local)
        if na:
            float_only = info.copy(fields=[f for f in info.fields if int(f.type) == 7])
            fo_codec = native.Codec(native.Plan(float_only), device=local_rank)
            fo_stream = fo_codec.encode_host([distinct[0]])[0][0]
            fo_codec.close()
            reg_bpp = (len(fo_stream) - 4 * n_ch_cloud) / pts_per_cloud
        alg_bytes = points_local * (step + reg_bpp)
        achieved = alg_bytes / (regular_ms * 1e-3) / 1e9
        # the regular-stream kernel of this plan: the fused FloatN kernel when the stream is exactly one FloatN encoder
        lead = 0
        for f in info.fields:
            if int(f.type) == 7 and f.resolution is not None:
                lead += 1
            else:
                break
        rest_adaptive = all(int(f.type) in (3, 4, 5, 6, 9, 10) for f in info.fields[lead:])
        floatn = lead in (3, 4) and rest_adaptive and int(info.version) >= 5
        # which kernel that is follows the codec's pipeline for this buffer (cldn_hip_codec_pipeline): the piece kernel
        # k_encode_fused (slot mode) by default, the tile kernel k_encode_floatn with pipeline 1
        pipeline = codec.pipeline(0, d_points.data_ptr())
        dominant = ("k_encode_fused" if pipeline >= 2 else "k_encode_floatn") if floatn else "k_encode_regular"
        # HBM bytes per launch of the same kernel from the PMC passes of tools/profile_round.sh (rocprofv3 cannot
        # collect counters from inside this process); only quoted when the committed pass ran this very workload
        traffic, traffic_src = None, None
        pdir = os.path.join(ROOT, "profiles")
        for name in sorted(os.listdir(pdir), reverse=True) if os.path.isdir(pdir) else []:
            if name.endswith("_traffic.json"):
                try:
                    t = json.load(open(os.path.join(pdir, name)))
                except (OSError, ValueError):
                    continue
                if (t.get("workload") == args.workload and t.get("clouds_per_gpu") == n_clouds
                        and t.get("points_per_cloud") == pts_per_cloud and args.shard == "clouds"):
                    traffic, traffic_src = float(t["hbm_bytes_per_launch"]), "profiles/" + name
                    break
        blocks_ms = np.asarray(block_times) / args.steps * 1e3
        if args.shard == "clouds":
            par = f"whole clouds sharded over {world} GPU(s), no data-path collective (RCCL: barrier, size all-gather)"
        else:
            par = (f"one cloud cut into chunk ranges over {world} GPU(s); per step: mode probe on rank 0, RCCL broadcast of "
                   f"{na} mode byte(s), encode with forced modes, RCCL all-gather of the byte counts")
        result = {
            "metric": "encode Mpoints/s (stage-1, pre-ZSTD)",
            "value": mpts,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32->i32 (u8 varint stream)",
            "data": f"synthetic lidar64 generator (cloudini_amd/synth.py), {len(distinct)} distinct clouds tiled, "
                    "inputs resident in HBM",
            "config": {"workload": WORKLOAD_DESC[args.workload].format(clouds=args.clouds, points=pts_per_cloud),
                       "clouds_per_gpu": n_clouds if args.shard == "clouds" else None, "points_per_cloud": pts_per_cloud,
                       "point_step": step, "shard": args.shard, "parallelism": par},
            "repeats": {"blocks": int(blocks_ms.size), "steps_per_block": args.steps,
                        "ms_per_step_median": float(np.median(blocks_ms)), "ms_per_step_min": float(blocks_ms.min()),
                        "ms_per_step_max": float(blocks_ms.max()),
                        "ms_per_step_first_block": float(blocks_ms[0]),
                        "value_median": points_job / (float(np.median(blocks_ms)) * 1e-3) / 1e6},
            "input_MBps": points_job * step * args.steps / elapsed / 1e6,
            "job_stage1_bytes": job_bytes,
            "bit_exact": bit_exact["ok"] if bit_exact else None,
            "bit_exact_detail": bit_exact,
            "decode": decode,
            "job": None if job is None else dict(job, n_gpus=world, workload=args.workload, shard=args.shard,
                                                 encode_ms_per_step=ms_per_step, encode_Mpoints_per_s=mpts,
                                                 decode_Mpoints_per_s=(points_job / (job["decode_ms_per_step_max_over_ranks"] * 1e-3) / 1e6
                                                                       if job["decode_ms_per_step_max_over_ranks"] else None),
                                                 stage1_bytes=job_bytes),
            "chunk_table": chunk_table,
            "stage1_bytes_per_point": out_bpp,
            "device_ms_per_step": {dominant: regular_ms, "sections": sections_ms,
                                   "k_finish": compact_ms, "all_kernels": device_ms},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_measured_in_run": False,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "read_only_GBps": points_local * step / (regular_ms * 1e-3) / 1e9,
                         "read_only_frac": points_local * step / (regular_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                         "whole_stage1_GBps": points_local * (step + out_bpp) / (device_ms * 1e-3) / 1e9,
                         # `value` times the WHOLE framed step (piece kernel + sections + k_finish + the gaps between the
                         # launches): its algorithmic bytes over ms_per_step. `frac` above describes the dominant kernel alone.
                         "whole_step_ms": ms_per_step,
                         "whole_step_GBps": points_local * (step + out_bpp) / (ms_per_step * 1e-3) / 1e9,
                         "whole_step_frac": points_local * (step + out_bpp) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS},
        }
        if args.cpu_baseline_seconds > 0 and world == 1:
            result["cpu_baseline"] = cpu_baseline(info, distinct[0], args.cpu_baseline_seconds)
            result["speedup_vs_cpu_1core"] = mpts / result["cpu_baseline"]["value"]
        else:
            result["cpu_baseline"] = None
        if args.e2e_seconds > 0 and world == 1 and args.shard == "clouds":
            try:
                result["e2e"] = e2e_figures(info, distinct[0], dev, args.e2e_seconds)
            except Exception as exc:  # the e2e legs must never cost the headline line
                result["e2e"] = {"error": repr(exc)}
        if args.transcode_messages > 0 and world == 1 and args.shard == "clouds":
            try:
                result["transcode"] = transcode_figures(args.transcode_messages)
            except Exception as exc:
                result["transcode"] = {"error": repr(exc)}
        if args.config_legs and world == 1 and args.shard == "clouds" and args.workload == "c2":
            try:
                result["configs"] = config_figures(dev, max(3, args.steps // 2))
            except Exception as exc:
                result["configs"] = {"error": repr(exc)}
        print(json.dumps(result))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
