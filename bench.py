#!/usr/bin/env python3
"""bench.py -- stage-1 encode throughput of the HIP path on BASELINE.json's headline configuration.

A "step" = one cldn_hip_encode_stage1 call over one batch of synthetic clouds that is already resident in
HBM (inputs and outputs are device buffers; nothing crosses PCIe inside the timed region). The default
workload is BASELINE.json configs[1]: 1M-point XYZI (float32 XYZ + uint16 intensity) clouds at 1 mm,
stage 1 only ("pre-ZSTD"), `--clouds` of them per step and per GPU (weak scaling: every rank encodes its
own batch, no data-path collective -- whole clouds are independent).

Prints ONE JSON line on rank 0 (see README/DESIGN.md for the field meanings):
  value            whole-job Mpoints/s over all ranks
  roofline         dominant kernel (k_encode_regular): algorithmic bytes per launch / its average duration
                   measured with HIP events on the codec's stream inside the timed region
  cpu_baseline     the real reference (oracle/_ref, kind "reference") or the C port (kind "port") timed on the
                   host cores on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
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
        ncores = os.cpu_count() or 1
        quota = None
        try:  # containers often grant fewer CPUs than they show (cgroup v2 cpu.max = "<quota> <period>")
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                quota = max(1, int(float(q) / float(per) + 0.5))
        except (OSError, ValueError):
            pass
        if quota is not None:
            ncores = min(ncores, quota)
        if ncores > 1:
            reps_mt = max(3, reps // 4)
            _size, tm = ref.bench_encode(info, cloud, reps=reps_mt, threads=ncores)
            agg = float(np.sum(pts / np.median(tm, axis=1))) / 1e6
            out["all_cores"] = {"value": agg, "cores": ncores, "visible_threads": os.cpu_count(), "cgroup_cpu_quota": quota,
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOAD_DESC))
    ap.add_argument("--clouds", type=int, default=32, help="clouds per step per GPU")
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--distinct", type=int, default=4, help="distinct synthetic clouds generated (tiled to --clouds)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run also a single rank takes the RCCL path
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("nccl", device_id=dev)

    from cloudini_amd import native

    info, distinct = make_workload(args.workload, args.points, max(1, min(args.distinct, args.clouds)))
    step = info.point_step
    pts_per_cloud = len(distinct[0]) // step
    n_clouds = args.clouds
    host = np.concatenate([distinct[(k + rank) % len(distinct)] for k in range(n_clouds)])
    d_points = torch.from_numpy(host).to(dev)
    cloud_points = np.full(n_clouds, pts_per_cloud, dtype=np.uint64)

    plan = native.Plan(info)
    stream = torch.cuda.current_stream(dev)
    codec = native.Codec(plan, device=local_rank, stream=stream.cuda_stream)
    cap = plan.stage1_bound(pts_per_cloud) * n_clouds
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_offsets = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
    n_chunks = n_clouds * ((pts_per_cloud + 32767) // 32768)
    d_chunk_sizes = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    d_modes = torch.zeros(max(1, n_clouds * max(1, plan.adaptive_fields)), dtype=torch.uint8, device=dev)

    def one_step():
        codec.encode_device(d_points.data_ptr(), cloud_points, d_out.data_ptr(), cap, d_offsets.data_ptr(),
                            d_chunk_sizes.data_ptr(), d_modes.data_ptr())

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_step()
    codec.status()  # raises on device-side errors
    codec.enable_timing(max(1, args.steps))

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
    codec.status()

    kms = [codec.kernel_ms(s) for s in range(args.steps)]
    regular_ms = float(np.mean([k["regular"] for k in kms]))
    sections_ms = float(np.mean([k["sections"] for k in kms]))
    compact_ms = float(np.mean([k["compact"] for k in kms]))
    device_ms = float(np.mean([k["total"] for k in kms]))

    offsets = d_offsets.cpu().numpy()
    total_out = int(offsets[-1])

    # extra (not `value`): decode of the streams just produced, device resident, same synchronisation bracket
    d_dec = torch.empty(host.size, dtype=torch.uint8, device=dev)
    so = offsets.astype(np.uint64)
    dec_steps = max(1, args.steps // 2)
    for _ in range(min(2, args.warmup)):
        codec.decode_device(d_out.data_ptr(), so, cloud_points, d_dec.data_ptr(), host.size)
    barrier()
    t1 = time.perf_counter()
    for _ in range(dec_steps):
        codec.decode_device(d_out.data_ptr(), so, cloud_points, d_dec.data_ptr(), host.size)
    torch.cuda.synchronize(dev)
    dec_elapsed = time.perf_counter() - t1
    codec.status()
    dec_stats = codec.decode_stats()
    del d_dec
    chunk_sizes = d_chunk_sizes.cpu().numpy().astype(np.int64)
    points_per_step = n_clouds * pts_per_cloud
    out_bpp = total_out / points_per_step

    if rank == 0:
        # algorithmic bytes of the dominant kernel (k_encode_regular): every input byte read once + its own output
        # (the interleaved float stream); the integer column hand-off to the section kernel is not counted.
        with torch.no_grad():
            pass
        total_points_all = points_per_step * world
        ms_per_step = elapsed / args.steps * 1e3
        mpts = total_points_all * args.steps / elapsed / 1e6
        # regular-stream bytes per point: payload minus sections. Measured from chunk sizes when there are no
        # adaptive fields; otherwise derived from the oracle-verified layout: sections are the tail of each chunk.
        reg_bpp = out_bpp - 4.0 * n_chunks / points_per_step
        if plan.adaptive_fields:
            # same buffer, schema restricted to the float fields: the HIP codec's own stream of that schema is the
            # regular stream of the full one (V5 == V4 bytes for float-only clouds, test_field_encoders.cpp:695-769)
            float_only = info.copy(fields=[f for f in info.fields if int(f.type) == 7])
            fo_codec = native.Codec(native.Plan(float_only), device=local_rank)
            fo_stream = fo_codec.encode_host([distinct[rank % len(distinct)]])[0][0]
            fo_codec.close()
            n_ch = (pts_per_cloud + 32767) // 32768
            reg_bpp = (len(fo_stream) - 4 * n_ch) / pts_per_cloud
        alg_bytes = points_per_step * (step + reg_bpp)
        achieved = alg_bytes / (regular_ms * 1e-3) / 1e9
        # the regular-stream kernel of this plan: k_encode_floatn when the stream is exactly one fused FloatN encoder
        lead = 0
        for f in info.fields:
            if int(f.type) == 7 and f.resolution is not None:
                lead += 1
            else:
                break
        rest_adaptive = all(int(f.type) in (3, 4, 5, 6, 9, 10) for f in info.fields[lead:])
        dominant = "k_encode_floatn" if lead in (3, 4) and rest_adaptive and int(info.version) >= 5 else "k_encode_regular"
        # HBM bytes per launch of the same kernel from the PMC passes of tools/profile_round.sh (rocprofv3 cannot
        # collect counters from inside this process); only quoted when the committed pass ran this very workload
        traffic, traffic_src = None, None
        for name in sorted(os.listdir(os.path.join(ROOT, "profiles")), reverse=True) if os.path.isdir(os.path.join(ROOT, "profiles")) else []:
            if name.endswith("_traffic.json"):
                try:
                    t = json.load(open(os.path.join(ROOT, "profiles", name)))
                except (OSError, ValueError):
                    continue
                if (t.get("workload") == args.workload and t.get("clouds_per_gpu") == n_clouds
                        and t.get("points_per_cloud") == pts_per_cloud):
                    traffic, traffic_src = float(t["hbm_bytes_per_launch"]), "profiles/" + name
                    break
        result = {
            "metric": "encode Mpoints/s (stage-1, pre-ZSTD)",
            "value": mpts,
            "unit": "Mpoints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32->i32 (u8 varint stream)",
            "data": f"synthetic lidar64 generator (cloudini_amd/synth.py), {len(distinct)} distinct clouds tiled, "
                    "inputs resident in HBM",
            "config": {"workload": WORKLOAD_DESC[args.workload].format(clouds=n_clouds, points=pts_per_cloud),
                       "clouds_per_gpu": n_clouds, "points_per_cloud": pts_per_cloud, "point_step": step,
                       "parallelism": f"whole clouds sharded over {world} GPU(s), no data-path collective"},
            "input_MBps": total_points_all * step * args.steps / elapsed / 1e6,
            "decode": {"value": points_per_step * dec_steps / dec_elapsed / 1e6, "unit": "Mpoints/s (rank 0, stage-1 decode, "
                       "device resident)", "ms_per_step": dec_elapsed / dec_steps * 1e3,
                       "chunks_parallel_regular/parallel_sections/serial/serial_sections": list(dec_stats)},
            "stage1_bytes_per_point": out_bpp,
            "device_ms_per_step": {dominant: regular_ms, "sections": sections_ms,
                                   "offsets+compact": compact_ms, "all_kernels": device_ms},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "read_only_GBps": points_per_step * step / (regular_ms * 1e-3) / 1e9,
                         "whole_stage1_GBps": points_per_step * (step + out_bpp) / (device_ms * 1e-3) / 1e9},
        }
        if args.cpu_baseline_seconds > 0 and world == 1:
            result["cpu_baseline"] = cpu_baseline(info, distinct[0], args.cpu_baseline_seconds)
            result["speedup_vs_cpu_1core"] = mpts / result["cpu_baseline"]["value"]
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
