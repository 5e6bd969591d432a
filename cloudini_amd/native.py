"""ctypes binding of the C ABI in include/cloudini_hip.h (cloudini_amd/lib/libcloudini_hip.so).

This is the product path's only entry to the HIP kernels from Python (tests, bench.py, smoke). It fails
loudly when the library has not been built or when no GPU is present -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("CLDN_HIP_LIB_OVERRIDE") or os.path.join(HERE, "lib", "libcloudini_hip.so")  # override: A/B builds

HOST, DEVICE = 0, 1

ERRORS = {-1: "ARG", -2: "CAPACITY", -3: "UNSUPPORTED", -4: "DEVICE", -5: "NO_DEVICE", -6: "CORRUPT", -7: "NOMEM"}


class _Segment(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("size", C.c_uint32)]


class ChunkTable(C.Structure):
    """cldn_hip_chunk_table_t: device pointers into the codec's workspace (valid until the codec's next call)."""
    _fields_ = [("payload_base", C.c_void_p), ("chunk_stride", C.c_uint64), ("segments", C.c_void_p),
                ("segments_per_chunk", C.c_uint32), ("n_chunks", C.c_uint32), ("chunk_sizes", C.c_void_p),
                ("not_contiguous", C.c_void_p)]


class CloudiniHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"cloudini_hip error {code} ({ERRORS.get(code, '?')}): {message}")
        self.code = code
        self.message = message


class _Field(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("type", C.c_uint8), ("has_resolution", C.c_uint8),
                ("reserved", C.c_uint8 * 2), ("resolution", C.c_float)]


_lib = None


def lib() -> C.CDLL:
    """Load libcloudini_hip.so (once). Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(HIP_SO):
        raise ImportError(
            f"{HIP_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(cloudini_amd has no CPU fallback)")
    # torch bundles its own HIP runtime (same SONAME, libamdhip64.so.7): whichever copy is loaded first serves the
    # whole process. When torch is going to provide device memory / streams it must be the first one in.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(HIP_SO)
    vp, u8p, u32p, u64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.cldn_hip_last_error.restype = C.c_char_p
    L.cldn_hip_abi_version.restype = C.c_int
    L.cldn_hip_device_count.restype = C.c_int
    L.cldn_hip_current_device.restype = C.c_int
    L.cldn_hip_set_current_device.restype = C.c_int
    L.cldn_hip_set_current_device.argtypes = [C.c_int]
    L.cldn_hip_codec_device.restype = C.c_int
    L.cldn_hip_codec_device.argtypes = [vp]
    L.cldn_hip_host_alloc.restype = vp
    L.cldn_hip_host_alloc.argtypes = [C.c_size_t]
    L.cldn_hip_host_free.restype = None
    L.cldn_hip_host_free.argtypes = [vp]
    L.cldn_hip_plan_create.restype = C.c_int
    L.cldn_hip_plan_create.argtypes = [C.POINTER(_Field), C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8,
                                       C.POINTER(vp)]
    L.cldn_hip_plan_destroy.argtypes = [vp]
    L.cldn_hip_plan_destroy.restype = None
    L.cldn_hip_plan_uses_v5.argtypes = [vp]
    L.cldn_hip_plan_uses_v5.restype = C.c_int
    L.cldn_hip_plan_adaptive_fields.argtypes = [vp]
    L.cldn_hip_plan_adaptive_fields.restype = C.c_uint32
    L.cldn_hip_plan_max_point_bytes.argtypes = [vp]
    L.cldn_hip_plan_max_point_bytes.restype = C.c_uint32
    L.cldn_hip_stage1_bound.argtypes = [vp, C.c_uint64]
    L.cldn_hip_stage1_bound.restype = C.c_uint64
    L.cldn_hip_codec_create.restype = C.c_int
    L.cldn_hip_codec_create.argtypes = [vp, C.c_int, vp, C.POINTER(vp)]
    L.cldn_hip_codec_destroy.argtypes = [vp]
    L.cldn_hip_codec_destroy.restype = None
    L.cldn_hip_codec_synchronize.argtypes = [vp]
    L.cldn_hip_codec_synchronize.restype = C.c_int
    L.cldn_hip_codec_stream.argtypes = [vp]
    L.cldn_hip_codec_stream.restype = vp
    L.cldn_hip_codec_status.argtypes = [vp]
    L.cldn_hip_codec_status.restype = C.c_int
    L.cldn_hip_codec_enable_timing.argtypes = [vp, C.c_uint32]
    L.cldn_hip_codec_enable_timing.restype = C.c_int
    L.cldn_hip_codec_kernel_ms.argtypes = [vp, C.c_uint32, C.POINTER(C.c_float)]
    L.cldn_hip_codec_kernel_ms.restype = C.c_int
    L.cldn_hip_codec_decode_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.cldn_hip_codec_decode_ms.restype = C.c_int
    L.cldn_hip_viz_preprocess.restype = C.c_int
    L.cldn_hip_viz_preprocess.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, vp, C.c_uint64,
                                          C.c_int, u64p]
    L.cldn_hip_codec_set_decode_fill.argtypes = [vp, C.c_int]
    L.cldn_hip_codec_set_decode_fill.restype = C.c_int
    L.cldn_hip_codec_decode_stats.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.cldn_hip_codec_decode_stats.restype = C.c_int
    L.cldn_hip_codec_finish_retries.argtypes = [vp]
    L.cldn_hip_codec_finish_retries.restype = C.c_uint32
    L.cldn_hip_codec_force_modes.argtypes = [vp, C.POINTER(C.c_uint8), C.c_uint32]
    L.cldn_hip_codec_force_modes.restype = C.c_int
    L.cldn_hip_encode_stage1_chunks.restype = C.c_int
    L.cldn_hip_encode_stage1_chunks.argtypes = [vp, vp, C.c_int, u64p, C.c_uint32, C.POINTER(ChunkTable), vp]
    L.cldn_hip_frame_chunks.restype = C.c_int
    L.cldn_hip_frame_chunks.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, vp]
    L.cldn_hip_codec_set_stage2.argtypes = [vp, C.c_int]
    L.cldn_hip_codec_set_stage2.restype = C.c_int
    L.cldn_hip_stage2_bound.argtypes = [vp, C.c_uint64, C.c_int]
    L.cldn_hip_stage2_bound.restype = C.c_uint64
    L.cldn_hip_codec_pipeline.argtypes = [vp, C.c_int, vp]
    L.cldn_hip_codec_pipeline.restype = C.c_int
    L.cldn_hip_encode_stage1.restype = C.c_int
    L.cldn_hip_encode_stage1.argtypes = [vp, vp, C.c_int, u64p, C.c_uint32, vp, C.c_uint64, C.c_int, vp, vp, vp]
    L.cldn_hip_encode_stage1_gather.restype = C.c_int
    L.cldn_hip_encode_stage1_gather.argtypes = [vp, C.POINTER(vp), u64p, C.c_uint32, vp, C.c_uint64, C.c_int, vp, vp, vp]
    L.cldn_hip_codec_fetch_output.restype = C.c_int
    L.cldn_hip_codec_fetch_output.argtypes = [vp, vp, C.c_uint64]
    L.cldn_hip_decode_stage1.restype = C.c_int
    L.cldn_hip_decode_stage1.argtypes = [vp, vp, C.c_int, u64p, u64p, C.c_uint32, vp, C.c_uint64, C.c_int]
    L.cldn_hip_decode_stage1_sized.restype = C.c_int
    L.cldn_hip_decode_stage1_sized.argtypes = [vp, vp, C.c_int, u64p, u64p, C.c_uint32, vp, C.c_int, vp, C.c_uint64, C.c_int]
    L.cldn_hip_decode_stage1_unframed.restype = C.c_int
    L.cldn_hip_decode_stage1_unframed.argtypes = [vp, vp, C.c_uint64, C.c_int, vp, C.c_uint64, C.c_int]
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc < 0:
        raise CloudiniHipError(rc, lib().cldn_hip_last_error().decode(errors="replace"))


def device_count() -> int:
    n = lib().cldn_hip_device_count()
    _check(n)
    return n


def _fields_array(info):
    arr = (_Field * max(1, len(info.fields)))()
    for i, f in enumerate(info.fields):
        arr[i].offset = int(f.offset)
        arr[i].type = int(f.type)
        arr[i].has_resolution = 0 if f.resolution is None else 1
        arr[i].resolution = 0.0 if f.resolution is None else float(f.resolution)
    return arr


class Plan:
    """cldn_hip_plan_t: the encoder/decoder selection for an EncodingInfo."""

    def __init__(self, info):
        self._h = C.c_void_p()
        arr = _fields_array(info)
        _check(lib().cldn_hip_plan_create(arr, len(info.fields), int(info.point_step), int(info.version),
                                          int(info.encoding_opt), C.byref(self._h)))
        self.point_step = int(info.point_step)

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.cldn_hip_plan_destroy(self._h)
            self._h = C.c_void_p()

    @property
    def uses_v5(self) -> bool:
        return bool(lib().cldn_hip_plan_uses_v5(self._h))

    @property
    def adaptive_fields(self) -> int:
        return int(lib().cldn_hip_plan_adaptive_fields(self._h))

    @property
    def max_point_bytes(self) -> int:
        return int(lib().cldn_hip_plan_max_point_bytes(self._h))

    def stage1_bound(self, n_points: int) -> int:
        return int(lib().cldn_hip_stage1_bound(self._h, int(n_points)))

    def stage2_bound(self, n_points: int, stage2: int) -> int:
        return int(lib().cldn_hip_stage2_bound(self._h, int(n_points), int(stage2)))


class Codec:
    """cldn_hip_codec_t: device, stream and workspace. `stream` is a raw hipStream_t value (int) or None."""

    def __init__(self, plan: Plan, device: int = -1, stream: Optional[int] = None):
        self.plan = plan
        self._h = C.c_void_p()
        _check(lib().cldn_hip_codec_create(plan._h, device, C.c_void_p(stream or 0), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value and _lib is not None:
            _lib.cldn_hip_codec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()

    def synchronize(self):
        _check(lib().cldn_hip_codec_synchronize(self._h))

    def status(self):
        _check(lib().cldn_hip_codec_status(self._h))

    def viz_preprocess_host(self, cloud, point_step: int, xyz_offset: int, resolution: float) -> np.ndarray:
        """applyVizLossyPreprocessing's data path on host buffers: the surviving points, order preserved."""
        data = np.ascontiguousarray(cloud).view(np.uint8).reshape(-1)
        n = data.size // point_step
        out = np.empty(max(1, data.size), dtype=np.uint8)
        kept = C.c_uint64(0)
        _check(lib().cldn_hip_viz_preprocess(self._h, data.ctypes.data_as(C.c_void_p), HOST, n, point_step, xyz_offset,
                                             resolution, out.ctypes.data_as(C.c_void_p), out.size, HOST, C.byref(kept)))
        return out[: kept.value * point_step].copy()

    def viz_preprocess_device(self, points_ptr: int, n_points: int, point_step: int, xyz_offset: int, resolution: float,
                              out_ptr: int, out_capacity: int) -> int:
        kept = C.c_uint64(0)
        _check(lib().cldn_hip_viz_preprocess(self._h, C.c_void_p(points_ptr), DEVICE, n_points, point_step, xyz_offset,
                                             resolution, C.c_void_p(out_ptr), out_capacity, DEVICE, C.byref(kept)))
        return int(kept.value)

    def set_decode_fill(self, zero: bool):
        """cldn_hip_codec_set_decode_fill: True = bytes no field covers may be written as 0 (no round trip of a host buffer)."""
        _check(lib().cldn_hip_codec_set_decode_fill(self._h, 1 if zero else 0))

    def decode_stats(self):
        """Chunks of the last decode call per kernel: (fast regular, fast sections, serial chunks, serial sections)."""
        v = (C.c_uint32 * 4)()
        _check(lib().cldn_hip_codec_decode_stats(self._h, v))
        return tuple(int(x) for x in v)

    def finish_retries(self) -> int:
        """Calls redone through k_finish's ticket order after ST_FINISH_TIMEOUT (cldn_hip_codec_finish_retries)."""
        return int(lib().cldn_hip_codec_finish_retries(self._h))

    def force_modes(self, modes=None):
        """Adaptive-int modes committed elsewhere (cldn_hip_codec_force_modes); None / empty returns to probing."""
        if modes is None or len(modes) == 0:
            _check(lib().cldn_hip_codec_force_modes(self._h, None, 0))
            return
        m = np.ascontiguousarray(modes, dtype=np.uint8)
        _check(lib().cldn_hip_codec_force_modes(self._h, m.ctypes.data_as(C.POINTER(C.c_uint8)), m.size))

    def set_stage2(self, stage2: int):
        """0 = stage-1 streams (default), 1 = [u32 size][LZ4 block] per chunk, compressed on the device, 2 = the same with the
        FAST parameters (4 KiB sub-ranges: ~1.5 x the speed, blocks ~3 % larger)."""
        _check(lib().cldn_hip_codec_set_stage2(self._h, int(stage2)))
        self._stage2 = int(stage2)

    def pipeline(self, mode: int = 0, points_ptr: int = 0) -> int:
        """Choose the encoder pipeline (cldn_hip_codec_pipeline: 0 auto, 1 tile kernel + slots, 2 piece kernel + slots);
        returns the pipeline the next call takes."""
        r = lib().cldn_hip_codec_pipeline(self._h, int(mode), C.c_void_p(points_ptr))
        _check(r)
        return int(r)

    def enable_timing(self, n_slots: int):
        _check(lib().cldn_hip_codec_enable_timing(self._h, int(n_slots)))

    def kernel_ms(self, slot: int):
        v = (C.c_float * 4)()
        _check(lib().cldn_hip_codec_kernel_ms(self._h, int(slot), v))
        return {"regular": v[0], "sections": v[1], "compact": v[2], "total": v[3]}

    def decode_ms(self):
        """HIP-event times of the last decode call (timing enabled): the regular-stream kernel, everything launched."""
        v = (C.c_float * 2)()
        _check(lib().cldn_hip_codec_decode_ms(self._h, v))
        return {"regular_kernel": v[0], "total": v[1]}

    # ---- host buffers (numpy) -------------------------------------------------------------------------
    def encode_host(self, clouds: Sequence[np.ndarray]):
        """Encode a batch of host-resident clouds. Returns (list of framed stage-1 streams, chunk_sizes, modes)."""
        step = self.plan.point_step
        arrs = [np.ascontiguousarray(c).view(np.uint8).reshape(-1) for c in clouds]
        npts = np.array([a.size // step for a in arrs], dtype=np.uint64)
        for a in arrs:
            if a.size % step:
                raise ValueError("Input cloud_data size is not a multiple of point_step")
        data = np.concatenate(arrs) if arrs else np.zeros(0, np.uint8)
        cap = int(sum(self.plan.stage2_bound(int(n), getattr(self, "_stage2", 0)) for n in npts))
        out = np.empty(max(cap, 1), dtype=np.uint8)
        offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
        n_chunks = int(sum((int(n) + 32767) // 32768 for n in npts))
        chunk_sizes = np.zeros(max(1, n_chunks), dtype=np.uint32)
        na = self.plan.adaptive_fields
        modes = np.zeros(max(1, len(arrs) * max(1, na)), dtype=np.uint8)
        _check(lib().cldn_hip_encode_stage1(
            self._h, data.ctypes.data_as(C.c_void_p), HOST, npts.ctypes.data_as(C.POINTER(C.c_uint64)), len(arrs),
            out.ctypes.data_as(C.c_void_p), cap, HOST, offs.ctypes.data_as(C.c_void_p),
            chunk_sizes.ctypes.data_as(C.c_void_p), modes.ctypes.data_as(C.c_void_p)))
        streams = [out[int(offs[k]):int(offs[k + 1])].copy() for k in range(len(arrs))]
        return streams, chunk_sizes[:n_chunks], modes[: len(arrs) * na].reshape(len(arrs), na)

    # ---- chunk-table output ----------------------------------------------------------------------------
    def encode_chunks_device(self, points_ptr: int, cloud_points: np.ndarray, modes_ptr: int = 0) -> ChunkTable:
        """cldn_hip_encode_stage1_chunks on device-resident points: stage 1 without the framing (asynchronous)."""
        cp = np.ascontiguousarray(cloud_points, dtype=np.uint64)
        table = ChunkTable()
        _check(lib().cldn_hip_encode_stage1_chunks(self._h, C.c_void_p(points_ptr), DEVICE, cp.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   cp.size, C.byref(table), C.c_void_p(modes_ptr)))
        return table

    def frame_chunks_device(self, out_ptr: int, out_capacity: int, stream_offsets_ptr: int = 0, chunk_sizes_ptr: int = 0):
        _check(lib().cldn_hip_frame_chunks(self._h, C.c_void_p(out_ptr), int(out_capacity), DEVICE, C.c_void_p(stream_offsets_ptr),
                                           C.c_void_p(chunk_sizes_ptr)))

    # ---- device buffers (raw pointers, e.g. torch tensors' data_ptr()) ---------------------------------
    def encode_device(self, points_ptr: int, cloud_points: np.ndarray, out_ptr: int, out_capacity: int,
                      stream_offsets_ptr: int = 0, chunk_sizes_ptr: int = 0, modes_ptr: int = 0):
        cp = np.ascontiguousarray(cloud_points, dtype=np.uint64)
        _check(lib().cldn_hip_encode_stage1(
            self._h, C.c_void_p(points_ptr), DEVICE, cp.ctypes.data_as(C.POINTER(C.c_uint64)), cp.size,
            C.c_void_p(out_ptr), int(out_capacity), DEVICE, C.c_void_p(stream_offsets_ptr),
            C.c_void_p(chunk_sizes_ptr), C.c_void_p(modes_ptr)))

    def decode_host(self, streams: Sequence[np.ndarray], cloud_points: Sequence[int],
                    out: Optional[np.ndarray] = None, chunk_sizes=None) -> List[np.ndarray]:
        step = self.plan.point_step
        arrs = [np.ascontiguousarray(s).view(np.uint8).reshape(-1) for s in streams]
        offs = np.zeros(len(arrs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([a.size for a in arrs])
        data = np.concatenate(arrs) if arrs else np.zeros(0, np.uint8)
        if data.size == 0:
            data = np.zeros(1, np.uint8)
        npts = np.array(list(cloud_points), dtype=np.uint64)
        total = int(npts.sum()) * step
        if out is None:
            out = np.zeros(max(1, total), dtype=np.uint8)
        cs = None if chunk_sizes is None else np.ascontiguousarray(chunk_sizes, dtype=np.uint32)
        if cs is not None:
            n_chunks = int(sum((int(n) + 32767) // 32768 for n in npts))
            if cs.size < n_chunks:  # the C side reads n_chunks sizes
                raise ValueError(f"chunk_sizes has {cs.size} entries, the batch has {n_chunks} chunks")
        _check(lib().cldn_hip_decode_stage1_sized(
            self._h, data.ctypes.data_as(C.c_void_p), HOST, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
            npts.ctypes.data_as(C.POINTER(C.c_uint64)), len(arrs), None if cs is None else cs.ctypes.data_as(C.c_void_p), HOST,
            out.ctypes.data_as(C.c_void_p), total, HOST))
        res, pos = [], 0
        for n in npts:
            res.append(out[pos:pos + int(n) * step])
            pos += int(n) * step
        return res

    def decode_device(self, streams_ptr: int, stream_offsets: np.ndarray, cloud_points: np.ndarray, out_ptr: int,
                      out_capacity: int, chunk_sizes_ptr: int = 0):
        """chunk_sizes_ptr: device array of the chunks' payload sizes (what encode_device reported): the chunk table is then
        built in parallel (cldn_hip_decode_stage1_sized)."""
        so = np.ascontiguousarray(stream_offsets, dtype=np.uint64)
        cp = np.ascontiguousarray(cloud_points, dtype=np.uint64)
        _check(lib().cldn_hip_decode_stage1_sized(
            self._h, C.c_void_p(streams_ptr), DEVICE, so.ctypes.data_as(C.POINTER(C.c_uint64)),
            cp.ctypes.data_as(C.POINTER(C.c_uint64)), cp.size, C.c_void_p(chunk_sizes_ptr), DEVICE, C.c_void_p(out_ptr),
            int(out_capacity), DEVICE))
