"""TEST INFRASTRUCTURE ONLY: ctypes bindings for the CPU oracle.

* ``Oracle``   -> oracle/libcloudini_oracle.so (plain-C restatement, cloudini_oracle.c)
* ``RefLib``   -> oracle/_ref/libcloudini_ref.so (the real reference, compiled by oracle/Makefile
                  from /root/reference; prebuilt file travels to the GPU box)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module, and only as
the checker / the reported CPU baseline. Nothing under cloudini_amd/ imports it.

`info` arguments are duck-typed: anything with .fields[i].{name,offset,type,resolution}, .point_step,
.width, .height, .encoding_opt, .compression_opt, .version, .use_threads (cloudini_amd.schema.EncodingInfo
fits).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional, Tuple

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libcloudini_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libcloudini_ref.so")


def build(force: bool = False) -> None:
    """Compile the C restatement and (when /root/reference exists) the reference itself."""
    args = ["make", "-C", HERE]
    if force:
        args.append("-B")
    subprocess.run(args, check=True, stdout=subprocess.DEVNULL)


class _OrcField(C.Structure):
    _fields_ = [("offset", C.c_uint32), ("type", C.c_uint8), ("has_resolution", C.c_uint8),
                ("pad", C.c_uint8 * 2), ("resolution", C.c_float)]


class _OrcSchema(C.Structure):
    _fields_ = [("fields", C.POINTER(_OrcField)), ("n_fields", C.c_uint32), ("point_step", C.c_uint32),
                ("encoding_opt", C.c_uint8), ("version", C.c_uint8)]


class _RefField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint32), ("type", C.c_uint8),
                ("has_resolution", C.c_uint8), ("pad", C.c_uint8 * 2), ("resolution", C.c_float)]


def _as_u8(buf) -> np.ndarray:
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class OracleError(RuntimeError):
    pass


class Oracle:
    """Plain-C restatement (stage 1 only; framed [u32 size][payload] chunks, no header)."""

    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        L = self.lib
        L.orc_encode_stage1.restype = C.c_int64
        L.orc_encode_stage1.argtypes = [C.POINTER(_OrcSchema), C.POINTER(C.c_uint8), C.c_uint64,
                                        C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8), C.c_uint32]
        L.orc_encode_stage1_continued.restype = C.c_int64
        L.orc_encode_stage1_continued.argtypes = [C.POINTER(_OrcSchema), C.POINTER(C.c_uint8), C.c_uint64,
                                                  C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8), C.c_uint32]
        L.orc_viz_preprocess.restype = C.c_int64
        L.orc_viz_preprocess.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32, C.c_uint32, C.c_float,
                                         C.POINTER(C.c_uint8)]
        L.orc_decode_stage1.restype = C.c_int64
        L.orc_decode_stage1.argtypes = [C.POINTER(_OrcSchema), C.POINTER(C.c_uint8), C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_uint8)]
        L.orc_stage1_bound.restype = C.c_uint64
        L.orc_stage1_bound.argtypes = [C.POINTER(_OrcSchema), C.c_uint64]
        L.orc_max_point_bytes.restype = C.c_uint64
        L.orc_max_point_bytes.argtypes = [C.POINTER(_OrcSchema)]
        for name in ("orc_uses_v5", "orc_leading_lossy_floats", "orc_adaptive_field_count"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.POINTER(_OrcSchema)]
        L.orc_encode_varint64.restype = C.c_size_t
        L.orc_encode_varint64.argtypes = [C.c_int64, C.POINTER(C.c_uint8)]
        L.orc_decode_varint.restype = C.c_int
        L.orc_decode_varint.argtypes = [C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_int64)]

    @staticmethod
    def _schema(info) -> Tuple[_OrcSchema, object]:
        arr = (_OrcField * max(1, len(info.fields)))()
        for i, f in enumerate(info.fields):
            arr[i].offset = int(f.offset)
            arr[i].type = int(f.type)
            arr[i].has_resolution = 0 if f.resolution is None else 1
            arr[i].resolution = 0.0 if f.resolution is None else float(f.resolution)
        s = _OrcSchema(arr, len(info.fields), int(info.point_step), int(info.encoding_opt), int(info.version))
        return s, arr

    def uses_v5(self, info) -> bool:
        s, _keep = self._schema(info)
        return bool(self.lib.orc_uses_v5(C.byref(s)))

    def adaptive_field_count(self, info) -> int:
        s, _keep = self._schema(info)
        return int(self.lib.orc_adaptive_field_count(C.byref(s)))

    def stage1_bound(self, info, n_points: int) -> int:
        s, _keep = self._schema(info)
        return int(self.lib.orc_stage1_bound(C.byref(s), n_points))

    def max_point_bytes(self, info) -> int:
        s, _keep = self._schema(info)
        return int(self.lib.orc_max_point_bytes(C.byref(s)))

    def encode_stage1(self, info, cloud, return_modes: bool = False):
        data = _as_u8(cloud)
        step = int(info.point_step)
        assert data.size % step == 0
        n = data.size // step
        s, _keep = self._schema(info)
        cap = int(self.lib.orc_stage1_bound(C.byref(s), n)) + 64 + 32 * n  # slack beyond the reference bound
        out = np.empty(cap, dtype=np.uint8)
        modes = np.zeros(max(64, len(info.fields)), dtype=np.uint8)
        r = self.lib.orc_encode_stage1(C.byref(s), _ptr(data), n, _ptr(out), cap, _ptr(modes), len(modes))
        if r < 0:
            raise OracleError(f"orc_encode_stage1 failed: {r}")
        res = out[:r].copy()
        if return_modes:
            return res, modes[: self.adaptive_field_count(info)].copy()
        return res

    def encode_stage1_continued(self, info, cloud, modes) -> np.ndarray:
        """Chunks of a larger cloud whose first chunk committed `modes` (orc_encode_stage1_continued)."""
        data = _as_u8(cloud)
        step = int(info.point_step)
        assert data.size % step == 0
        n = data.size // step
        s, _keep = self._schema(info)
        cap = int(self.lib.orc_stage1_bound(C.byref(s), n)) + 64 + 32 * n
        out = np.empty(cap, dtype=np.uint8)
        m = np.ascontiguousarray(modes, dtype=np.uint8)
        r = self.lib.orc_encode_stage1_continued(C.byref(s), _ptr(data), n, _ptr(out), cap,
                                                 _ptr(m) if m.size else None, m.size)
        if r < 0:
            raise OracleError(f"orc_encode_stage1_continued failed: {r}")
        return out[:r].copy()

    def viz_preprocess(self, cloud, point_step: int, xyz_offset: int, resolution: float) -> np.ndarray:
        """Surviving points of applyVizLossyPreprocessing's data path (orc_viz_preprocess)."""
        data = _as_u8(cloud)
        n = data.size // point_step
        out = np.empty(max(1, data.size), dtype=np.uint8)
        r = self.lib.orc_viz_preprocess(_ptr(data) if data.size else None, n, point_step, xyz_offset, resolution,
                                        _ptr(out))
        if r < 0:
            raise OracleError(f"orc_viz_preprocess failed: {r}")
        return out[: r * point_step].copy()

    def decode_stage1(self, info, stream, n_points: int, fill: int = 0) -> np.ndarray:
        st = _as_u8(stream)
        s, _keep = self._schema(info)
        out = np.full(n_points * int(info.point_step), fill, dtype=np.uint8)
        r = self.lib.orc_decode_stage1(C.byref(s), _ptr(st), st.size, n_points, _ptr(out))
        if r < 0:
            raise OracleError(f"orc_decode_stage1 failed: {r}")
        return out

    def lz4_model(self, payload, sub_bytes: int = 8192, hash_bits: int = 11, max_matches: int = 1024) -> np.ndarray:
        """Serial model of the device-side LZ4 block compressor (oracle/lz4_model.c): the block for `payload`."""
        a = _as_u8(payload)
        self.lib.orc_lz4_bound.restype = C.c_uint32
        self.lib.orc_lz4_compress.restype = C.c_int64
        self.lib.orc_lz4_compress.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        cap = int(self.lib.orc_lz4_bound(C.c_uint32(a.size)))
        out = np.empty(max(1, cap), dtype=np.uint8)
        n = self.lib.orc_lz4_compress(a.ctypes.data if a.size else None, a.size, out.ctypes.data, cap, sub_bytes, hash_bits, max_matches)
        if n < 0:
            raise RuntimeError("orc_lz4_compress failed")
        return out[:n].copy()

    def encode_varint64(self, v: int) -> bytes:
        buf = (C.c_uint8 * 16)()
        n = self.lib.orc_encode_varint64(v, buf)
        return bytes(buf[:n])

    def decode_varint(self, data: bytes, max_size: Optional[int] = None):
        a = np.frombuffer(data + b"\0", dtype=np.uint8).copy()
        val = C.c_int64(0)
        n = self.lib.orc_decode_varint(_ptr(a), len(data) if max_size is None else max_size, C.byref(val))
        return n, val.value


class RefLib:
    """The real reference (oracle/_ref). Raises FileNotFoundError when it has not been built."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            if os.path.exists("/root/reference/cloudini_lib/src/cloudini.cpp"):
                build()
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        L = self.lib
        L.ref_last_error.restype = C.c_char_p
        common = [C.POINTER(_RefField), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8,
                  C.c_uint8]
        L.ref_encode.restype = C.c_int64
        L.ref_encode.argtypes = common + [C.c_uint8, C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8),
                                          C.c_uint64]
        L.ref_max_compressed_size.restype = C.c_int64
        L.ref_max_compressed_size.argtypes = common + [C.c_uint64, C.c_uint8]
        L.ref_encode_header.restype = C.c_int64
        L.ref_encode_header.argtypes = common + [C.c_uint8, C.POINTER(C.c_uint8), C.c_uint64]
        L.ref_decode.restype = C.c_int64
        L.ref_decode.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64, C.c_char_p,
                                 C.c_uint64]
        L.ref_decode_noheader.restype = C.c_int64
        L.ref_decode_noheader.argtypes = common + [C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8),
                                                   C.c_uint64]
        L.ref_ros_compress.restype = C.c_int64
        L.ref_ros_compress.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_float, C.c_uint8,
                                       C.POINTER(C.c_uint8), C.c_uint64]
        L.ref_ros_decompress.restype = C.c_int64
        L.ref_ros_decompress.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64]
        L.ref_ros_describe.restype = C.c_int64
        L.ref_ros_describe.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.c_char_p, C.c_uint64,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ref_viz_preprocess.restype = C.c_int64
        L.ref_viz_preprocess.argtypes = [C.POINTER(_RefField), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint8), C.c_uint64,
                                         C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_float),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.ref_bench_encode.restype = C.c_int64
        L.ref_bench_encode.argtypes = common + [C.c_uint8, C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32,
                                                C.c_uint32, C.POINTER(C.c_double)]
        L.ref_bench_decode.restype = C.c_int64
        L.ref_bench_decode.argtypes = [C.POINTER(C.c_uint8), C.c_uint64, C.POINTER(C.c_uint8), C.c_uint64,
                                       C.c_uint32, C.POINTER(C.c_double)]

    def _err(self, what: str):
        raise OracleError(f"{what}: {self.lib.ref_last_error().decode(errors='replace')}")

    @staticmethod
    def _fields(info):
        arr = (_RefField * max(1, len(info.fields)))()
        keep: List[bytes] = []
        for i, f in enumerate(info.fields):
            nm = f.name.encode()
            keep.append(nm)
            arr[i].name = nm
            arr[i].offset = int(f.offset)
            arr[i].type = int(f.type)
            arr[i].has_resolution = 0 if f.resolution is None else 1
            arr[i].resolution = 0.0 if f.resolution is None else float(f.resolution)
        return arr, keep

    def _common(self, info):
        arr, keep = self._fields(info)
        return (arr, len(info.fields), int(info.point_step), int(info.width), int(info.height),
                int(info.encoding_opt), int(info.compression_opt), int(info.version)), keep

    def max_compressed_size(self, info, n_points: int, include_header: bool = True) -> int:
        args, _keep = self._common(info)
        r = self.lib.ref_max_compressed_size(*args, n_points, 1 if include_header else 0)
        if r < 0:
            self._err("MaxCompressedSize")
        return int(r)

    def encode(self, info, cloud) -> np.ndarray:
        """Full stream: header + chunks (PointcloudEncoder::encode(view, view, write_header=true))."""
        data = _as_u8(cloud)
        n = data.size // int(info.point_step)
        cap = self.max_compressed_size(info, n, True)
        out = np.empty(cap, dtype=np.uint8)
        args, _keep = self._common(info)
        r = self.lib.ref_encode(*args, 1 if info.use_threads else 0, _ptr(data), data.size, _ptr(out), cap)
        if r < 0:
            self._err("encode")
        return out[:r].copy()

    def header(self, info, binary: bool = False) -> bytes:
        args, _keep = self._common(info)
        out = np.empty(1 << 16, dtype=np.uint8)
        r = self.lib.ref_encode_header(*args, 1 if binary else 0, _ptr(out), out.size)
        if r < 0:
            self._err("EncodeHeader")
        return out[:r].tobytes()

    def encode_stage1(self, info, cloud) -> np.ndarray:
        """Framed stage-1 stream (header stripped); info.compression_opt must be NONE."""
        assert int(info.compression_opt) == 0
        full = self.encode(info, cloud)
        hdr = self.header(info)
        assert full[: len(hdr)].tobytes() == hdr
        return full[len(hdr):].copy()

    def decode(self, stream, out_size: int, fill: int = 0):
        st = _as_u8(stream)
        out = np.full(out_size, fill, dtype=np.uint8)
        yaml = C.create_string_buffer(1 << 16)
        r = self.lib.ref_decode(_ptr(st), st.size, _ptr(out), out.size, yaml, len(yaml))
        if r < 0:
            self._err("decode")
        return out[:r], yaml.value.decode()

    def decode_noheader(self, info, data, fill: int = 0) -> np.ndarray:
        st = _as_u8(data)
        size = int(info.width) * int(info.height) * int(info.point_step)
        out = np.full(size, fill, dtype=np.uint8)
        args, _keep = self._common(info)
        r = self.lib.ref_decode_noheader(*args, _ptr(st), st.size, _ptr(out), out.size)
        if r < 0:
            self._err("decode")
        return out

    def ros_compress(self, dds, resolution: float, compression_opt: int) -> np.ndarray:
        msg = _as_u8(dds)
        out = np.empty(msg.size * 3 + (1 << 20), dtype=np.uint8)
        r = self.lib.ref_ros_compress(_ptr(msg), msg.size, resolution, compression_opt, _ptr(out), out.size)
        if r < 0:
            self._err("ros_compress")
        return out[:r].copy()

    def ros_decompress(self, dds, capacity: int) -> np.ndarray:
        msg = _as_u8(dds)
        out = np.empty(capacity, dtype=np.uint8)
        r = self.lib.ref_ros_decompress(_ptr(msg), msg.size, _ptr(out), out.size)
        if r < 0:
            self._err("ros_decompress")
        return out[:r].copy()

    def ros_describe(self, dds):
        msg = _as_u8(dds)
        text = C.create_string_buffer(1 << 16)
        off = C.c_uint64(0)
        size = C.c_uint64(0)
        r = self.lib.ref_ros_describe(_ptr(msg), msg.size, text, len(text), C.byref(off), C.byref(size))
        if r < 0:
            self._err("ros_describe")
        return text.value.decode(), int(off.value), int(size.value)

    def viz_preprocess(self, info, cloud):
        """applyVizLossyPreprocessing of the reference: (surviving bytes, per-field resolution afterwards, width, height)."""
        data = _as_u8(cloud)
        arr, keep = self._fields(info)
        out = np.empty(max(1, data.size), dtype=np.uint8)
        res = (C.c_float * max(1, len(info.fields)))()
        w, h = C.c_uint32(0), C.c_uint32(0)
        r = self.lib.ref_viz_preprocess(arr, len(info.fields), int(info.point_step), _ptr(data) if data.size else None,
                                        data.size, _ptr(out), out.size, res, C.byref(w), C.byref(h))
        if r < 0:
            self._err("ref_viz_preprocess")
        return out[:r].copy(), [float(x) for x in res][: len(info.fields)], int(w.value), int(h.value)

    def bench_encode(self, info, cloud, reps: int = 10, threads: int = 1):
        data = _as_u8(cloud)
        times = np.zeros(reps * threads, dtype=np.float64)
        args, _keep = self._common(info)
        r = self.lib.ref_bench_encode(*args, 1 if info.use_threads else 0, _ptr(data), data.size, reps, threads,
                                      times.ctypes.data_as(C.POINTER(C.c_double)))
        if r < 0:
            self._err("bench_encode")
        return int(r), times.reshape(threads, reps)

    def bench_decode(self, stream, out_size: int, reps: int = 10):
        st = _as_u8(stream)
        out = np.zeros(out_size, dtype=np.uint8)
        times = np.zeros(reps, dtype=np.float64)
        r = self.lib.ref_bench_decode(_ptr(st), st.size, _ptr(out), out.size, reps,
                                      times.ctypes.data_as(C.POINTER(C.c_double)))
        if r < 0:
            self._err("bench_decode")
        return out, times


def ref_available() -> bool:
    return os.path.exists(REF_SO) or os.path.exists("/root/reference/cloudini_lib/src/cloudini.cpp")
