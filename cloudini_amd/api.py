"""Python face of the C++ host API (include/cloudini_lib/cloudini.hpp) through include/cloudini_amd_c.h.

Names follow the reference: PointcloudEncoder / PointcloudDecoder / EncodeHeader / DecodeHeader /
MaxCompressedSize and the ROS message converters (cloudini_lib/include/cloudini_lib/cloudini.hpp:126-244,
ros_msg_utils.hpp:175-221). Errors surface as RuntimeError carrying the C++ exception text.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Tuple

import numpy as np

from .schema import EncodingInfo, EncodingOptions, CompressionOption, FieldType, PointField

HERE = os.path.dirname(os.path.abspath(__file__))
HOST_SO = os.path.join(HERE, "lib", "libcloudini_amd.so")


class _Field(C.Structure):
    _fields_ = [("name", C.c_char_p), ("offset", C.c_uint32), ("type", C.c_uint8), ("has_resolution", C.c_uint8),
                ("reserved", C.c_uint8 * 2), ("resolution", C.c_float)]


class _Info(C.Structure):
    _fields_ = [("fields", C.POINTER(_Field)), ("n_fields", C.c_uint32), ("width", C.c_uint32),
                ("height", C.c_uint32), ("point_step", C.c_uint32), ("encoding_opt", C.c_uint8),
                ("compression_opt", C.c_uint8), ("version", C.c_uint8), ("use_threads", C.c_uint8)]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    from . import native
    native.lib()  # loads torch's HIP runtime first (if torch is around) and libcloudini_hip.so
    if not os.path.exists(HOST_SO):
        raise ImportError(f"{HOST_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    L = C.CDLL(HOST_SO)
    u8p = C.POINTER(C.c_uint8)
    L.cldn_amd_last_error.restype = C.c_char_p
    L.cldn_LastError.restype = C.c_char_p
    L.cldn_amd_max_compressed_size.restype = C.c_int64
    L.cldn_amd_max_compressed_size.argtypes = [C.POINTER(_Info), C.c_uint64, C.c_int]
    L.cldn_amd_encode_header.restype = C.c_int64
    L.cldn_amd_encode_header.argtypes = [C.POINTER(_Info), C.c_int, u8p, C.c_uint64]
    L.cldn_amd_encode.restype = C.c_int64
    L.cldn_amd_encode.argtypes = [C.POINTER(_Info), u8p, C.c_uint64, u8p, C.c_uint64, C.c_int]
    L.cldn_amd_decode.restype = C.c_int64
    L.cldn_amd_decode.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64, C.c_char_p, C.c_uint64, u8p]
    L.cldn_amd_decode_noheader.restype = C.c_int64
    L.cldn_amd_decode_noheader.argtypes = [C.POINTER(_Info), u8p, C.c_uint64, u8p, C.c_uint64]
    L.cldn_amd_decode_noheader_zeroed.restype = C.c_int64
    L.cldn_amd_decode_noheader_zeroed.argtypes = [C.POINTER(_Info), u8p, C.c_uint64, u8p, C.c_uint64]
    L.cldn_amd_ros_compress.restype = C.c_int64
    L.cldn_amd_ros_compress.argtypes = [u8p, C.c_uint64, C.c_float, C.c_uint8, u8p, C.c_uint64]
    L.cldn_amd_ros_decompress.restype = C.c_int64
    L.cldn_amd_ros_decompress.argtypes = [u8p, C.c_uint64, u8p, C.c_uint64]
    L.cldn_amd_viz_preprocess.restype = C.c_int64
    L.cldn_amd_viz_preprocess.argtypes = [C.POINTER(_Info), u8p, C.c_uint64, u8p, C.c_uint64, C.POINTER(C.c_float),
                                          C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.cldn_amd_transcode_directory_on.restype = C.c_int64
    L.cldn_amd_transcode_directory_on.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_uint8, C.c_int, C.c_uint32,
                                                  C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_double)]
    L.cldn_amd_decode_directory_on.restype = C.c_int64
    L.cldn_amd_decode_directory_on.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32,
                                               C.POINTER(C.c_double)]
    L.cldn_amd_transcode_directory.restype = C.c_int64
    L.cldn_amd_transcode_directory.argtypes = [C.c_char_p, C.c_char_p, C.c_float, C.c_uint8, C.c_int, C.c_uint32, C.POINTER(C.c_double)]
    L.cldn_amd_decode_directory.restype = C.c_int64
    L.cldn_amd_decode_directory.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.POINTER(C.c_double)]
    L.cldn_amd_stage2_threads.restype = C.c_uint32
    L.cldn_amd_set_stage2_threads.restype = C.c_uint32
    L.cldn_amd_set_stage2_threads.argtypes = [C.c_uint32]
    for name in ("cldn_GetHeaderAsYAML", "cldn_GetHeaderAsYAMLFromDDS", "cldn_ConvertCompressedMsgToPointCloud2Msg",
                 "cldn_DecodeCompressedData", "cldn_DecodeCompressedMessage"):
        getattr(L, name).restype = C.c_uint32
        getattr(L, name).argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.cldn_ComputeCompressedSize.restype = C.c_uint32
    L.cldn_ComputeCompressedSize.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
    L.cldn_GetDecompressedSize.restype = C.c_uint32
    L.cldn_GetDecompressedSize.argtypes = [C.c_void_p, C.c_uint32]
    L.cldn_EncodePointcloudMessage.restype = C.c_uint32
    L.cldn_EncodePointcloudMessage.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_void_p]
    L.cldn_EncodePointcloudData.restype = C.c_uint32
    L.cldn_EncodePointcloudData.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p]
    _lib = L
    return L


def _c_info(info) -> Tuple[_Info, object]:
    arr = (_Field * max(1, len(info.fields)))()
    keep = []
    for i, f in enumerate(info.fields):
        nm = f.name.encode()
        keep.append(nm)
        arr[i].name = nm
        arr[i].offset = int(f.offset)
        arr[i].type = int(f.type)
        arr[i].has_resolution = 0 if f.resolution is None else 1
        arr[i].resolution = 0.0 if f.resolution is None else float(f.resolution)
    ci = _Info(arr, len(info.fields), int(info.width), int(info.height), int(info.point_step), int(info.encoding_opt),
               int(info.compression_opt), int(info.version), 1 if info.use_threads else 0)
    return ci, (arr, keep)


def _u8(buf) -> np.ndarray:
    if isinstance(buf, np.ndarray):
        return np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(buf), dtype=np.uint8)


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def _check(r: int) -> int:
    if r < 0:
        raise RuntimeError(lib().cldn_amd_last_error().decode(errors="replace"))
    return r


def stage2_threads() -> int:
    """Threads one encode()/decode() call may use for LZ4/ZSTD, the caller included (bounded pool)."""
    return int(lib().cldn_amd_stage2_threads())


def set_stage2_threads(n: int) -> int:
    return int(lib().cldn_amd_set_stage2_threads(int(n)))


def device_lz4() -> bool:
    return bool(lib().cldn_amd_device_lz4())


def set_device_lz4(on) -> int:
    """LZ4 streams with stage 2 on the GPU (valid LZ4 blocks, not lz4's own bytes): False / 0 off, True / 1 on, 2 = the FAST
    parameters (4 KiB windows). Returns the level in effect."""
    return int(lib().cldn_amd_set_device_lz4(int(on)))


def _device_list(devices):
    if not devices:
        return None, 0
    arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    return arr, len(devices)


def transcode_directory(in_dir: str, out_dir: str, resolution: float = 0.001, compression_opt: int = 2,
                        viz_lossy: bool = False, batch_messages: int = 64, devices=None) -> dict:
    """Batch transcoder (include/cloudini_amd/batch_transcoder.hpp): every CDR PointCloud2 file of in_dir ->
    CompressedPointCloud2 file of the same name in out_dir. `devices`: GPUs to spread the batches over (one GPU stage per
    entry; None = the current device). Returns the statistics."""
    st = (C.c_double * 8)()
    dv, nd = _device_list(devices)
    _check(lib().cldn_amd_transcode_directory_on(in_dir.encode(), out_dir.encode(), resolution, compression_opt,
                                                 1 if viz_lossy else 0, batch_messages, dv, nd, st))
    keys = ("messages", "points", "input_bytes", "output_bytes", "gpu_batches", "seconds_total", "seconds_gpu", "seconds_stage2")
    return dict(zip(keys, [float(x) for x in st]))


def decode_directory(in_dir: str, out_dir: str, batch_messages: int = 64, devices=None) -> dict:
    """The way back: every CDR CompressedPointCloud2 file of in_dir -> PointCloud2 file of the same name in out_dir
    (batched GPU decode). Returns the statistics."""
    st = (C.c_double * 8)()
    dv, nd = _device_list(devices)
    _check(lib().cldn_amd_decode_directory_on(in_dir.encode(), out_dir.encode(), batch_messages, dv, nd, st))
    keys = ("messages", "points", "input_bytes", "output_bytes", "gpu_batches", "seconds_total", "seconds_gpu", "seconds_stage2")
    return dict(zip(keys, [float(x) for x in st]))


def MaxCompressedSize(info, points_count: int, include_header: bool = True) -> int:
    ci, _keep = _c_info(info)
    return _check(lib().cldn_amd_max_compressed_size(C.byref(ci), points_count, 1 if include_header else 0))


def EncodeHeader(info, binary: bool = False) -> bytes:
    ci, _keep = _c_info(info)
    out = np.empty(1 << 16, dtype=np.uint8)
    n = _check(lib().cldn_amd_encode_header(C.byref(ci), 1 if binary else 0, _ptr(out), out.size))
    return out[:n].tobytes()


def parse_yaml_info(yaml: str, version: int) -> EncodingInfo:
    """EncodingInfoToYAML text -> EncodingInfo (Python side, for the tests)."""
    info = EncodingInfo(fields=[], version=version)
    cur = None
    for line in yaml.splitlines():
        s = line.strip()
        if not s or s == "fields:":
            continue
        if s.startswith("- "):
            cur = PointField("")
            info.fields.append(cur)
            s = s[2:]
        k, _, v = s.partition(":")
        k, v = k.strip(), v.strip()
        if cur is None:
            if k in ("width", "height", "point_step"):
                setattr(info, k, int(v))
            elif k == "encoding_opt":
                info.encoding_opt = EncodingOptions[v]
            elif k == "compression_opt":
                info.compression_opt = CompressionOption[v]
            elif k == "encoding_config":
                info.encoding_config = v
        else:
            if k == "name":
                cur.name = v
            elif k == "offset":
                cur.offset = int(v)
            elif k == "type":
                cur.type = FieldType[v]
            elif k == "resolution":
                cur.resolution = None if v == "null" else float(np.float32(v))
    return info


class PointcloudEncoder:
    def __init__(self, info: EncodingInfo):
        self.info = info

    def getHeader(self) -> bytes:
        return EncodeHeader(self.info)

    def encode(self, cloud, write_header: bool = True) -> np.ndarray:
        data = _u8(cloud)
        step = int(self.info.point_step)
        points = data.size // step if step else 0
        ci, _keep = _c_info(self.info)
        cap = MaxCompressedSize(self.info, points, True)
        out = np.empty(cap, dtype=np.uint8)
        n = _check(lib().cldn_amd_encode(C.byref(ci), _ptr(data) if data.size else None, data.size, _ptr(out), cap,
                                         1 if write_header else 0))
        return out[:n].copy()


class PointcloudDecoder:
    def decode_stream(self, stream, fill: int = 0):
        """Full stream (header + chunks) -> (decoded bytes, EncodingInfo of the header)."""
        st = _u8(stream)
        yaml = C.create_string_buffer(1 << 16)
        ver = C.c_uint8(0)
        # first pass with a zero-size buffer is not possible (decode needs the output); size it from the header text
        hdr_end = st.tobytes().find(b"\0")
        info = None
        if st[:10].tobytes() == b"CLOUDINI_V" and hdr_end > 0 and st[12] == 0x0A:
            info = parse_yaml_info(st[13:hdr_end].tobytes().decode(), int(st[10:12].tobytes()))
            size = info.width * info.height * info.point_step
        else:
            size = 1 << 26
        out = np.full(max(1, size), fill, dtype=np.uint8)
        n = _check(lib().cldn_amd_decode(_ptr(st), st.size, _ptr(out), out.size, yaml, len(yaml), C.byref(ver)))
        return out[:n], parse_yaml_info(yaml.value.decode(), ver.value)

    def decode(self, info: EncodingInfo, data, fill: int = 0) -> np.ndarray:
        """PointcloudDecoder::decode(info, compressed_data (no header), output)."""
        st = _u8(data)
        size = int(info.width) * int(info.height) * int(info.point_step)
        out = np.full(max(1, size), fill, dtype=np.uint8)
        ci, _keep = _c_info(info)
        n = _check(lib().cldn_amd_decode_noheader(C.byref(ci), _ptr(st) if st.size else None, st.size, _ptr(out), size))
        return out[:n]


def ros_compress(dds, resolution: float, compression_opt: int) -> np.ndarray:
    msg = _u8(dds)
    out = np.empty(msg.size * 3 + (1 << 20), dtype=np.uint8)
    n = _check(lib().cldn_amd_ros_compress(_ptr(msg), msg.size, resolution, compression_opt, _ptr(out), out.size))
    return out[:n].copy()


def ros_decompress(dds, capacity: int) -> np.ndarray:
    msg = _u8(dds)
    out = np.empty(capacity, dtype=np.uint8)
    n = _check(lib().cldn_amd_ros_decompress(_ptr(msg), msg.size, _ptr(out), out.size))
    return out[:n].copy()


def applyVizLossyPreprocessing(info, cloud):
    """cloudini_ros::applyVizLossyPreprocessing on a bare point buffer -> (surviving bytes, per-field resolution
    afterwards (None = none), width, height)."""
    data = _u8(cloud)
    ci, _keep = _c_info(info)
    out = np.empty(max(1, data.size), dtype=np.uint8)
    res = (C.c_float * max(1, len(info.fields)))()
    w, h = C.c_uint32(0), C.c_uint32(0)
    n = _check(lib().cldn_amd_viz_preprocess(C.byref(ci), _ptr(data) if data.size else None, data.size, _ptr(out),
                                             out.size, res, C.byref(w), C.byref(h)))
    return out[:n].copy(), [None if x != x else float(x) for x in list(res)[: len(info.fields)]], int(w.value), int(h.value)
