"""Host-side mirror of the reference API (include/cloudini_lib/*.hpp, cloudini_amd/csrc/host/): header bytes,
capacity bounds, error behaviour, stage 2 and the ROS message converters."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import cases
from cloudini_amd import api, synth
from cloudini_amd.schema import CompressionOption, EncodingInfo, EncodingOptions, FieldType, PointField

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _infos():
    out = []
    for name, info, data in cases.encode_cases(small=True)[::4]:
        out.append((name, info))
    info = synth.xyzi_info(1234)
    info.encoding_config = "profile=outdoor"
    out.append(("with_config", info))
    out.append(("odd_res", EncodingInfo(fields=[PointField("a", 0, FieldType.FLOAT32, 1e-6),
                                                PointField("b", 4, FieldType.FLOAT64, 0.25),
                                                PointField("c", 12, FieldType.FLOAT32, 123456.789)],
                                        width=7, height=3, point_step=16)))
    return out


INFOS = _infos()


# ---------------------------------------------------------------------------------------------------- no GPU
def test_c_abi_exports_every_declared_symbol():
    """Both shared libraries load and export what include/*.h declares (no compute without a GPU)."""
    from cloudini_amd import native
    hip = native.lib()
    host = api.lib()
    decl = re.compile(r"\b(cldn_[A-Za-z0-9_]+)\s*\(")
    hip_syms = set(decl.findall(open(os.path.join(ROOT, "include", "cloudini_hip.h")).read()))
    host_syms = set(decl.findall(open(os.path.join(ROOT, "include", "cloudini_amd_c.h")).read()))
    host_syms |= set(decl.findall(open(os.path.join(ROOT, "include", "cloudini_lib", "wasm_functions.h")).read()))
    assert len(hip_syms) >= 15 and len(host_syms) >= 18
    for s in hip_syms:
        assert hasattr(hip, s), s
    for s in host_syms:
        assert hasattr(host, s), s
    assert hip.cldn_hip_abi_version() == 1


def test_shipped_libraries_read_only_the_production_environment_variables():
    """The default build carries no development switch: libcloudini_hip.so names no CLDN_HIP_*
    variable at all (dev_env() is a constant outside -DCLDN_DEV builds), the host mirror names its three CLOUDINI_AMD_*
    settings and the transcoder's timing print; the superseded kernel generations are not in the device code."""
    libdir = os.path.join(ROOT, "cloudini_amd", "lib")
    hip = open(os.path.join(libdir, "libcloudini_hip.so"), "rb").read()
    host = open(os.path.join(libdir, "libcloudini_amd.so"), "rb").read()
    names = lambda blob: sorted({m.decode() for m in re.findall(rb"(?:CLDN|CLOUDINI)_[A-Z0-9_]{3,}(?=\x00)", blob)})
    assert [n for n in names(hip) if n.startswith(("CLDN_HIP_", "CLOUDINI_AMD_"))] == []
    assert [n for n in names(host) if n.startswith(("CLDN_", "CLOUDINI_AMD_"))] == [
        "CLDN_HOST_TIMING", "CLOUDINI_AMD_DEVICE_LZ4", "CLOUDINI_AMD_PIPELINE", "CLOUDINI_AMD_STAGE2_THREADS"]
    for dead in (b"k_compact", b"k_chunk_offsets", b"10k_lz4_emitE", b"15k_decode_pointsI"):
        assert dead not in hip, dead


@pytest.mark.parametrize("seed", cases.VERY_WIDE_SEEDS[::5])
def test_plans_of_any_size_are_accepted_without_a_gpu(oracle, seed):
    """Round 5 (no compute, runs without a GPU): cldn_hip_plan_create takes schemas beyond the launch-argument plan -- 65-200
    fields, points of 1-4 KiB -- and answers the reference's capacity bound and adaptive-field count for them."""
    from cloudini_amd import native
    info, data = cases.very_wide_schema(seed)
    plan = native.Plan(info)
    n = data.size // info.point_step
    for pts in (0, 1, n, 32768, 32769):
        assert plan.stage1_bound(pts) == oracle.stage1_bound(info, pts), (seed, pts)
    assert plan.adaptive_fields == oracle.adaptive_field_count(info)
    assert plan.point_step == info.point_step


@pytest.mark.parametrize("name,info", INFOS, ids=[i[0] for i in INFOS])
def test_header_bytes_match_reference(reflib, name, info):
    assert api.EncodeHeader(info) == reflib.header(info)
    assert api.EncodeHeader(info, binary=True) == reflib.header(info, binary=True)


@pytest.mark.parametrize("name,info", INFOS, ids=[i[0] for i in INFOS])
def test_max_compressed_size_matches_reference(reflib, name, info):
    for comp in (CompressionOption.NONE, CompressionOption.LZ4, CompressionOption.ZSTD):
        i2 = info.copy(compression_opt=comp)
        for n in (0, 1, 32768, 32769, 100000):
            for hdr in (True, False):
                assert api.MaxCompressedSize(i2, n, hdr) == reflib.max_compressed_size(i2, n, hdr), (comp, n, hdr)


def test_header_magic_and_yaml_shape():
    """test_header.cpp:107-163: magic CLOUDINI_V05 / V04, YAML text, NUL terminator."""
    info = synth.xyzi_info(10)
    h = api.EncodeHeader(info)
    assert h.startswith(b"CLOUDINI_V05\n") and h.endswith(b"\0")
    assert api.EncodeHeader(info.copy(version=4)).startswith(b"CLOUDINI_V04\n")
    y = h[13:-1].decode()
    assert y.splitlines()[0] == "version: 5" and "resolution: 0.001" in y and "resolution: null" in y
    back = api.parse_yaml_info(y, 5)
    assert [f.name for f in back.fields] == ["x", "y", "z", "intensity"] and back.point_step == 16


def test_malformed_headers_are_rejected():
    """HeaderTruncatedInput / HeaderMissingYamlTerminator (test_header.cpp:165, :243) through the reference's C ABI
    convention (0 on failure)."""
    L = api.lib()
    good = np.frombuffer(api.EncodeHeader(synth.xyzi_info(10)), dtype=np.uint8).copy()
    out = np.zeros(4096, dtype=np.uint8)

    def yaml_of(buf):
        return L.cldn_GetHeaderAsYAML(buf.ctypes.data, buf.size, out.ctypes.data)

    assert yaml_of(good) > 0
    assert yaml_of(good[:5].copy()) == 0 and b"too small" in L.cldn_LastError()
    assert yaml_of(good[:-1].copy()) == 0 and b"null terminator" in L.cldn_LastError()
    bad = good.copy()
    bad[0] = ord("X")
    assert yaml_of(bad) == 0 and b"Invalid magic header" in L.cldn_LastError()
    bad = good.copy()
    bad[10:12] = np.frombuffer(b"09", dtype=np.uint8)
    assert yaml_of(bad) == 0 and b"Unsupported encoding version" in L.cldn_LastError()


def test_encoder_argument_errors():
    info = synth.xyz_info(10)
    with pytest.raises(RuntimeError, match="point_step cannot be 0"):
        api.MaxCompressedSize(info.copy(point_step=0), 10)


# ------------------------------------------------------------------------------------------------------- GPU
GPU_CASES = cases.encode_cases(small=True)


@pytest.mark.gpu
@pytest.mark.parametrize("name,info,data", GPU_CASES, ids=[c[0] for c in GPU_CASES])
def test_full_stream_none_equals_reference(reflib, name, info, data):
    got = api.PointcloudEncoder(info).encode(data)
    want = reflib.encode(info, data)
    assert np.array_equal(got, want)
    dec, hdr = api.PointcloudDecoder().decode_stream(want, fill=0x33)
    ref_dec, _ = reflib.decode(want, len(data), fill=0x33)
    assert np.array_equal(dec, ref_dec)


@pytest.mark.gpu
@pytest.mark.parametrize("comp", [CompressionOption.LZ4, CompressionOption.ZSTD])
def test_chunk_group_pipeline_equals_reference(reflib, comp):
    """Clouds of 8 chunks and more go through the chunk-group pipeline of PointcloudEncoder::encode (stage 2 of group g
    next to the GPU's work on group g + 1, later groups with the modes of the cloud's head forced): 2, 3 and 4 groups,
    with and without adaptive fields, a ragged last chunk, the generic kernel -- the stream must still be the
    reference's (src/cloudini.cpp:572-588)."""
    before = api.stage2_threads()
    api.set_stage2_threads(8)  # the pipeline is taken from 8 stage-2 threads on (CLOUDINI_AMD_PIPELINE=n forces n groups)
    try:
        _pipeline_cases(reflib, comp)
    finally:
        api.set_stage2_threads(before)


def _pipeline_cases(reflib, comp):
    for info, data in (synth.lidar_xyzi(300_001), synth.velodyne_xyzir(600_000, seed=9), synth.lidar_xyz(1_000_000),
                       synth.depthcam_xyzrgba(640, 480), cases.mixed_schema(280_000, 5)):
        info = info.copy(compression_opt=comp, use_threads=True)
        enc = api.PointcloudEncoder(info)
        want = reflib.encode(info, data)
        for _ in range(2):  # the second call reuses the codec: the forced modes must not leak into it
            got = enc.encode(data)
            assert np.array_equal(got, want)
        # a short cloud through the same encoder's schema afterwards: probes its own modes again
        n_small = 5000
        small = data[: n_small * info.point_step]
        info_s = info.copy(width=n_small, height=1)
        assert np.array_equal(api.PointcloudEncoder(info_s).encode(small), reflib.encode(info_s, small))


@pytest.mark.gpu
@pytest.mark.parametrize("comp", [CompressionOption.LZ4, CompressionOption.ZSTD])
@pytest.mark.parametrize("threads", [False, True])
def test_stage2_streams_equal_reference(reflib, comp, threads):
    """Both sides link the same liblz4 / libzstd here, so even the compressed bytes must agree."""
    for info, data in (synth.lidar_xyzi(100000), synth.velodyne_xyzir(70000), cases.mixed_schema(40000, 5)):
        info = info.copy(compression_opt=comp, use_threads=threads)
        got = api.PointcloudEncoder(info).encode(data)
        want = reflib.encode(info, data)
        assert np.array_equal(got, want)
        dec, _ = api.PointcloudDecoder().decode_stream(got, fill=0)
        ref_dec, _ = reflib.decode(want, len(data), fill=0)
        assert np.array_equal(dec, ref_dec)


@pytest.mark.gpu
def test_roundtrip_tolerance_like_the_reference_tests():
    """FloatLossy / PCD tolerances: |decoded - original| <= resolution * 1.0001 / 2-ish (test_field_encoders.cpp:129)."""
    info, data = synth.lidar_xyz(200000)
    enc = api.PointcloudEncoder(info.copy(compression_opt=CompressionOption.ZSTD)).encode(data)
    dec, _ = api.PointcloudDecoder().decode_stream(enc)
    a = data.view(np.float32)
    b = dec.view(np.float32)
    assert np.max(np.abs(a - b)) <= 0.001 * 0.5 * 1.01


@pytest.mark.gpu
def test_encoder_errors_match_reference_messages():
    info, data = synth.lidar_xyz(100)
    with pytest.raises(RuntimeError, match="not a multiple of point_step"):
        api.PointcloudEncoder(info).encode(data[:-1])
    # view overload with a buffer below the worst case (cloudini.cpp:531-534)
    L = api.lib()
    ci, _keep = api._c_info(info)
    out = np.zeros(200, dtype=np.uint8)
    r = L.cldn_amd_encode(C.byref(ci), data.ctypes.data_as(C.POINTER(C.c_uint8)), data.size,
                          out.ctypes.data_as(C.POINTER(C.c_uint8)), out.size, 1)
    assert r == -1 and b"Output buffer too small for worst-case compressed size" in L.cldn_amd_last_error()
    # a stream that still carries its header must be refused by decode()
    enc = api.PointcloudEncoder(info).encode(data)
    with pytest.raises(RuntimeError, match="contains the header"):
        api.PointcloudDecoder().decode(info, enc)
    body = enc[len(api.EncodeHeader(info)):]
    with pytest.raises(RuntimeError, match="ended before all declared points"):
        api.PointcloudDecoder().decode(info.copy(width=32768 + 5), body)  # a second chunk is declared but missing
    with pytest.raises(RuntimeError, match="more chunks than declared points"):
        api.PointcloudDecoder().decode(info, np.concatenate([body, body]))
    with pytest.raises(RuntimeError, match="malformed stage-1 stream"):
        api.PointcloudDecoder().decode(info.copy(width=200), body)  # chunk shorter than its declared points


_cdr_pointcloud2 = synth.cdr_pointcloud2  # (the serialiser lives next to the synthetic clouds)


@pytest.mark.gpu
@pytest.mark.parametrize("comp", [CompressionOption.NONE, CompressionOption.ZSTD])
def test_ros_message_conversion_equals_reference(reflib, comp):
    """convertPointCloud2ToCompressedCloud / convertCompressedCloudToPointCloud2 (ros_msg_utils.cpp:135-213)."""
    info, data = synth.lidar_xyzi(50000)
    msg = _cdr_pointcloud2(info, data)
    got = api.ros_compress(msg, 0.001, int(comp))
    want = reflib.ros_compress(msg, 0.001, int(comp))
    assert np.array_equal(got, want)
    back = api.ros_decompress(got, msg.size + 4096)
    ref_back = reflib.ros_decompress(want, msg.size + 4096)
    assert np.array_equal(back, ref_back)
    # the reference's own C entry points over the same message
    L = api.lib()
    assert L.cldn_GetDecompressedSize(got.ctypes.data, got.size) == data.size
    out = np.zeros(data.size, dtype=np.uint8)
    assert L.cldn_DecodeCompressedMessage(got.ctypes.data, got.size, out.ctypes.data) == data.size
    assert np.max(np.abs(out.view(np.float32).reshape(-1, 4)[:, :3] - data.view(np.float32).reshape(-1, 4)[:, :3])) <= 0.00051


@pytest.mark.gpu
def test_dds_fixture_layout_roundtrip(reflib):
    """The schema of samples/dds_message.bin (test_ros_msg.cpp:91-144): FLOAT64 stamp without resolution -> Gorilla.
    Full CDR message in, CompressedPointCloud2 out, byte-identical to the reference; and back."""
    info, data = cases.ouster_like(20000)
    msg = _cdr_pointcloud2(info, data)
    got = api.ros_compress(msg, 0.001, int(CompressionOption.ZSTD))
    want = reflib.ros_compress(msg, 0.001, int(CompressionOption.ZSTD))
    assert np.array_equal(got, want)
    back = api.ros_decompress(got, msg.size + 4096)
    assert np.array_equal(back, reflib.ros_decompress(want, msg.size + 4096))


@pytest.mark.gpu
def test_encoders_on_concurrent_threads(reflib):
    """Distinct encoder / decoder instances are independent (cloudini.hpp contract): four host threads, each with its
    own schema and compression option, hammer the codec pool, the stage-2 worker pool and the per-thread buffers."""
    import threading
    jobs = []
    for k, (comp, maker) in enumerate(((CompressionOption.LZ4, lambda: synth.lidar_xyzi(90_000, seed=31)),
                                        (CompressionOption.ZSTD, lambda: synth.velodyne_xyzir(50_000, seed=32)),
                                        (CompressionOption.NONE, lambda: synth.lidar_xyz(120_000, seed=33)),
                                        (CompressionOption.ZSTD, lambda: synth.depthcam_xyzrgba(320, 240, seed=34)))):
        info, data = maker()
        info = info.copy(compression_opt=comp, use_threads=True)
        jobs.append((info, data, reflib.encode(info, data)))
    errors = []

    def worker(info, data, want):
        try:
            for _ in range(6):
                got = api.PointcloudEncoder(info).encode(data)
                if not np.array_equal(got, want):
                    errors.append("encode differs")
                dec, _i = api.PointcloudDecoder().decode_stream(want, fill=0x21)
                n = data.size
                ref_dec, _y = reflib.decode(want, n, fill=0x21)
                if not np.array_equal(dec[:n], ref_dec[:n]):
                    errors.append("decode differs")
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[:3]


@pytest.mark.gpu
def test_legacy_v3_wire_format_like_the_reference_test(reflib):
    """test_header.cpp:173-241 (DecodeV3_FromLegacyEncoder): version 3 writes the "CLOUDINI_V03" magic, encodes
    lossless FLOAT64 with the XOR codec (no Gorilla before v4), and decodes with the current library. Same struct,
    same generator; additionally the bytes must equal the reference's."""
    n = 64 * 1024 + 7
    i = np.arange(n)
    fields = [("x", 0, FieldType.FLOAT32, 0.001), ("y", 4, FieldType.FLOAT32, 0.001), ("z", 8, FieldType.FLOAT32, 0.001),
              ("stamp", 16, FieldType.FLOAT64, None)]
    info = cases.make_info(fields, 24, n, version=3, comp=CompressionOption.ZSTD)
    cols = {"x": (np.float32(0.01) * i.astype(np.float32)), "y": (np.float32(-0.02) * i.astype(np.float32) + np.float32(0.5)),
            "z": (np.float32(0.001) * i.astype(np.float32) - np.float32(0.25)), "stamp": 1700000000.0 + 0.000001 * i}
    data = cases.pack(info, cols, n)
    got = api.PointcloudEncoder(info).encode(data)
    assert got[:12].tobytes() == b"CLOUDINI_V03"
    assert np.array_equal(got, reflib.encode(info, data))
    dec, dinfo = api.PointcloudDecoder().decode_stream(got, fill=0xA5)
    assert dinfo.version == 3
    a = data.reshape(n, 24)
    b = dec[: n * 24].reshape(n, 24)
    for k in range(3):
        x = a[:, 4 * k:4 * k + 4].copy().view(np.float32).reshape(-1)
        y = b[:, 4 * k:4 * k + 4].copy().view(np.float32).reshape(-1)
        assert np.all(np.abs(x - y) <= 0.001 * 1.01)
    assert np.array_equal(a[:, 16:24], b[:, 16:24])  # lossless stamp, bit exact
    # v4 of the same cloud uses Gorilla for the stamp: a different payload
    info4 = info.copy(version=4)
    got4 = api.PointcloudEncoder(info4).encode(data)
    assert got4[:12].tobytes() == b"CLOUDINI_V04" and not np.array_equal(got4[13:], got[13:])
    assert np.array_equal(got4, reflib.encode(info4, data))


def _stage2(comp, payload: bytes) -> bytes:
    """LZ4 block / ZSTD frame of `payload` through the system libraries (what CompressChunk calls, codec_common.cpp:220-258)."""
    if comp == CompressionOption.NONE:
        return payload
    if comp == CompressionOption.LZ4:
        lz4 = C.CDLL("/usr/lib/x86_64-linux-gnu/liblz4.so.1")
        cap = lz4.LZ4_compressBound(len(payload))
        out = C.create_string_buffer(cap)
        n = lz4.LZ4_compress_default(payload, out, len(payload), cap)
        assert n > 0
        return out.raw[:n]
    zstd = C.CDLL("/usr/lib/x86_64-linux-gnu/libzstd.so.1")
    zstd.ZSTD_compressBound.restype = C.c_size_t
    zstd.ZSTD_compressBound.argtypes = [C.c_size_t]
    zstd.ZSTD_compress.restype = C.c_size_t
    zstd.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int]
    cap = zstd.ZSTD_compressBound(len(payload))
    out = C.create_string_buffer(cap)
    n = zstd.ZSTD_compress(out, cap, payload, len(payload), 1)
    return out.raw[:n]


@pytest.mark.gpu
@pytest.mark.parametrize("comp", [CompressionOption.NONE, CompressionOption.LZ4, CompressionOption.ZSTD])
def test_wire_version_2_streams_decode_like_the_reference(reflib, comp):
    """Streams of wire version 2 (no chunks: the whole payload is one run of points, decoded until it is empty) are still
    read by the reference (src/cloudini.cpp:665-667, src/v4_codec.cpp:108-115), and by the host mirror."""
    n = 20000
    info3, data = synth.lidar_xyzi(n, seed=4)
    info3 = info3.copy(version=3, compression_opt=CompressionOption.NONE)
    framed = reflib.encode_stage1(info3, data)               # one chunk: [u32 size][payload]
    size = int.from_bytes(framed[:4].tobytes(), "little")
    assert size + 4 == framed.size
    payload = framed[4:].tobytes()
    for declared, keep in ((n, n), (n + 500, n), (n - 1, n)):
        info2 = info3.copy(version=2, compression_opt=comp, width=declared, height=1)
        stream = np.frombuffer(reflib.header(info2) + _stage2(comp, payload), dtype=np.uint8)
        assert stream[:12].tobytes() == b"CLOUDINI_V02"
        out_size = declared * info2.point_step
        if declared < keep:   # one point more than the output holds
            with pytest.raises(RuntimeError):
                reflib.decode(stream, out_size, fill=0x5A)
            with pytest.raises(RuntimeError):
                api.PointcloudDecoder().decode_stream(stream, fill=0x5A)
            continue
        want, _yaml = reflib.decode(stream, out_size, fill=0x5A)
        got, got_info = api.PointcloudDecoder().decode_stream(stream, fill=0x5A)
        assert int(got_info.version) == 2
        assert got.size == want.size and np.array_equal(got, want), (comp, declared)
    # a payload cut inside a point is "Truncated encoded data"
    info2 = info3.copy(version=2, compression_opt=comp, width=n, height=1)
    stream = np.frombuffer(reflib.header(info2) + _stage2(comp, payload[:-1]), dtype=np.uint8)
    with pytest.raises(RuntimeError):
        reflib.decode(stream, n * info2.point_step)
    with pytest.raises(RuntimeError):
        api.PointcloudDecoder().decode_stream(stream)
