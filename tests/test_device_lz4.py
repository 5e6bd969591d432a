"""Stage 2 on the device: LZ4 blocks written by cloudini_amd/csrc/lz4_kernels.hip.

The blocks need not be the bytes lz4's own compressor writes; they must be valid LZ4 blocks that the reference's
DecompressChunk (LZ4_decompress_safe, src/codec_common.cpp:260-299) turns back into the exact stage-1 payload. The device
algorithm is deterministic and restated serially in oracle/lz4_model.c, so:
  CPU  the model's blocks decode to their input through the system's liblz4 (edge sizes, incompressible, repetitive,
       matches across every boundary rule);
  GPU  the kernels write the model's bytes, chunk by chunk; full streams decode through the compiled reference itself to
       what the reference's own LZ4 stream decodes to."""
import ctypes as C

import numpy as np
import pytest

from cloudini_amd import synth
from cloudini_amd.schema import CompressionOption

LZ4_SO = "/usr/lib/x86_64-linux-gnu/liblz4.so.1"


def _lz4_decompress(block: np.ndarray, size: int) -> bytes:
    lz4 = C.CDLL(LZ4_SO)
    out = C.create_string_buffer(max(1, size))
    n = lz4.LZ4_decompress_safe(block.ctypes.data_as(C.c_char_p), out, int(block.size), int(size))
    assert n == size, f"LZ4_decompress_safe returned {n}, expected {size}"
    return out.raw[:size]


def _payload_kinds(rs, n):
    yield "random", rs.randint(0, 256, n).astype(np.uint8).tobytes()
    yield "zeros", bytes(n)
    yield "four_symbols", rs.randint(0, 4, n).astype(np.uint8).tobytes()
    yield "period7", (bytes(range(7)) * (n // 7 + 1))[:n]
    yield "sparse_matches", bytes(b if (i // 5) % 2 else (i * 37) & 0xff for i, b in enumerate(rs.randint(0, 3, n).astype(np.uint8)))


@pytest.mark.parametrize("n", list(range(0, 24)) + [63, 64, 65, 255, 256, 270, 1000, 4096, 8191, 8192, 8193, 8204, 8208, 16383, 16384, 16385,
                                                    32768 + 11, 70001])
def test_model_blocks_are_valid_lz4(oracle, n):
    rs = np.random.RandomState(n)
    for kind, payload in _payload_kinds(rs, n):
        for sub, hb, mm in ((8192, 11, 1024), (4096, 10, 512), (8192, 12, 1024), (16384, 12, 2048), (64, 4, 3), (100, 6, 1 << 20), (8192, 12, 5)):
            block = oracle.lz4_model(payload, sub, hb, mm)
            assert _lz4_decompress(block, n) == payload, (kind, sub, hb, mm)
            assert block.size <= n + n // 255 + 16


def test_model_compresses_what_is_compressible(oracle):
    rs = np.random.RandomState(3)
    assert oracle.lz4_model(bytes(100000)).size < 1600                      # one long match per 8 KiB sub-range
    assert oracle.lz4_model((bytes(range(7)) * 20000)[:100000]).size < 1700
    noise = rs.randint(0, 256, 100000).astype(np.uint8).tobytes()
    assert oracle.lz4_model(noise).size <= 100000 + 100000 // 255 + 16
    info, data = synth.depthcam_xyzrgba(320, 240, seed=1)
    stream = oracle.encode_stage1(info, data)
    size = int.from_bytes(stream[:4].tobytes(), "little")
    payload = stream[4:4 + size].tobytes()
    assert oracle.lz4_model(payload).size < 0.9 * size                      # the rgba DeltaVarint section repeats


def _chunks(stream: np.ndarray):
    out, o = [], 0
    while o < stream.size:
        size = int.from_bytes(stream[o:o + 4].tobytes(), "little")
        out.append(stream[o + 4:o + 4 + size])
        o += 4 + size
    assert o == stream.size
    return out


GPU_CASES = {
    "xyzi_70k": lambda: synth.lidar_xyzi(70000, seed=7),
    "xyzi_1": lambda: synth.lidar_xyzi(1, seed=7),
    "xyzi_3": lambda: synth.lidar_xyzi(3, seed=7),
    "depth_rgba": lambda: synth.depthcam_xyzrgba(320, 240, seed=2),
    "velodyne": lambda: synth.velodyne_xyzir(130048, seed=3),
    "xyz_200k": lambda: synth.lidar_xyz(200000, seed=5),
}


_MODEL_PARAMS = {1: (8192, 11, 1024), 2: (4096, 10, 512)}   # CLDN_HIP_STAGE2_LZ4, CLDN_HIP_STAGE2_LZ4_FAST (stage1_launch.h)


@pytest.mark.gpu
@pytest.mark.parametrize("stage2", [1, 2])
@pytest.mark.parametrize("name", sorted(GPU_CASES))
def test_device_blocks_equal_the_model_and_decode(oracle, name, stage2):
    from cloudini_amd import native
    info, data = GPU_CASES[name]()
    codec = native.Codec(native.Plan(info))
    want_s1 = oracle.encode_stage1(info, data)
    codec.set_stage2(stage2)
    streams, chunk_sizes, _modes = codec.encode_host([data, data[: (data.size // info.point_step // 2) * info.point_step]])
    codec.set_stage2(0)
    plain, plain_sizes, _ = codec.encode_host([data])
    assert np.array_equal(plain[0], want_s1)                                 # switching back gives stage-1 streams again
    payloads = _chunks(want_s1)
    blocks = _chunks(streams[0])
    assert len(blocks) == len(payloads) and [b.size for b in blocks] == [int(x) for x in chunk_sizes[: len(blocks)]]
    for k, (block, payload) in enumerate(zip(blocks, payloads)):
        assert _lz4_decompress(np.ascontiguousarray(block), payload.size) == payload.tobytes(), (name, k)
        model = oracle.lz4_model(payload, *_MODEL_PARAMS[stage2])
        assert block.size == model.size and np.array_equal(block, model), (name, k)
    codec.close()


@pytest.mark.gpu
def test_empty_and_ragged_batches_with_device_lz4(oracle):
    from cloudini_amd import native
    info, _ = synth.lidar_xyzi(10)
    codec = native.Codec(native.Plan(info))
    codec.set_stage2(1)
    clouds = [synth.lidar_xyzi(n, seed=30 + k)[1] for k, n in enumerate([0, 5, 40000, 0, 32768, 33000])]
    streams, chunk_sizes, _ = codec.encode_host(clouds)
    for k, cloud in enumerate(clouds):
        want = _chunks(oracle.encode_stage1(info, cloud))
        got = _chunks(streams[k])
        assert len(got) == len(want)
        for block, payload in zip(got, want):
            assert np.array_equal(block, oracle.lz4_model(payload))
    streams, _, _ = codec.encode_host([clouds[0]])
    assert streams[0].size == 0
    codec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["xyzi_70k", "depth_rgba", "velodyne"])
def test_host_mirror_lz4_stream_decodes_through_the_reference(reflib, name):
    """PointcloudEncoder::encode with compression_opt LZ4 and stage 2 on the device: a full stream (header + chunks) that
    the compiled reference decodes to exactly what it decodes its own LZ4 stream to."""
    from cloudini_amd import api
    info, data = GPU_CASES[name]()
    info = info.copy(compression_opt=CompressionOption.LZ4)
    ref_stream = reflib.encode(info, data)
    want, _ = reflib.decode(ref_stream, data.size, fill=0x11)
    assert not api.device_lz4()
    host = api.PointcloudEncoder(info).encode(data)
    assert np.array_equal(host, ref_stream)                                  # host stage 2: the reference's bytes
    hdr = reflib.header(info)
    sizes = []
    for level in (1, 2):                                                     # CLDN_HIP_STAGE2_LZ4, ..._FAST (4 KiB windows)
        assert api.set_device_lz4(level) == level
        try:
            dev = api.PointcloudEncoder(info).encode(data)
        finally:
            assert api.set_device_lz4(False) == 0
        assert dev[: len(hdr)].tobytes() == hdr
        got, _ = reflib.decode(dev, data.size, fill=0x11)
        assert np.array_equal(got, want)
        ours, _ = api.PointcloudDecoder().decode_stream(dev, fill=0x11)      # and through the host mirror's decoder
        assert np.array_equal(ours, want)
        sizes.append(dev.size)
    assert sizes[0] <= sizes[1] or sizes[1] > 0.99 * sizes[0]                # (smaller windows: the same bytes or a few more)
