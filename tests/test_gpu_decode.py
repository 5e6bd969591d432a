"""GPU parity: the HIP stage-1 decoder (through the C ABI) against the CPU oracle, bit for bit, including the
bytes of a point that no field covers (they must keep the caller's content)."""
import numpy as np
import pytest

import cases
from cloudini_amd import synth

pytestmark = pytest.mark.gpu

ALL = cases.encode_cases(small=False)


def check_decode(oracle, info, clouds, fill=0x5A):
    from cloudini_amd import native
    plan = native.Plan(info)
    codec = native.Codec(plan)
    step = info.point_step
    streams = [oracle.encode_stage1(info, c) for c in clouds]
    npts = [len(c) // step for c in clouds]
    out = np.full(max(1, sum(npts) * step), fill, dtype=np.uint8)
    got = codec.decode_host(streams, npts, out=out)
    for k, cloud in enumerate(clouds):
        want = oracle.decode_stage1(info, streams[k], npts[k], fill=fill)
        assert np.array_equal(got[k], want), f"cloud {k}: first diff at byte {int(np.nonzero(got[k] != want)[0][0])}"
    codec.close()


@pytest.mark.parametrize("name,info,data", ALL, ids=[c[0] for c in ALL])
def test_decode_all_schema_families(oracle, name, info, data):
    check_decode(oracle, info, [data])


def test_decode_batch_ragged(oracle):
    clouds = []
    info = None
    for k, n in enumerate([100, 70000, 0, 4096, 32768, 33000, 1]):
        info, data = synth.lidar_xyzi(n, seed=7 + k)
        clouds.append(data)
    check_decode(oracle, info, clouds)


def test_roundtrip_gpu_encode_gpu_decode(oracle):
    """encode on the GPU, decode on the GPU, compare with the oracle's decode of the oracle's stream"""
    from cloudini_amd import native
    for info, data in (synth.lidar_xyz(100000), synth.lidar_xyzi(100000), synth.velodyne_xyzir(130048)):
        codec = native.Codec(native.Plan(info))
        n = len(data) // info.point_step
        streams, _, _ = codec.encode_host([data])
        got = codec.decode_host(streams, [n])[0]
        want = oracle.decode_stage1(info, oracle.encode_stage1(info, data), n)
        assert np.array_equal(got, want)
        codec.close()


@pytest.mark.parametrize("mutation", ["truncate", "extra_chunk", "bad_size", "bad_mode", "trailing"])
def test_malformed_streams_are_rejected(oracle, mutation):
    """PointcloudDecoderRejectsMissingChunksForDeclaredPoints (test_field_encoders.cpp:771-791) and the other decoder
    hardening checks (cloudini.cpp:645-664, v5_codec.cpp:770-775, :1008-1010)."""
    from cloudini_amd import native
    info, data = synth.lidar_xyzi(40000)
    n = 40000
    s = oracle.encode_stage1(info, data).copy()
    first = int(np.frombuffer(s[:4].tobytes(), "<u4")[0])
    if mutation == "truncate":
        s = s[: 4 + first]                      # second chunk missing
    elif mutation == "extra_chunk":
        s = np.concatenate([s, s[: 4 + first]])  # more chunks than declared points
    elif mutation == "bad_size":
        s[0:4] = np.frombuffer(np.uint32(len(s) + 100).tobytes(), np.uint8)
    elif mutation == "bad_mode":
        # the section's mode byte of the last chunk: find it by re-encoding the float part only
        float_only = info.copy(fields=info.fields[:3])
        f2 = oracle.encode_stage1(float_only, data)
        reg0 = int(np.frombuffer(f2[:4].tobytes(), "<u4")[0])
        s[4 + reg0] = 9
    elif mutation == "trailing":
        s = np.concatenate([s[: 4 + first], np.zeros(3, np.uint8), s[4 + first:]])
        s[0:4] = np.frombuffer(np.uint32(first + 3).tobytes(), np.uint8)
    codec = native.Codec(native.Plan(info))
    with pytest.raises(native.CloudiniHipError) as e:
        codec.decode_host([s], [n])
    assert e.value.code == -6
    with pytest.raises(Exception):
        oracle.decode_stage1(info, s, n)
    codec.close()


def _reframe(payloads):
    out = []
    for p in payloads:
        out.append(np.frombuffer(np.uint32(len(p)).tobytes(), np.uint8))
        out.append(p)
    return np.concatenate(out)


def _split_chunks(stream):
    pos, res = 0, []
    while pos < len(stream):
        size = int(np.frombuffer(stream[pos:pos + 4].tobytes(), "<u4")[0])
        res.append(stream[pos + 4:pos + 4 + size].copy())
        pos += 4 + size
    return res


def _overlong(token, total_len):
    """The same varint value padded with zero-payload continuation groups to `total_len` bytes (decodeVarint,
    encoding_utils.hpp:98-148, accepts non-canonical encodings)."""
    t = list(token)
    t[-1] |= 0x80
    t += [0x80] * (total_len - len(t) - 1) + [0x00]
    return np.array(t, dtype=np.uint8)


@pytest.mark.parametrize("pad_to", [5, 7, 8, 9, 10])
def test_overlong_tokens_decode_like_the_reference(oracle, pad_to):
    """Tokens longer than the parallel decoder's 7-byte window (legal, non-canonical varints) must come out the
    same: those chunks fall back to the serial decoder. Shorter paddings stay on the parallel path."""
    from cloudini_amd import native
    info, data = synth.lidar_xyz(70000, seed=5)
    chunks = _split_chunks(oracle.encode_stage1(info, data))
    rs = np.random.RandomState(pad_to)
    new_chunks = []
    for ci, ch in enumerate(chunks):
        ends = np.nonzero((ch & 0x80) == 0)[0]
        starts = np.concatenate([[0], ends[:-1] + 1])
        pick = set(rs.choice(len(ends), 25, replace=False).tolist()) if ci != 1 else set()  # chunk 1 stays canonical
        parts = []
        for k, (a, b) in enumerate(zip(starts, ends)):
            tok = ch[a:b + 1]
            if k in pick and tok[0] != 0 and len(tok) < pad_to:
                parts.append(_overlong(tok, pad_to))
            else:
                parts.append(tok)
        new_chunks.append(np.concatenate(parts))
    s = _reframe(new_chunks)
    n = 70000
    want = oracle.decode_stage1(info, s, n, fill=0x11)
    codec = native.Codec(native.Plan(info))
    out = np.full(n * info.point_step, 0x11, dtype=np.uint8)
    got = codec.decode_host([s], [n], out=out)[0]
    assert np.array_equal(got, want)
    # and it still is the original cloud's decode
    assert np.array_equal(want, oracle.decode_stage1(info, oracle.encode_stage1(info, data), n, fill=0x11))
    codec.close()


def test_marker_inside_integer_token_stream_is_rejected(oracle):
    """V4 wire with an integer field: a 0x00 byte where an integer varint is expected is corrupt data
    (decodeVarint rejects value 0); the parallel path must hand the chunk to the serial checks."""
    from cloudini_amd import native
    from cloudini_amd.schema import FieldType as F
    n = 5000
    rs = np.random.RandomState(8)
    fields = [("u", 0, F.FLOAT32, 0.01), ("v", 4, F.FLOAT32, 0.01), ("a16", 8, F.INT16, None), ("e16", 10, F.UINT16, None)]
    info = cases.make_info(fields, 12, n, version=4)
    data = cases.pack(info, {"u": rs.uniform(0, 9, n).astype(np.float32), "v": rs.uniform(0, 9, n).astype(np.float32),
                             "a16": rs.randint(-300, 300, n).astype(np.int16),
                             "e16": rs.randint(0, 65536, n).astype(np.uint16)}, n)
    s = oracle.encode_stage1(info, data).copy()
    plan = native.Plan(info)
    ch = _split_chunks(s)[0]
    ends = np.nonzero((ch & 0x80) == 0)[0]
    assert len(ends) % n == 0 and len(ends) // n == 4  # two scalar floats + two integers, all varint tokens
    tpp = len(ends) // n
    # last token of point 100 belongs to the last (integer) field
    k = 100 * tpp + (tpp - 1)
    a = 0 if k == 0 else ends[k - 1] + 1
    bad = np.concatenate([ch[:a], np.zeros(1, np.uint8), ch[ends[k] + 1:]])
    s2 = _reframe([bad])
    codec = native.Codec(plan)
    with pytest.raises(native.CloudiniHipError) as e:
        codec.decode_host([s2], [n])
    assert e.value.code == -6
    with pytest.raises(Exception):
        oracle.decode_stage1(info, s2, n)
    codec.close()
