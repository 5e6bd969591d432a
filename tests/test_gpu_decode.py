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


@pytest.mark.parametrize("length", [2, 3, 5])
@pytest.mark.parametrize("gen", ["xyz", "xyzi", "velodyne"])
def test_overlong_zero_in_a_float_lane_is_rejected(oracle, gen, length):
    """A varint token of two or more bytes whose value bits are all 0 (0x80 0x00, ...) is not the NaN marker -- that is the
    single byte 0x00 in front of decodeVarint (src/field_decoder.cpp:58-62) -- and decodeVarint rejects every zero it is
    handed (encoding_utils.hpp:139-141). The wave kernel hands such a chunk back; the tile kernels behind it used to take
    the zero for a marker and accept the stream (found by the fuzz campaign's second seed range, round 5)."""
    from cloudini_amd import native
    info, data = {"xyz": synth.lidar_xyz, "xyzi": synth.lidar_xyzi, "velodyne": synth.velodyne_xyzir}[gen](40000, seed=9)
    n = 40000
    chunks = _split_chunks(oracle.encode_stage1(info, data))
    for ci in (0, len(chunks) - 1):                      # a full chunk, the ragged last one
        ch = chunks[ci]
        n_reg = (32768 if ci == 0 else n - 32768) * (3 if gen == "xyz" else 4)   # tokens of the regular stream (V5: sections follow)
        ends = np.nonzero((ch & 0x80) == 0)[0][:n_reg]
        starts = np.concatenate([[0], ends[:-1] + 1])
        cand = np.nonzero((ends - starts + 1 == 1) & (ch[starts] != 0))[0]
        k = int(cand[len(cand) // 2])                    # a one-byte token in the middle of the stream
        bad = [c.copy() for c in chunks]
        bad[ci] = np.concatenate([ch[:starts[k]], np.array([0x80] * (length - 1) + [0x00], dtype=np.uint8), ch[ends[k] + 1:]])
        s = _reframe(bad)
        with pytest.raises(Exception):
            oracle.decode_stage1(info, s, n, fill=0x11)
        codec = native.Codec(native.Plan(info))
        with pytest.raises(native.CloudiniHipError) as e:
            codec.decode_host([s], [n], out=np.full(n * info.point_step, 0x11, dtype=np.uint8))
        assert e.value.code == -6
        codec.close()


@pytest.mark.parametrize("layout", ["scalar_lossy", "varint_and_raw"])
def test_overlong_zero_in_scalar_and_mixed_layouts_is_rejected(oracle, layout):
    """The same rule on the other decode routes: scalar lossy floats and 64-bit integers (the stream kernel in MSB mode, behind
    it the 64-bit tile kernel), and a varint next to a raw field (byte automaton + bitmap mode, behind it the tile kernel
    with the end bitmap)."""
    from cloudini_amd import native
    from cloudini_amd.schema import FieldType as F
    n = 9000
    rs = np.random.RandomState(4)
    if layout == "scalar_lossy":
        fields = [("t", 0, F.FLOAT64, 1e-6), ("a", 8, F.FLOAT32, 0.01), ("k", 12, F.INT64, None)]
        step = 20
        cols = {"t": np.cumsum(rs.uniform(0, 1e-3, n)) + 1.7e9, "a": np.cumsum(rs.normal(0, 0.05, n)).astype(np.float32),
                "k": np.cumsum(rs.randint(-50, 50, n)).astype(np.int64)}
        n_ops, victim = 3, 1
    else:
        fields = [("x", 0, F.FLOAT32, 0.001), ("rgb", 4, F.FLOAT32, None)]
        step = 8
        cols = {"x": np.cumsum(rs.normal(0, 0.01, n)).astype(np.float32), "rgb": rs.randint(0, 1 << 24, n).astype(np.uint32).view(np.float32)}
        n_ops, victim = 2, 0
    info = cases.make_info(fields, step, n, version=4)
    data = cases.pack(info, cols, n)
    s0 = oracle.encode_stage1(info, data)
    assert np.array_equal(oracle.decode_stage1(info, s0, n, fill=0x11), oracle.decode_stage1(info, s0, n, fill=0x11))
    ch = _split_chunks(s0)[0]
    # walk the points: token k of a point is a varint unless it is the raw field of the second layout (4 bytes)
    pos, hit = 0, None
    for pt in range(n):
        for op in range(n_ops):
            if layout == "varint_and_raw" and op == 1:
                pos += 4
                continue
            start = pos
            while ch[pos] & 0x80:
                pos += 1
            pos += 1
            if hit is None and pt > n // 2 and op == victim and pos - start == 1 and ch[start] != 0:
                hit = start
        if hit is not None:
            break
    assert hit is not None
    for length in (2, 4):
        bad = np.concatenate([ch[:hit], np.array([0x80] * (length - 1) + [0x00], dtype=np.uint8), ch[hit + 1:]])
        s = _reframe([bad])
        with pytest.raises(Exception):
            oracle.decode_stage1(info, s, n, fill=0x11)
        codec = native.Codec(native.Plan(info))
        with pytest.raises(native.CloudiniHipError) as e:
            codec.decode_host([s], [n], out=np.full(n * step, 0x11, dtype=np.uint8))
        assert e.value.code == -6
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


def _stats_after_decode(oracle, info, data):
    from cloudini_amd import native
    codec = native.Codec(native.Plan(info))
    step = info.point_step
    n = data.size // step
    stream, modes = oracle.encode_stage1(info, data, return_modes=True)
    out = np.full(max(1, n * step), 0x3C, dtype=np.uint8)
    got = codec.decode_host([stream], [n], out=out)[0]
    want = oracle.decode_stage1(info, stream, n, fill=0x3C)
    assert np.array_equal(got, want)
    stats = codec.decode_stats()
    codec.close()
    return stats, modes.tolist(), (n + 32767) // 32768


def test_parallel_kernels_take_the_common_schemas(oracle):
    """The parallel decoders must really run (decode_stats): regular stream AND sections of every chunk for the
    BASELINE schemas, whose sections cover Palette (C2), DeltaVarint (C3) and DeltaRle (C4)."""
    for (info, data), want_mode in ((synth.lidar_xyzi(100_000, seed=3), 1), (synth.depthcam_xyzrgba(320, 240), 0),
                                    (synth.velodyne_xyzir(130048, seed=4), 3)):
        stats, modes, n_chunks = _stats_after_decode(oracle, info, data)
        assert want_mode in modes
        assert stats == (n_chunks, n_chunks, 0, 0), (stats, modes)
    stats, _modes, n_chunks = _stats_after_decode(oracle, *synth.lidar_xyz(70_000))
    assert stats == (n_chunks, 0, 0, 0)


@pytest.mark.parametrize("kind", ["constant", "long_and_short", "alternating", "delta_runs_i64", "drle_then_noise"])
def test_parallel_sections_run_modes(oracle, kind):
    """Integer-only clouds (no regular tokens at all) in the run-length modes: sections by the parallel kernel,
    except where a token is longer than its 7-byte window (64-bit deltas) -- those chunks go serial, and still match."""
    info, data = cases.rle_stress(kind)
    stats, modes, n_chunks = _stats_after_decode(oracle, info, data)
    assert stats[0] == n_chunks and stats[2] == 0          # an empty regular stream is trivially parallel
    assert stats[1] + stats[3] == n_chunks
    if kind in ("constant", "long_and_short", "alternating"):
        assert stats[1] == n_chunks, (stats, modes)


@pytest.mark.parametrize("kind", ["grows_u16", "all_distinct_u32", "single_value", "two_values_i16", "u5000_u32"])
def test_parallel_sections_palette_and_delta(oracle, kind):
    info, data = cases.palette_stress(kind)
    stats, modes, n_chunks = _stats_after_decode(oracle, info, data)
    assert stats[0] == n_chunks and stats[1] == n_chunks and stats[2] == 0 and stats[3] == 0, (stats, modes)


@pytest.mark.parametrize("what", ["palette_index", "run_too_long", "runs_short"])
def test_corrupt_sections_are_rejected(oracle, what):
    """Damage inside a section: the parallel kernel must step aside and the serial checks must fire
    (decodeV5AdaptiveIntSection, src/v5_codec.cpp:764-879)."""
    from cloudini_amd import native
    if what == "palette_index":
        rs = np.random.RandomState(1)
        info, data = cases.int_only((rs.randint(0, 3, 5000) * 7).astype(np.uint16), cases.F.UINT16)
        s = oracle.encode_stage1(info, data).copy()
        assert s[4] == 1 and s[5] == 3                                # Palette, 3 entries, 2 bits per index
        s[4 + 1 + 2 + 6 + 10] = 0xFF                                  # index value 3 >= count
    else:
        info, data = cases.int_only(np.repeat(np.arange(50, dtype=np.uint16), 100), cases.F.UINT16)
        s = oracle.encode_stage1(info, data).copy()
        assert s[4] in (2, 3)
        if what == "run_too_long":
            pos = 4 + 1 + 4                                            # first run record
            pos += 2 if s[4] == 2 else 1                               # raw value (u16) or 1-byte diff token
            assert s[pos] == 100                                       # run_len 100, one byte
            s[pos] = 101
        else:
            s[4 + 1] = 49                                              # one run less than written: trailing bytes
    n = data.size // info.point_step
    codec = native.Codec(native.Plan(info))
    with pytest.raises(native.CloudiniHipError) as e:
        codec.decode_host([s], [n])
    assert e.value.code == -6
    with pytest.raises(Exception):
        oracle.decode_stage1(info, s, n)
    codec.close()


def _wrapping_run_stream(mode):
    """One chunk of an integer-only UINT16 cloud whose SECOND run carries run_len = 2^64 - 1 (nine 0xFF bytes and
    0x01). The reference's bound test `out_index + run_len > n` (v5_codec.cpp:836, :858) wraps on it; a decoder
    that copies the test fills memory far beyond the output."""
    huge = bytes([0xFF] * 9 + [0x01])
    if mode == 2:   # Rle: raw u16 value, uvarint run_len
        body = bytes([2]) + (2).to_bytes(4, "little") + bytes([7, 0, 1]) + bytes([9, 0]) + huge
    else:           # DeltaRle: varint diff, uvarint run_len
        body = bytes([3]) + (2).to_bytes(4, "little") + bytes([0x03, 1]) + bytes([0x03]) + huge
    return np.frombuffer(len(body).to_bytes(4, "little") + body, dtype=np.uint8).copy()


@pytest.mark.parametrize("mode", [2, 3])
def test_run_length_that_wraps_the_bound_check_is_rejected(oracle, mode):
    from cloudini_amd import native
    n = 200
    info, _data = cases.int_only(np.zeros(n, dtype=np.uint16), cases.F.UINT16)
    s = _wrapping_run_stream(mode)
    codec = native.Codec(native.Plan(info))
    out = np.full(n * info.point_step, 0x5A, dtype=np.uint8)
    with pytest.raises(native.CloudiniHipError) as e:
        codec.decode_host([s], [n], out=out)
    assert e.value.code == -6
    with pytest.raises(Exception):
        oracle.decode_stage1(info, s, n)
    # the codec is still usable afterwards (no fault, no hang)
    good = oracle.encode_stage1(info, _data)
    assert np.array_equal(codec.decode_host([good], [n])[0], oracle.decode_stage1(info, good, n))
    codec.close()


@pytest.mark.parametrize("field", ["palette3", "growing"])
def test_section_guess_with_a_false_hit_in_the_token_stream(oracle, field):
    """k_decode_points finds the one section of a chunk by trying every Palette size from the end of the payload. A
    cloud that stands still has a regular stream of 0x01 bytes: the three bytes 01 01 01 read as a Palette header of
    257 entries wherever the 257-entry guess lands. With a real 3-entry Palette behind the stream the smaller (real)
    hit must win and the section is folded; with a section in another mode the false hit is all there is, and the kernel
    must notice that its tiles end elsewhere and leave the section to the section kernels. Either way the bytes are
    the oracle's (decodeV5AdaptiveIntSection, src/v5_codec.cpp:764-879)."""
    from cloudini_amd.schema import FieldType as F
    n = 20000
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001), ("i", 12, F.UINT16, None)]
    info = cases.make_info(fields, 16, n)
    rs = np.random.RandomState(5)
    vals = (rs.randint(0, 3, n) * 11).astype(np.uint16) if field == "palette3" else (np.arange(n) * 3 % 60000).astype(np.uint16)
    data = cases.pack(info, {"x": np.full(n, 1.5, np.float32), "y": np.full(n, -2.25, np.float32),
                             "z": np.full(n, 0.75, np.float32), "i": vals}, n)
    stream, modes = oracle.encode_stage1(info, data, return_modes=True)
    payload = _split_chunks(stream)[0]
    assert (modes.tolist() == [1]) == (field == "palette3")   # "growing": a run-length mode, anything but Palette
    # the 257-entry guess really lands on 01 01 01 inside the token stream
    guess = len(payload) - (3 + 257 * 2 + (9 * n + 7) // 8)
    assert 3 < guess < 3 * n - 3 and bytes(payload[guess:guess + 3]) == b"\x01\x01\x01"
    stats, _modes, n_chunks = _stats_after_decode(oracle, info, data)
    assert stats == (n_chunks, n_chunks, 0, 0)


def test_epoch_time_stamps_stay_on_the_parallel_decoder(oracle):
    """A FLOAT64 stamp with a resolution (1 us) next to XYZI: the first value of every chunk is the epoch time itself, a
    varint of 8 bytes (1.7e15 ticks); later ones are small. The 64-bit token walk takes tokens of up to 10 bytes from
    the history bytes in front of its 8-byte window -- such chunks used to fall back to the one-lane decoder
    (decodeVarint, include/cloudini_lib/encoding_utils.hpp:69-90)."""
    F = cases.F
    n = 70000
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001), ("intensity", 12, F.FLOAT32, 0.001),
              ("ring", 16, F.UINT16, None), ("timestamp", 18, F.FLOAT64, 1e-6)]
    info = cases.make_info(fields, 26, n)
    rs = np.random.RandomState(11)
    p = rs.randn(n, 3).astype(np.float32) * 20
    for t0 in (1.7e9, 9.2e12, 0.0):  # 8-byte, 10-byte (wraps the int64 like the reference) and 1-byte first tokens
        data = cases.pack(info, {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": rs.randint(0, 255, n).astype(np.float32),
                                 "ring": (np.arange(n) % 64).astype(np.uint16), "timestamp": t0 + np.arange(n) * 1e-5}, n)
        check_decode(oracle, info, [data])
        stats, _modes, n_chunks = _stats_after_decode(oracle, info, data)
        assert stats[2] == 0, (t0, stats)


def test_gorilla_coded_streams_stay_on_the_parallel_decoder(oracle):
    """Streams with Gorilla-coded FLOAT64 fields (FieldDecoderFloat_Gorilla, include/cloudini_lib/field_decoder.hpp:158-302;
    the layout of the reference's samples/dds_message.bin) are decoded by the stream kernel's window chain, not one lane per
    chunk: one and two Gorilla fields, window changes, repeated stamps, random bit patterns, raw first values -- and the
    bytes are the oracle's."""
    F = cases.F
    for info, data in (cases.ouster_like(), cases.gorilla_pair()):
        stats, _modes, n_chunks = _stats_after_decode(oracle, info, data)
        assert stats[0] == n_chunks and stats[2] == 0, stats
    # smooth stamps (one window for long stretches) over several chunks, and windows that change every few points
    n = 100_000
    rs = np.random.RandomState(5)
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001), ("intensity", 12, F.FLOAT32, 0.001),
              ("ring", 16, F.UINT16, None), ("timestamp", 18, F.FLOAT64, None)]
    info = cases.make_info(fields, 26, n)
    p = rs.randn(n, 3).astype(np.float32) * 20
    variants = (("smooth", 1.7e9 + np.arange(n) * 1e-5), ("runs", np.repeat(rs.uniform(0, 1, n // 50), 50)),
                ("wild", np.cumsum(rs.choice([1e-9, 1e-3, 7.0, 1e6], n))))
    for kind, stamps in variants:
        data = cases.pack(info, {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": rs.randint(0, 255, n).astype(np.float32),
                                 "ring": (np.arange(n) % 64).astype(np.uint16), "timestamp": stamps}, n)
        stats, _modes, n_chunks = _stats_after_decode(oracle, info, data)
        assert stats[0] + stats[2] == n_chunks
        # (a chunk in which the window changes more than 40 times inside one KiB of stream goes to the serial decoder:
        # the "wild" stamps may do that -- their bytes are still the oracle's)
        if kind != "wild":
            assert stats[0] == n_chunks, (kind, stats)


def test_decode_fill_zero_for_fresh_buffers(oracle):
    """cldn_hip_codec_set_decode_fill(ZERO): the caller's buffer content is not needed, the bytes of a point that no field
    covers read 0 afterwards (what the reference leaves in a freshly resized vector); KEEP (default) preserves them."""
    from cloudini_amd import native
    for info, data in (synth.lidar_xyzi(50000, seed=3), synth.depthcam_xyzrgba(320, 200, seed=4), synth.velodyne_xyzir(40000, seed=5)):
        n = len(data) // info.point_step
        stream = oracle.encode_stage1(info, data)
        codec = native.Codec(native.Plan(info))
        codec.set_decode_fill(True)
        out = np.full(data.size, 0x5A, dtype=np.uint8)
        got = codec.decode_host([stream], [n], out=out)[0]
        assert np.array_equal(got, oracle.decode_stage1(info, stream, n, fill=0))
        codec.set_decode_fill(False)
        out = np.full(data.size, 0x5A, dtype=np.uint8)
        got = codec.decode_host([stream], [n], out=out)[0]
        assert np.array_equal(got, oracle.decode_stage1(info, stream, n, fill=0x5A))
        codec.close()


def test_decode_fill_zero_device_resident(oracle):
    """Device buffers with CLDN_HIP_FILL_ZERO: the two padded layouts the decoder writes as whole 16-byte stores come
    out with zero padding whatever the buffer held; a layout without such stores keeps the buffer's bytes."""
    import torch
    from cloudini_amd import native
    dev = torch.device("cuda", 0)
    xyz20 = cases.make_info([("x", 0, cases.F.FLOAT32, 0.001), ("y", 4, cases.F.FLOAT32, 0.001), ("z", 8, cases.F.FLOAT32, 0.001)],
                            20, 5000)
    rs = np.random.RandomState(3)
    xyz20_data = cases.pack(xyz20, {k: rs.randn(5000).astype(np.float32) * 10 for k in "xyz"}, 5000)
    for (info, data), zeroed in ((synth.lidar_xyzi(70000, seed=8), True), (synth.depthcam_xyzrgba(320, 200, seed=9), True),
                                 ((xyz20, xyz20_data), False)):
        n = len(data) // info.point_step
        stream = oracle.encode_stage1(info, data)
        codec = native.Codec(native.Plan(info), device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
        codec.set_decode_fill(True)
        d_stream = torch.from_numpy(np.ascontiguousarray(stream)).to(dev)
        d_out = torch.full((data.size,), 0x5A, dtype=torch.uint8, device=dev)
        codec.decode_device(d_stream.data_ptr(), np.array([0, stream.size], dtype=np.uint64), np.array([n], dtype=np.uint64),
                            d_out.data_ptr(), data.size, 0)
        codec.status()
        got = d_out.cpu().numpy()
        assert np.array_equal(got, oracle.decode_stage1(info, stream, n, fill=0 if zeroed else 0x5A))
        codec.close()


def test_section_guess_lookalikes_in_lidar_streams(oracle):
    """Token streams of ordinary lidar clouds hold places that read as the header of a Palette section of the very size
    that would put it there (velodyne generator, seeds 45 and 47: 695 entries in chunk 1; XYZI seed 42: 723 and 929 next
    to the real 256): pal_guess_from_end checks its candidates (token end in front, distinct entries, indexes in range)
    before anything is built on them. The bytes are the oracle's whatever the guess does."""
    for gen, n, seed in ((synth.velodyne_xyzir, 130048, 45), (synth.velodyne_xyzir, 130048, 47), (synth.lidar_xyzi, 500000, 42)):
        info, data = gen(n, seed=seed)
        check_decode(oracle, info, [data])
        stats, _modes, n_chunks = _stats_after_decode(oracle, info, data)
        assert stats == (n_chunks, n_chunks, 0, 0)


def test_decode_with_the_chunk_sizes_given(oracle):
    """cldn_hip_decode_stage1_sized: the caller's payload sizes replace the serial walk over the [u32] prefixes; every size
    is still checked against its prefix."""
    from cloudini_amd import native
    info, _ = synth.lidar_xyzi(10)
    clouds = [synth.lidar_xyzi(n, seed=60 + k)[1] for k, n in enumerate([100000, 0, 5, 32768, 70001])]
    codec = native.Codec(native.Plan(info))
    streams, chunk_sizes, _ = codec.encode_host(clouds)
    npts = [c.size // info.point_step for c in clouds]
    want = codec.decode_host(streams, npts)
    got = codec.decode_host(streams, npts, chunk_sizes=chunk_sizes)
    for a, b, cloud in zip(want, got, clouds):
        assert np.array_equal(a, b)
        assert np.array_equal(a, oracle.decode_stage1(info, oracle.encode_stage1(info, cloud), cloud.size // info.point_step))
    assert codec.decode_stats()[0] == len(chunk_sizes)
    for k in (0, len(chunk_sizes) - 1):
        wrong = np.array(chunk_sizes, dtype=np.uint32)
        wrong[k] += 1
        with pytest.raises(native.CloudiniHipError):
            codec.decode_host(streams, npts, chunk_sizes=wrong)
    # a 10-chunk cloud through the device-resident entry point
    import torch
    dev = torch.device("cuda", 0)
    info5, data = synth.lidar_xyz(320000, seed=2)
    c5 = native.Codec(native.Plan(info5))
    d_in = torch.from_numpy(data).to(dev)
    cap = c5.plan.stage1_bound(320000)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(2, dtype=torch.int64, device=dev)
    d_sizes = torch.zeros(10, dtype=torch.int32, device=dev)
    c5.encode_device(d_in.data_ptr(), np.array([320000], dtype=np.uint64), d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr())
    c5.synchronize()
    offs = d_off.cpu().numpy().astype(np.uint64)
    d_dec = torch.zeros(data.size, dtype=torch.uint8, device=dev)
    c5.decode_device(d_out.data_ptr(), offs, np.array([320000], dtype=np.uint64), d_dec.data_ptr(), data.size, d_sizes.data_ptr())
    c5.synchronize()
    c5.status()
    stream = d_out[: int(offs[1])].cpu().numpy()
    assert np.array_equal(d_dec.cpu().numpy(), oracle.decode_stage1(info5, stream, 320000))
    c5.close()
    codec.close()


def test_palette_hint_of_an_earlier_call_never_changes_bytes(oracle):
    """Round 5: when every chunk of a decode call had its section folded as a small Palette the point kernel found by itself,
    the codec's next call does not launch the kernels that locate sections and decode them into columns (a launch hint, like
    the encoder's mode hint). The same codec then gets streams whose sections are NOT Palettes: same bytes as the oracle."""
    from cloudini_amd import native
    info, pal = synth.lidar_xyzi(100_000, seed=3)                    # intensity: 256 levels -> Palette
    rs = np.random.RandomState(9)
    noisy = pal.copy().reshape(-1, info.point_step)
    noisy[:, 12:14] = rs.randint(0, 256, (noisy.shape[0], 2)).astype(np.uint8)   # intensity: 65536 levels -> DeltaVarint
    noisy = noisy.reshape(-1)
    ring = pal.copy().reshape(-1, info.point_step)
    ring[:, 12:14] = (np.arange(ring.shape[0]) % 64).astype(np.uint16).view(np.uint8).reshape(-1, 2)  # -> DeltaRle
    ring = ring.reshape(-1)
    codec = native.Codec(native.Plan(info))
    n = pal.size // info.point_step
    streams = {k: oracle.encode_stage1(info, d) for k, d in (("pal", pal), ("noisy", noisy), ("ring", ring))}
    wants = {k: oracle.decode_stage1(info, s, n, fill=0x77) for k, s in streams.items()}
    for k in ("pal", "pal", "pal", "noisy", "noisy", "pal", "ring", "pal", "pal", "noisy"):
        out = np.full(pal.size, 0x77, dtype=np.uint8)
        got = codec.decode_host([streams[k]], [n], out=out)[0]
        assert np.array_equal(got, wants[k]), k
        codec.synchronize()
    # a batch that mixes them
    out = np.full(3 * pal.size, 0x77, dtype=np.uint8)
    got = codec.decode_host([streams["pal"], streams["noisy"], streams["ring"]], [n, n, n], out=out)
    for g, k in zip(got, ("pal", "noisy", "ring")):
        assert np.array_equal(g, wants[k]), k
    codec.close()


def test_launch_hints_of_earlier_calls_never_change_bytes(oracle):
    """Round 6: the counters behind the launch hints are read back with every 4th decode call / 16th encode call only, and two
    more hints ride on them (dec_dv_hint: which of the DeltaVarint accelerators a call launches; the encoder's mode hint picks the
    fused Palette). ONE codec sees long runs of streams whose integer field takes each of the four modes, then single calls in
    turn: whatever the hints say at that moment, bytes and modes are the oracle's."""
    from cloudini_amd import native
    info, pal = synth.lidar_xyzi(100_000, seed=5)                    # intensity: 256 levels -> Palette
    n = pal.size // info.point_step
    rs = np.random.RandomState(19)

    def with_field(values_u16):
        a = pal.copy().reshape(-1, info.point_step)
        a[:, 12:14] = values_u16.astype(np.uint16).view(np.uint8).reshape(-1, 2)
        return a.reshape(-1)
    clouds = {"pal": pal,
              "noisy": with_field(rs.randint(0, 65536, n)),          # -> DeltaVarint
              "ring": with_field(np.arange(n) % 64),                 # -> DeltaRle
              "const": with_field(np.repeat(rs.randint(0, 65536, n // 500 + 1), 500)[:n])}   # steps of 500 equal values -> Rle
    streams, modes, wants = {}, {}, {}
    for k, d in clouds.items():
        streams[k], modes[k] = oracle.encode_stage1(info, d, return_modes=True)
        wants[k] = oracle.decode_stage1(info, streams[k], n, fill=0x3C)
    assert len({int(m[0]) for m in modes.values()}) == 4, modes     # the four modes are all there
    codec = native.Codec(native.Plan(info))
    order = ["pal"] * 6 + ["noisy"] * 6 + ["ring"] * 6 + ["const"] * 6 + ["noisy"] * 6 + ["pal"] * 6 + ["pal", "noisy", "ring", "const"] * 3
    for i, k in enumerate(order):
        out = np.full(pal.size, 0x3C, dtype=np.uint8)
        got = codec.decode_host([streams[k]], [n], out=out)[0]
        assert np.array_equal(got, wants[k]), (i, k)
    enc_order = ["pal"] * 18 + ["noisy"] * 18 + ["ring"] * 3 + ["pal"] * 2 + ["ring"] * 18 + ["const"] * 17 + ["pal", "noisy", "ring", "const"] * 5
    for i, k in enumerate(enc_order):
        got, _sizes, got_modes = codec.encode_host([clouds[k]])
        assert np.array_equal(got[0], streams[k]), (i, k)
        assert list(got_modes[0]) == list(modes[k]), (i, k)
    # and both directions interleaved on the one codec
    for i, k in enumerate(["noisy", "pal", "ring", "const"] * 4):
        got, _sizes, _m = codec.encode_host([clouds[k]])
        assert np.array_equal(got[0], streams[k]), (i, k)
        out = np.full(pal.size, 0x3C, dtype=np.uint8)
        assert np.array_equal(codec.decode_host([got[0]], [n], out=out)[0], wants[k]), (i, k)
    codec.close()


@pytest.mark.parametrize("parts", [1, 2, 16])
def test_chained_and_split_launches_of_the_point_kernel_agree(oracle, parts):
    """Small batches take the SPLIT launches of k_decode_points_w (round 5: the pieces of a chunk over several workgroups,
    token counts and carries through global memory); batches that fill the chip keep the chained launch. The debug call
    cldn_hip_debug_decode_split (outside the boundary of include/cloudini_hip.h; rounds 4-5 read environment variables for
    this, which the shipped library no longer does) forces one shape: every schema family and the marker / ragged / padded
    cases with the chained launch and with 2 and 16 workgroups per chunk, against the oracle."""
    from cloudini_amd import native
    todo = [(n, i, d) for n, i, d in cases.encode_cases(small=False)]
    for k, gen in enumerate((synth.lidar_xyzi, synth.lidar_xyz, synth.velodyne_xyzir)):
        info, data = gen(130048 + 777 * k, seed=20 + k)
        f32 = data.view(np.uint8).reshape(-1, info.point_step)[:, :4].copy().view(np.float32)
        f32[::1013] = np.nan                       # markers: the lanes reset inside pieces
        d2 = data.view(np.uint8).reshape(-1, info.point_step).copy()
        d2[:, :4] = f32.view(np.uint8).reshape(-1, 4)
        todo.append((gen.__name__ + "_nan", info, d2.reshape(-1)))
    assert len(todo) > 30
    for name, info, data in todo:
        step = info.point_step
        n = data.size // step
        codec = native.Codec(native.Plan(info))
        assert native.lib().cldn_hip_debug_decode_split(codec._h, parts) == 0
        stream = oracle.encode_stage1(info, data)
        out = np.full(max(1, data.size), 0x5A, dtype=np.uint8)
        got = codec.decode_host([stream, stream], [n, n], out=np.concatenate([out, out]))
        want = oracle.decode_stage1(info, stream, n, fill=0x5A)
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want), name
        codec.close()


def test_a_wrong_delta_varint_guess_from_the_end_is_found_out(oracle):
    """k_locate_sections guesses a lone DeltaVarint section from the payload's end (round 6): the (n + 1)-th byte with a clear MSB
    counted from the end is taken for the mode byte if it is 0. Here that byte IS 0 -- a NaN marker of the regular stream -- while
    the section is a short Rle one: k_section_dv_w decodes n "tokens" of regular stream into the column, the point kernel finds
    the regular stream ending elsewhere, and the chunk's sections must be redone from the right place."""
    from cloudini_amd import native
    hits = 0
    for n in range(3000, 3400):
        info, data = synth.lidar_xyzi(n, seed=n)
        a = data.reshape(n, info.point_step).copy()
        a[:, 0:4] = np.frombuffer(np.float32(np.nan).tobytes(), np.uint8)          # every x is NaN: a marker byte per point
        a[:, 12:14] = np.repeat(np.arange((n + 99) // 100, dtype=np.uint16), 100)[:n].view(np.uint8).reshape(n, 2) * 0 + \
            np.repeat((np.arange((n + 99) // 100) % 7 * 1000).astype(np.uint16), 100)[:n].view(np.uint8).reshape(n, 2)  # runs of 100: Rle / DeltaRle
        cloud = a.reshape(-1)
        stream, modes = oracle.encode_stage1(info, cloud, return_modes=True)
        payload = stream[4:]
        ends = np.nonzero((payload & 0x80) == 0)[0]
        if len(ends) < n + 1 or payload[ends[-(n + 1)]] != 0 or ends[-(n + 1)] < 3 * n:
            continue   # the guess would not fire for this cloud
        hits += 1
        codec = native.Codec(native.Plan(info))
        out = np.full(cloud.size, 0x5A, dtype=np.uint8)
        got = codec.decode_host([stream], [n], out=out)[0]
        assert np.array_equal(got, oracle.decode_stage1(info, stream, n, fill=0x5A)), (n, list(modes))
        codec.close()
        if hits >= 6:
            break
    assert hits >= 3
