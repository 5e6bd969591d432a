"""Pins the CPU oracle (oracle/cloudini_oracle.c) to the real reference compiled into oracle/_ref.

Runs without a GPU. When /root/reference is absent and oracle/_ref has not been built these tests skip; the
golden-fixture tests (test_golden.py) still pin the oracle in that situation.
"""
import numpy as np
import pytest

import cases
from cloudini_amd.schema import CompressionOption


def _chunk_modes(stream, n_adaptive_hint=1):
    modes, pos = [], 0
    while pos < len(stream):
        size = int(np.frombuffer(stream[pos:pos + 4].tobytes(), "<u4")[0])
        modes.append(int(stream[pos + 4]))
        pos += 4 + size
    return modes


ALL = cases.encode_cases(small=True)


@pytest.mark.parametrize("name,info,data", ALL, ids=[c[0] for c in ALL])
def test_encode_matches_reference(oracle, reflib, name, info, data):
    want = reflib.encode_stage1(info, data)
    got = oracle.encode_stage1(info, data)
    assert len(got) == len(want)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name,info,data", ALL, ids=[c[0] for c in ALL])
def test_decode_matches_reference(oracle, reflib, name, info, data):
    n = len(data) // info.point_step
    full = reflib.encode(info, data)
    want, _yaml = reflib.decode(full, len(data), fill=0x5A)
    stream = reflib.encode_stage1(info, data)
    got = oracle.decode_stage1(info, stream, n, fill=0x5A)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seed", cases.VERY_WIDE_SEEDS)
def test_very_wide_schemas_match_reference(oracle, reflib, seed):
    """65-200 fields, points of 1-4 KiB (the WIDE route of the library): the oracle against the reference itself, both ways."""
    info, data = cases.very_wide_schema(seed)
    n = len(data) // info.point_step
    want = reflib.encode_stage1(info, data)
    got = oracle.encode_stage1(info, data)
    assert np.array_equal(got, want), seed
    full = reflib.encode(info, data)
    want_dec, _yaml = reflib.decode(full, len(data), fill=0x5A)
    assert np.array_equal(oracle.decode_stage1(info, want, n, fill=0x5A), want_dec), seed


@pytest.mark.parametrize("seed", list(range(7000, 7200)))
def test_corner_schemas_match_reference(oracle, reflib, seed):
    """The dense corner-case generator of tests/test_gpu_fuzz.py (round 6: integer fields packed behind 3 or 4 float lanes at every
    alignment, half of them 64-bit, full-range values): the checker itself against the compiled reference on the seeds the GPU
    test runs, both ways."""
    import test_gpu_fuzz
    info, data = test_gpu_fuzz._corner_case(seed)
    n = len(data) // info.point_step
    want = reflib.encode_stage1(info, data)
    got = oracle.encode_stage1(info, data)
    assert np.array_equal(got, want), seed
    full = reflib.encode(info, data)
    want_dec, _yaml = reflib.decode(full, len(data), fill=0x5A)
    assert np.array_equal(oracle.decode_stage1(info, want, n, fill=0x5A), want_dec), seed


def test_damaged_streams_of_diverse_schemas_match_reference(oracle, reflib):
    """The seeds tests/test_gpu_fuzz.py::test_damaged_streams_of_diverse_schemas_decode_like_the_oracle runs by default: the checker's
    accept / reject decision and bytes against the compiled reference's."""
    import test_gpu_fuzz
    for seed in range(9000, 9200):
        rs, info, data = test_gpu_fuzz._damaged_case(seed)
        n = data.size // info.point_step
        s = oracle.encode_stage1(info, data)
        if len(s) < 8:
            continue
        s = test_gpu_fuzz._damage(rs, s)
        try:
            a = oracle.decode_stage1(info, s, n, fill=0xE1)
        except Exception:
            a = None
        try:
            b = reflib.decode_noheader(info.copy(width=n, height=1), s, fill=0xE1)
        except Exception:
            b = None
        assert (a is None) == (b is None), seed
        if a is not None:
            assert np.array_equal(a, b), seed


def test_reference_mode_bytes(oracle):
    """Per-chunk mode bytes pinned by test_field_encoders.cpp:590-674."""
    for name, info, data, modes in cases.reference_int_sequences():
        stream = oracle.encode_stage1(info, data)
        got = _chunk_modes(stream)
        if modes is None:
            assert len(got) == 2 and all(m != 3 for m in got), name
        else:
            assert got == modes, name
    for name, info, data in cases.probe_boundaries():
        assert all(m == 3 for m in _chunk_modes(oracle.encode_stage1(info, data))), name


def test_known_answer_vectors(oracle):
    """SURVEY.md appendix A.8 (bytes produced by the compiled reference)."""
    for name, info, data, payload in cases.kat_vectors():
        stream = oracle.encode_stage1(info, data).tobytes()
        assert stream[4:] == payload, name
        assert int.from_bytes(stream[:4], "little") == len(payload), name


def test_v5_equals_v4_for_float_only(oracle):
    """PointcloudV5_LossyFloatOnlyRoundTrip (test_field_encoders.cpp:695-769): byte-identical payloads."""
    info, data = cases.xyzi_struct_4133()
    a = oracle.encode_stage1(info, data)
    b = oracle.encode_stage1(info.copy(version=4), data)
    assert np.array_equal(a, b)


def test_bound_matches_reference(oracle, reflib):
    for name, info, data in ALL[:12]:
        n = len(data) // info.point_step
        for pts in (0, 1, n, 32768, 32769, 100000):
            assert oracle.stage1_bound(info, pts) == reflib.max_compressed_size(
                info.copy(compression_opt=CompressionOption.NONE), pts, include_header=False), (name, pts)


def test_varint_differential(oracle):
    """decodeVarintOracle differential of test_field_encoders.cpp:165-278, seed 0xC10D1217 (restated):
    every 1- and 2-byte prefix plus random longer encodings must round-trip or be rejected consistently."""
    for b0 in range(256):
        for b1 in range(256):
            n, val = oracle.decode_varint(bytes([b0, b1]))
            if b0 < 0x80:
                uval, want_n = b0, 1
            elif b1 < 0x80:
                uval, want_n = (b0 & 0x7F) | (b1 << 7), 2
            else:
                assert n < 0  # truncated 3+ byte varint
                continue
            if uval == 0:
                assert n < 0  # the NaN marker is not a varint (encoding_utils.hpp:140-142)
                continue
            u = uval - 1
            assert n == want_n and val == ((u >> 1) ^ -(u & 1))
    rs = np.random.RandomState(0xC10D1217 & 0x7FFFFFFF)
    for _ in range(20000):
        v = int(rs.randint(-2**62, 2**62, dtype=np.int64)) >> int(rs.randint(0, 62))
        enc = oracle.encode_varint64(v)
        n, val = oracle.decode_varint(enc)
        assert n == len(enc) and val == v
        if len(enc) > 1:
            n2, _ = oracle.decode_varint(enc, max_size=len(enc) - 1)
            assert n2 < 0


def _corrupt(rs, s):
    kind = rs.randint(0, 4)
    if kind == 0:
        s = s.copy()
        for _ in range(int(rs.randint(1, 4))):
            s[rs.randint(0, len(s))] ^= np.uint8(1 << rs.randint(0, 8))
        return s
    if kind == 1:
        return s[: rs.randint(1, len(s))]
    if kind == 2:
        pos = rs.randint(4, len(s))
        return np.concatenate([s[:pos], rs.randint(0, 256, int(rs.randint(1, 4))).astype(np.uint8), s[pos:]])
    pos = rs.randint(4, len(s) - 1)
    return np.concatenate([s[:pos], s[pos + 1:]])


def test_decoder_hardening_matches_the_reference_on_damaged_streams(oracle, reflib):
    """The oracle's decoder must accept and reject exactly what the reference's does (and return the same bytes):
    random flips, truncations, insertions and deletions on valid streams of the BASELINE schemas."""
    from cloudini_amd import synth
    agree = 0
    for seed in range(5000, 5240):
        rs = np.random.RandomState(seed)
        pick = rs.randint(0, 4)
        if pick == 0:
            info, data = synth.lidar_xyzi(int(rs.choice([500, 5000, 40000])), seed=seed)
        elif pick == 1:
            info, data = synth.lidar_xyz(int(rs.choice([300, 33000])), seed=seed)
        elif pick == 2:
            info, data = synth.velodyne_xyzir(int(rs.choice([1000, 20000])), seed=seed)
        else:
            info, data = synth.depthcam_xyzrgba(64, 48, seed=seed)
        n = data.size // info.point_step
        s = _corrupt(rs, oracle.encode_stage1(info, data))
        try:
            a = oracle.decode_stage1(info, s, n, fill=0xE1)
        except Exception:
            a = None
        try:
            b = reflib.decode_noheader(info.copy(width=n, height=1), s, fill=0xE1)
        except Exception:
            b = None
        assert (a is None) == (b is None), seed
        if a is not None:
            assert np.array_equal(a, b), seed
        agree += 1
    assert agree == 240


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5"])
def test_full_size_baseline_configs_match_reference(oracle, reflib, name):
    """The oracle against the reference at BASELINE.json's FULL sizes (the cases above use reduced clouds): C2 1M-pt
    XYZI, C3 1280x800 XYZRGBA with NaN pixels, C4 130048-pt packed XYZI+ring, C5 10M-pt XYZ. Encode bytes and decoded
    bytes."""
    from cloudini_amd import synth
    info, data = {"c2": lambda: synth.lidar_xyzi(1_000_000), "c3": lambda: synth.depthcam_xyzrgba(1280, 800),
                  "c4": lambda: synth.velodyne_xyzir(130048), "c5": lambda: synth.lidar_xyz(10_000_000)}[name]()
    n = len(data) // info.point_step
    want = reflib.encode_stage1(info, data)
    got = oracle.encode_stage1(info, data)
    assert len(got) == len(want) and np.array_equal(got, want)
    ref_dec, _yaml = reflib.decode(reflib.encode(info, data), len(data), fill=0x5A)
    assert np.array_equal(oracle.decode_stage1(info, want, n, fill=0x5A), ref_dec)


def _tail_kinds():
    import test_gpu_fused
    return test_gpu_fused.TAIL_KINDS


@pytest.mark.parametrize("kind", _tail_kinds())
def test_tail_layouts_match_reference(oracle, reflib, kind):
    """The layouts the TAIL instantiations of the piece kernel take (FloatN lanes + one more per-point encoder: raw copy,
    Float_Lossy<float/double>, Gorilla) are checked on the GPU against the oracle only: here the oracle itself is pinned
    to the reference on exactly those inputs, encode and decode."""
    import test_gpu_fused
    info, data = test_gpu_fused._tail_layout(kind, 32768 * 2 + 777)
    n = len(data) // info.point_step
    want = reflib.encode_stage1(info, data)
    assert np.array_equal(oracle.encode_stage1(info, data), want)
    full = reflib.encode(info, data)
    ref_dec, _yaml = reflib.decode(full, len(data), fill=0x5A)
    assert np.array_equal(oracle.decode_stage1(info, want, n, fill=0x5A), ref_dec)
