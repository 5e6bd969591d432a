"""Randomised schemas (fixed seeds): field types, offsets with padding, resolutions, encoding options and wire versions
drawn at random; GPU encode and decode against the oracle, bit for bit. Round 5: no schema the oracle accepts may be
refused any more (cldn_hip_plan_create has no UNSUPPORTED answer left: schemas beyond the launch-argument plan take the
WIDE route)."""
import numpy as np
import pytest

import cases
from cloudini_amd.schema import EncodingOptions, FieldType as F

pytestmark = pytest.mark.gpu

_SIZE = {F.INT8: 1, F.UINT8: 1, F.INT16: 2, F.UINT16: 2, F.INT32: 4, F.UINT32: 4, F.FLOAT32: 4, F.FLOAT64: 8,
         F.INT64: 8, F.UINT64: 8}
_NP = {F.INT8: np.int8, F.UINT8: np.uint8, F.INT16: np.int16, F.UINT16: np.uint16, F.INT32: np.int32, F.UINT32: np.uint32,
       F.FLOAT32: np.float32, F.FLOAT64: np.float64, F.INT64: np.int64, F.UINT64: np.uint64}


def _random_case(seed, wide=False):
    """wide: up to 64 fields and points of up to 1024 bytes (the limits the library states in include/cloudini_hip.h)."""
    rs = np.random.RandomState(seed)
    n = int(rs.choice([1, 100, 4095, 4097, 33000, 70001])) if not wide else int(rs.choice([1, 100, 4097, 33000]))
    n_fields = int(rs.randint(1, 9)) if not wide else int(rs.choice([9, 16, 33, 40, 64]))
    lead_floats = int(rs.choice([0, 1, 2, 3, 3, 4, 4, 5]))
    types = []
    for i in range(n_fields):
        if i < lead_floats:
            types.append(F.FLOAT32)
        else:
            types.append(F(int(rs.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10]))))
    fields, off, cols = [], int(rs.choice([0, 0, 0, 1, 2, 4])), {}
    for i, t in enumerate(types):
        res = None
        if t == F.FLOAT32 and (i < lead_floats or rs.rand() < 0.5):
            res = float(rs.choice([0.001, 0.01, 0.0005, 1.0]))
        if t == F.FLOAT64 and rs.rand() < 0.5:
            res = float(rs.choice([1e-6, 0.001]))
        name = f"f{i}"
        fields.append((name, off, t, res))
        kind = rs.randint(0, 4)
        if t in (F.FLOAT32, F.FLOAT64):
            if kind == 0:
                v = np.cumsum(rs.normal(0, 0.01, n))
            elif kind == 1:
                v = rs.uniform(-100, 100, n)
            elif kind == 2:
                v = np.round(rs.uniform(-5, 5, n), 2)
            else:
                v = rs.uniform(0, 1, n) + 1.6e9
            v = v.astype(_NP[t])
            if rs.rand() < 0.5 and n > 10:
                v[rs.randint(0, n, max(1, n // 50))] = np.nan
        else:
            info = np.iinfo(_NP[t])
            if kind == 0:
                v = rs.randint(0, 7, n) * 3
            elif kind == 1:
                v = np.arange(n) % 50
            elif kind == 2:
                v = np.repeat(rs.randint(0, 100, n // 300 + 1), 300)[:n]
            else:
                v = rs.randint(max(info.min, -2**62), min(info.max, 2**62), n, dtype=np.int64)
            v = v.astype(_NP[t])
        cols[name] = v
        off += _SIZE[t] + int(rs.choice([0, 0, 0, 1, 2, 4]))
    step = off + int(rs.choice([0, 0, 3, 8]))
    if wide:
        step = max(step, int(rs.choice([257, 300, 512, 777, 1024])))
    enc = EncodingOptions(int(rs.choice([0, 1, 1, 1, 2])))
    version = int(rs.choice([4, 5, 5, 5]))
    info = cases.make_info(fields, step, n, enc=enc, version=version)
    return info, cases.pack(info, cols, n)


import os

# CLDN_FUZZ_EXTRA extra seeds per test (tools/runs/r5_fuzz.sh: 30000); CLDN_FUZZ_BASE: where the extra seeds begin (default:
# right behind the fixed ones -- another value runs a campaign over other schemas)
_EXTRA = int(os.environ.get("CLDN_FUZZ_EXTRA", "0"))
_BASE = int(os.environ.get("CLDN_FUZZ_BASE", "0"))
_SEEDS = list(range(1000, 1100)) + list(range(_BASE or 1100, (_BASE or 1100) + _EXTRA))  # 1082 once caught an uncovered-field bug


@pytest.mark.parametrize("seed", _SEEDS)
def test_random_schema(oracle, seed):
    from cloudini_amd import native
    info, data = _random_case(seed)
    n = data.size // info.point_step
    want, want_modes = oracle.encode_stage1(info, data, return_modes=True)
    plan = native.Plan(info)  # (never refused)
    codec = native.Codec(plan)
    streams, _sizes, modes = codec.encode_host([data])
    assert np.array_equal(streams[0], want), (seed, [(f.name, int(f.type), f.offset, f.resolution) for f in info.fields],
                                              info.point_step, int(info.encoding_opt), info.version)
    if plan.adaptive_fields:
        assert list(modes[0]) == list(want_modes)
    out = np.full(max(1, data.size), 0xC3, dtype=np.uint8)
    got = codec.decode_host([want], [n], out=out)[0]
    assert np.array_equal(got, oracle.decode_stage1(info, want, n, fill=0xC3)), seed
    codec.close()


def _other_entry_points(oracle, codec, plan, info, data, want, n, seed):
    """The legs behind the single-cloud host call (round 6): ragged batches, device-resident buffers at odd addresses, the
    chunk-table entry points, CLDN_HIP_FILL_ZERO -- each for a share of the seeds."""
    from cloudini_amd import native
    if seed % 3 == 0:
        # the same points as a ragged BATCH through the same codec (a cloud that ends inside a chunk, an empty one, one point):
        # every cloud commits its own modes and starts its own delta chains
        rs = np.random.RandomState(seed)
        step = info.point_step
        cuts = sorted({0, n} | {int(c) for c in rs.randint(0, n + 1, 11 if seed % 9 == 0 else 3)})  # (more than 8 clouds: the decoder's tables are uploaded, not passed as a kernel argument)
        parts = [data[a * step:b * step] for a, b in zip(cuts[:-1], cuts[1:])] + [data[:0], data[:step]]
        wants = [oracle.encode_stage1(info, q) for q in parts]
        got_streams, _sizes, _modes = codec.encode_host(parts)
        for k, (g, w) in enumerate(zip(got_streams, wants)):
            assert np.array_equal(g, w), (seed, "batch cloud", k, len(parts[k]) // step)
        npts = [len(q) // step for q in parts]
        out = np.full(max(1, sum(npts) * step), 0x6D, dtype=np.uint8)
        got_clouds = codec.decode_host(wants, npts, out=out)
        for k, g in enumerate(got_clouds):
            assert np.array_equal(g, oracle.decode_stage1(info, wants[k], npts[k], fill=0x6D)), (seed, "batch cloud", k)
    if seed % 5 == 0:
        # DEVICE-RESIDENT buffers at odd addresses: device inputs pick the kernel variant by their address (host inputs are
        # staged into an aligned buffer), the streams are written and read back at an odd address too, the outputs
        # (stream_offsets, chunk_sizes) by the kernels themselves
        import torch
        dev = torch.device("cuda", 0)
        mis_in, mis_out = int(seed // 5 % 4), int(seed // 20 % 4)
        d_buf = torch.zeros(data.size + 16, dtype=torch.uint8, device=dev)
        d_buf[mis_in:mis_in + data.size] = torch.from_numpy(data).to(dev)
        cap = plan.stage1_bound(n)
        d_out = torch.zeros(cap + 16, dtype=torch.uint8, device=dev)
        d_off = torch.zeros(2, dtype=torch.int64, device=dev)
        n_chunks = (n + 32767) // 32768
        d_sizes = torch.zeros(max(1, n_chunks), dtype=torch.int32, device=dev)
        codec.encode_device(d_buf.data_ptr() + mis_in, np.array([n], dtype=np.uint64), d_out.data_ptr() + mis_out, cap,
                            d_off.data_ptr(), d_sizes.data_ptr(), 0)
        codec.status()
        offs = d_off.cpu().numpy().astype(np.uint64)
        assert int(offs[0]) == 0 and int(offs[1]) == want.size, (seed, offs, want.size)
        assert np.array_equal(d_out[mis_out:mis_out + want.size].cpu().numpy(), want), (seed, "device resident", mis_in, mis_out)
        d_dec = torch.full((data.size + 16,), 0x6D, dtype=torch.uint8, device=dev)
        codec.decode_device(d_out.data_ptr() + mis_out, offs, np.array([n], dtype=np.uint64), d_dec.data_ptr() + mis_in, data.size,
                            d_sizes.data_ptr())
        codec.status()
        assert np.array_equal(d_dec[mis_in:mis_in + data.size].cpu().numpy(), oracle.decode_stage1(info, want, n, fill=0x6D)), (seed, "device decode")
    if seed % 7 == 0:
        # the CHUNK-TABLE entry points (cldn_hip_encode_stage1_chunks: intra-chunk placement, sections appended behind the regular
        # stream; cldn_hip_frame_chunks frames the table later, here at an odd output address)
        import torch
        dev = torch.device("cuda", 0)
        d_pts = torch.from_numpy(data).to(dev)
        mis = int(seed // 7 % 16)
        cap = plan.stage1_bound(n)
        d_out = torch.zeros(cap + 32, dtype=torch.uint8, device=dev)
        d_off = torch.zeros(2, dtype=torch.int64, device=dev)
        for _round in range(2):
            codec.encode_chunks_device(d_pts.data_ptr(), np.array([n], dtype=np.uint64))
            codec.frame_chunks_device(d_out.data_ptr() + mis, cap, d_off.data_ptr())
            codec.status()
            total = int(d_off.cpu().numpy()[1])
            assert total == want.size, (seed, "chunk table", total, want.size)
            assert np.array_equal(d_out[mis:mis + total].cpu().numpy(), want), (seed, "chunk table + framing", mis)
    if seed % 13 == 0 and n > 32768:
        # the CHUNK-RANGE sharding of one cloud (the multi-GPU path of one large cloud): the modes are probed on the
        # cloud's head, every rank encodes its contiguous range of whole chunks with cldn_hip_codec_force_modes, the concatenation
        # is the single-GPU stream
        from cloudini_amd import sharding
        step = info.point_step
        world = 2 + seed // 13 % 3
        modes = None
        if plan.adaptive_fields:
            codec.force_modes(None)
            modes = codec.encode_host([data[: min(n, sharding.PROBE_POINTS) * step]])[2][0]
            codec.force_modes(modes)
        parts = []
        for r in range(world):
            p0, cnt = sharding.shard_chunks(n, world, r)
            if cnt:
                parts.append(codec.encode_host([data[p0 * step:(p0 + cnt) * step]])[0][0])
        codec.force_modes(None)
        assert np.array_equal(np.concatenate(parts), want), (seed, "chunk-range shards", world)
    if seed % 11 == 0:
        # CLDN_HIP_FILL_ZERO: the bytes of a point that no field covers may come out as 0 instead of keeping the buffer's content --
        # every covered byte is the oracle's, every uncovered one is the buffer's old byte or 0
        c2 = native.Codec(plan)
        c2.set_decode_fill(True)
        out = np.full(max(1, data.size), 0xE7, dtype=np.uint8)
        got = c2.decode_host([want], [n], out=out)[0].reshape(n, info.point_step)
        ref = oracle.decode_stage1(info, want, n, fill=0xE7).reshape(n, info.point_step)
        covered = np.zeros(info.point_step, dtype=bool)
        for f in info.fields:
            covered[f.offset:f.offset + _SIZE[F(int(f.type))]] = True
        assert np.array_equal(got[:, covered], ref[:, covered]), (seed, "fill zero: covered bytes")
        unc = got[:, ~covered]
        assert np.all((unc == 0xE7) | (unc == 0)), (seed, "fill zero: uncovered bytes")
        c2.close()


def _corner_case(seed):
    """Round 6, after seed 923696: schemas of the shape the piece kernel treats specially, drawn densely -- 3 or 4 leading lossy
    floats, then a few integer fields (half of them 64-bit) packed with gaps of 0..3 bytes so that they land inside, across and
    behind the kernel's loaded window at every alignment, mostly full-range values; now and then a double or a byte behind."""
    rs = np.random.RandomState(seed)
    n = int(rs.choice([4097, 9000, 33000]))
    lanes = int(rs.choice([3, 4]))
    off = int(rs.choice([0, 0, 1, 2, 4]))
    fields, cols = [], {}
    for k in range(lanes):
        fields.append((f"f{k}", off, F.FLOAT32, float(rs.choice([0.001, 0.01, 0.0005]))))
        cols[f"f{k}"] = np.cumsum(rs.normal(0, 0.01, n)).astype(np.float32)
        off += 4 + (int(rs.choice([0, 0, 0, 4])) if k == lanes - 2 else 0)   # (now and then the padded fourth lane of PCL / Ouster points)
    for i in range(int(rs.randint(1, 5))):
        t = F(int(rs.choice([9, 10, 9, 10, 5, 6, 3, 4])))
        off += int(rs.choice([0, 0, 1, 2, 3]))
        ii = np.iinfo(_NP[t])
        kind = rs.randint(0, 5)
        if kind <= 2:
            v = rs.randint(max(ii.min, -2**62), min(ii.max, 2**62), n, dtype=np.int64)
        elif kind == 3:
            v = rs.randint(0, 100, n)
        else:
            v = np.arange(n) % 77
        fields.append((f"i{i}", off, t, None))
        cols[f"i{i}"] = v.astype(_NP[t])
        off += _SIZE[t]
    if rs.rand() < 0.3:
        off += int(rs.choice([0, 1, 3]))
        fields.append(("d", off, F.FLOAT64, None if rs.rand() < 0.5 else 1e-6))
        cols["d"] = (rs.uniform(0, 1, n) + 1.6e9).astype(np.float64)
        off += 8
    if rs.rand() < 0.3:
        fields.append(("b", off, F.UINT8, None))
        cols["b"] = rs.randint(0, 256, n).astype(np.uint8)
        off += 1
    step = off + int(rs.choice([0, 0, 1, 3, 8]))
    info = cases.make_info(fields, step, n, enc=EncodingOptions.LOSSY, version=int(rs.choice([5, 5, 5, 4])))
    return info, cases.pack(info, cols, n)


# (5007720: `out` at an address that is no multiple of 16 -- k_finish laid its copy items out by the stream offset and lost a
# 16-byte unit where the two disagreed about a segment's head across an item boundary; found by this generator's campaign)
@pytest.mark.parametrize("seed", list(range(7000, 7200)) + [5007720] + list(range((_BASE or 7200) + 5_000_000, (_BASE or 7200) + 5_000_000 + _EXTRA // 10)))
def test_corner_schema(oracle, seed):
    from cloudini_amd import native
    info, data = _corner_case(seed)
    n = data.size // info.point_step
    want, want_modes = oracle.encode_stage1(info, data, return_modes=True)
    plan = native.Plan(info)
    codec = native.Codec(plan)
    streams, _sizes, modes = codec.encode_host([data])
    assert np.array_equal(streams[0], want), (seed, [(f.name, int(f.type), f.offset, f.resolution) for f in info.fields], info.point_step, info.version)
    if plan.adaptive_fields:
        assert list(modes[0]) == list(want_modes)
    out = np.full(max(1, data.size), 0x6D, dtype=np.uint8)
    got = codec.decode_host([want], [n], out=out)[0]
    assert np.array_equal(got, oracle.decode_stage1(info, want, n, fill=0x6D)), seed
    _other_entry_points(oracle, codec, plan, info, data, want, n, seed)
    codec.close()


@pytest.mark.parametrize("seed", list(range(11000, 11100)) + list(range((_BASE or 11100) + 11_000_000, (_BASE or 11100) + 11_000_000 + _EXTRA // 10)))
def test_random_schema_through_the_other_entry_points(oracle, seed):
    """The random generator's schemas (every kernel route: generic regular streams, fixed-size streams, lossless floats, ...) through
    the same legs as the corner cases."""
    from cloudini_amd import native
    info, data = _random_case(seed)
    n = data.size // info.point_step
    want = oracle.encode_stage1(info, data)
    plan = native.Plan(info)
    codec = native.Codec(plan)
    streams, _sizes, _modes = codec.encode_host([data])
    assert np.array_equal(streams[0], want), seed
    _other_entry_points(oracle, codec, plan, info, data, want, n, seed * 3 if seed % 3 else seed)  # (every seed takes the batch leg or more)
    codec.close()


@pytest.mark.parametrize("seed", list(range(5000, 5040)))
def test_random_wide_schema(oracle, seed):
    """Up to 64 fields, points of up to 1024 bytes (the launch-argument plan's limits): byte-exact."""
    from cloudini_amd import native
    info, data = _random_case(seed, wide=True)
    n = data.size // info.point_step
    want, want_modes = oracle.encode_stage1(info, data, return_modes=True)
    plan = native.Plan(info)
    codec = native.Codec(plan)
    streams, _sizes, modes = codec.encode_host([data])
    assert np.array_equal(streams[0], want), (seed, len(info.fields), info.point_step, int(info.encoding_opt), info.version)
    if plan.adaptive_fields:
        assert list(modes[0]) == list(want_modes)
    out = np.full(max(1, data.size), 0xC3, dtype=np.uint8)
    got = codec.decode_host([want], [n], out=out)[0]
    assert np.array_equal(got, oracle.decode_stage1(info, want, n, fill=0xC3)), seed
    codec.close()


@pytest.mark.parametrize("seed", cases.VERY_WIDE_SEEDS)
def test_very_wide_schema(oracle, seed):
    """65-200 fields, points of 1-4 KiB: beyond the launch-argument plan of the ordinary kernels (more than 64 per-point
    tokens / adaptive fields / Gorilla-coded doubles, point_step > 1024) -- the WIDE route. Byte-exact both ways, modes
    included; the oracle is pinned to the reference on the same seeds by test_oracle_vs_reference.py."""
    from cloudini_amd import native
    info, data = cases.very_wide_schema(seed)
    n = data.size // info.point_step
    want, want_modes = oracle.encode_stage1(info, data, return_modes=True)
    plan = native.Plan(info)
    codec = native.Codec(plan)
    streams, _sizes, modes = codec.encode_host([data])
    assert np.array_equal(streams[0], want), (seed, len(info.fields), info.point_step, int(info.encoding_opt), info.version)
    if plan.adaptive_fields:
        assert list(modes[0]) == list(want_modes)
    out = np.full(max(1, data.size), 0xC3, dtype=np.uint8)
    got = codec.decode_host([want], [n], out=out)[0]
    assert np.array_equal(got, oracle.decode_stage1(info, want, n, fill=0xC3)), seed
    # a batch of three clouds (one of them empty) through the same codec
    streams3, _s, modes3 = codec.encode_host([data, data[:0], data[: (n // 2) * info.point_step]])
    assert np.array_equal(streams3[0], want) and streams3[1].size == 0
    assert np.array_equal(streams3[2], oracle.encode_stage1(info, data[: (n // 2) * info.point_step]))
    codec.close()


@pytest.mark.parametrize("seed", cases.VERY_WIDE_SEEDS[::8])
def test_very_wide_schema_through_the_host_mirror(reflib, seed):
    """PointcloudEncoder / PointcloudDecoder (header, framing, LZ4 / ZSTD) on very wide schemas against the reference itself."""
    from cloudini_amd import api
    from cloudini_amd.schema import CompressionOption
    info, data = cases.very_wide_schema(seed)
    info = info.copy(compression_opt=CompressionOption(seed % 3))
    want = reflib.encode(info, data)
    got = api.PointcloudEncoder(info).encode(data)
    assert np.array_equal(got, want), seed
    n = data.size // info.point_step
    want_dec, _ = reflib.decode(want, max(1, data.size), fill=0x42)
    got_dec, _got_info = api.PointcloudDecoder().decode_stream(want, fill=0x42)
    assert np.array_equal(got_dec[: n * info.point_step], want_dec[: n * info.point_step]), seed


_CORRUPT_SEEDS = list(range(3000, 3080)) + list(range(_BASE + 2000 if _BASE else 3080, (_BASE + 2000 if _BASE else 3080) + _EXTRA))


@pytest.mark.parametrize("seed", _CORRUPT_SEEDS)
def test_corrupted_streams_decode_like_the_oracle(oracle, seed):
    """Random damage (byte flips, truncation, inserted bytes) to valid streams: whenever the oracle's decoder accepts
    the stream the GPU must return the same bytes, and whenever it rejects it the GPU must report corrupt data."""
    from cloudini_amd import native, synth
    rs = np.random.RandomState(seed)
    pick = rs.randint(0, 5)
    if pick == 0:
        info, data = synth.lidar_xyzi(int(rs.choice([500, 5000, 40000])), seed=seed)
    elif pick == 1:
        info, data = synth.lidar_xyz(int(rs.choice([300, 33000])), seed=seed)
    elif pick == 2:
        info, data = synth.velodyne_xyzir(int(rs.choice([1000, 20000])), seed=seed)
    elif pick == 3:
        info, data = synth.depthcam_xyzrgba(64, 48, seed=seed)
    else:
        info, data = _random_case(1000 + seed % 100)
    n = data.size // info.point_step
    s = oracle.encode_stage1(info, data).copy()
    if len(s) < 8:
        pytest.skip("empty stream")
    kind = rs.randint(0, 4)
    if kind == 0:
        for _ in range(int(rs.randint(1, 4))):
            s[rs.randint(0, len(s))] ^= np.uint8(1 << rs.randint(0, 8))
    elif kind == 1:
        s = s[: rs.randint(1, len(s))]
    elif kind == 2:
        pos = rs.randint(4, len(s))
        s = np.concatenate([s[:pos], rs.randint(0, 256, int(rs.randint(1, 4))).astype(np.uint8), s[pos:]])
    else:
        pos = rs.randint(4, len(s) - 1)
        s = np.concatenate([s[:pos], s[pos + 1:]])
    try:
        want = oracle.decode_stage1(info, s, n, fill=0xE1)
    except Exception:
        want = None
    codec = native.Codec(native.Plan(info))
    out = np.full(max(1, data.size), 0xE1, dtype=np.uint8)
    if want is None:
        with pytest.raises(native.CloudiniHipError) as e:
            codec.decode_host([s], [n], out=out)
        assert e.value.code == -6
    else:
        got = codec.decode_host([s], [n], out=out)[0]
        assert np.array_equal(got, want), seed
    codec.close()


def _damaged_case(seed):
    """Round 6: damage on streams of DIVERSE schemas (the test above draws its random schemas from 100 seeds only): the schema from
    the random generator (even seeds) or the dense corner-case generator (odd seeds) at this very seed, then a flip / truncation /
    insertion / deletion, now and then aimed at a chunk's last bytes (where its sections lie)."""
    rs = np.random.RandomState(seed + 77)
    info, data = _corner_case(seed) if seed & 1 else _random_case(seed)
    return rs, info, data


def _damage(rs, s):
    kind = rs.randint(0, 5)
    s = s.copy()
    if kind == 0:
        for _ in range(int(rs.randint(1, 4))):
            s[rs.randint(0, len(s))] ^= np.uint8(1 << rs.randint(0, 8))
    elif kind == 1:
        s = s[: rs.randint(1, len(s))]
    elif kind == 2:
        pos = rs.randint(4, len(s))
        s = np.concatenate([s[:pos], rs.randint(0, 256, int(rs.randint(1, 4))).astype(np.uint8), s[pos:]])
    elif kind == 3:
        pos = rs.randint(4, len(s) - 1)
        s = np.concatenate([s[:pos], s[pos + 1:]])
    else:  # the first chunk's tail: its sections
        size = int(s[0]) | int(s[1]) << 8 | int(s[2]) << 16 | int(s[3]) << 24
        end = min(len(s), 4 + size)
        pos = max(4, end - 1 - int(rs.randint(0, min(64, max(1, end - 4)))))
        s[pos] = np.uint8(rs.randint(0, 256))
    return s


@pytest.mark.parametrize("seed", list(range(9000, 9200)) + list(range((_BASE or 9200) + 9_000_000, (_BASE or 9200) + 9_000_000 + _EXTRA // 10)))
def test_damaged_streams_of_diverse_schemas_decode_like_the_oracle(oracle, seed):
    from cloudini_amd import native
    rs, info, data = _damaged_case(seed)
    n = data.size // info.point_step
    s = oracle.encode_stage1(info, data)
    if len(s) < 8:
        pytest.skip("empty stream")
    s = _damage(rs, s)
    try:
        want = oracle.decode_stage1(info, s, n, fill=0xE1)
    except Exception:
        want = None
    codec = native.Codec(native.Plan(info))
    out = np.full(max(1, data.size), 0xE1, dtype=np.uint8)
    if want is None:
        with pytest.raises(native.CloudiniHipError) as e:
            codec.decode_host([s], [n], out=out)
        assert e.value.code == -6, seed
    else:
        got = codec.decode_host([s], [n], out=out)[0]
        assert np.array_equal(got, want), seed
    codec.close()


@pytest.mark.parametrize("seed", list(range(7000, 7040)))
def test_host_mirror_full_streams_match_the_reference(reflib, seed):
    """PointcloudEncoder / PointcloudDecoder of the host mirror (header, chunk framing, NONE / LZ4 / ZSTD, with and
    without the worker thread) against the compiled reference on random schemas: identical streams, identical decode."""
    from cloudini_amd import api
    from cloudini_amd.schema import CompressionOption
    rs = np.random.RandomState(seed)
    info, data = _random_case(2000 + seed)
    info = info.copy(compression_opt=CompressionOption(int(rs.choice([0, 1, 2]))), use_threads=bool(rs.randint(0, 2)))
    want = reflib.encode(info, data)
    got = api.PointcloudEncoder(info).encode(data)
    assert np.array_equal(got, want), (seed, int(info.compression_opt), info.use_threads)
    n = data.size // info.point_step
    want_dec, _ = reflib.decode(want, max(1, data.size), fill=0x42)
    got_dec, got_info = api.PointcloudDecoder().decode_stream(want, fill=0x42)
    assert np.array_equal(got_dec[: n * info.point_step], want_dec[: n * info.point_step]), seed


@pytest.mark.parametrize("seed", list(range(7000, 7060)) + list(range((_BASE or 7060) + 13_000_000, (_BASE or 7060) + 13_000_000 + _EXTRA // 50)))
def test_host_mirror_corner_streams_match_the_reference(reflib, seed):
    """The same through the host mirror for the dense corner cases (round 6): the full stream -- header, framing, NONE / LZ4 / ZSTD --
    equals the compiled reference's, and decodes to the same points."""
    from cloudini_amd import api
    from cloudini_amd.schema import CompressionOption
    rs = np.random.RandomState(seed)
    info, data = _corner_case(seed)
    info = info.copy(compression_opt=CompressionOption(int(rs.choice([0, 1, 2]))), use_threads=bool(rs.randint(0, 2)))
    want = reflib.encode(info, data)
    got = api.PointcloudEncoder(info).encode(data)
    assert np.array_equal(got, want), (seed, int(info.compression_opt), info.use_threads)
    n = data.size // info.point_step
    want_dec, _ = reflib.decode(want, max(1, data.size), fill=0x42)
    got_dec, got_info = api.PointcloudDecoder().decode_stream(want, fill=0x42)
    assert np.array_equal(got_dec[: n * info.point_step], want_dec[: n * info.point_step]), seed


# ---- device LZ4 on the random schemas: every chunk's block equals the serial model's (oracle/lz4_model.c), both modes ----

_LZ4_PARAMS = {1: (8192, 11, 1024), 2: (4096, 10, 512)}  # CLDN_HIP_STAGE2_LZ4, CLDN_HIP_STAGE2_LZ4_FAST (stage1_launch.h)
_LZ4_SEEDS = list(range(1000, 1040)) + list(range(_BASE or 1040, (_BASE or 1040) + _EXTRA // 20))


@pytest.mark.parametrize("seed", _LZ4_SEEDS)
def test_random_schema_with_device_lz4(oracle, seed):
    """The stage-1 streams of random schemas hold everything from noise (random 64-bit integers) to long repeats
    (constant fields, raw copies of padded structures): literal runs of kilobytes, matches that fill a sub-range, lists
    that run full. Blocks byte-equal to the model, sizes and stream offsets consistent."""
    from cloudini_amd import native
    info, data = _random_case(seed)
    n = data.size // info.point_step
    want = oracle.encode_stage1(info, data)
    payloads, o = [], 0
    while o < want.size:
        size = int.from_bytes(want[o:o + 4].tobytes(), "little")
        payloads.append(want[o + 4:o + 4 + size])
        o += 4 + size
    codec = native.Codec(native.Plan(info))
    half = data[: (n // 2) * info.point_step]
    for stage2 in (1, 2):
        codec.set_stage2(stage2)
        streams, chunk_sizes, _modes = codec.encode_host([data, half, data])
        assert np.array_equal(streams[0], streams[2]), (seed, stage2)
        blocks, o = [], 0
        while o < streams[0].size:
            size = int.from_bytes(streams[0][o:o + 4].tobytes(), "little")
            blocks.append(streams[0][o + 4:o + 4 + size])
            o += 4 + size
        assert o == streams[0].size and len(blocks) == len(payloads), (seed, stage2)
        assert [b.size for b in blocks] == [int(x) for x in chunk_sizes[: len(blocks)]], (seed, stage2)
        for k, (block, payload) in enumerate(zip(blocks, payloads)):
            model = oracle.lz4_model(payload, *_LZ4_PARAMS[stage2])
            assert block.size == model.size and np.array_equal(block, model), (seed, stage2, k)
    codec.close()
