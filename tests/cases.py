"""Seeded parity cases shared by the CPU tests (oracle vs reference) and the GPU tests (HIP vs oracle).

Where a case restates a generator of the reference's own test-suite the source is cited
(paths under /root/reference/cloudini_lib/test/).
"""
from __future__ import annotations

import numpy as np

from cloudini_amd import synth
from cloudini_amd.schema import CompressionOption, EncodingInfo, EncodingOptions, FieldType, PointField

F = FieldType

_NP = {F.INT8: "<i1", F.UINT8: "<u1", F.INT16: "<i2", F.UINT16: "<u2", F.INT32: "<i4", F.UINT32: "<u4",
       F.FLOAT32: "<f4", F.FLOAT64: "<f8", F.INT64: "<i8", F.UINT64: "<u8"}


def make_info(fields, step, n, enc=EncodingOptions.LOSSY, version=5, comp=CompressionOption.NONE):
    return EncodingInfo(fields=[PointField(*f) for f in fields], width=n, height=1, point_step=step,
                        encoding_opt=enc, compression_opt=comp, version=version, use_threads=False)


def pack(info, columns, n):
    """columns: dict name -> array. Returns the AoS bytes (padding = 0xA5 so that it is visibly untouched)."""
    dt = np.dtype({"names": [f.name for f in info.fields], "formats": [_NP[f.type] for f in info.fields],
                   "offsets": [f.offset for f in info.fields], "itemsize": info.point_step})
    raw = np.full(n * info.point_step, 0xA5, dtype=np.uint8)
    pts = raw.view(dt)
    for f in info.fields:
        pts[f.name] = columns[f.name]
    return raw


def int_only(values, ftype):
    """makeV5IntOnlyInfo / encodeV5IntOnly, test_field_encoders.cpp:336-362."""
    values = np.asarray(values)
    n = len(values)
    info = make_info([("value", 0, ftype, None)], values.dtype.itemsize, n)
    return info, np.ascontiguousarray(values).view(np.uint8).reshape(-1)


def reference_int_sequences():
    """PointcloudV5_AdaptiveIntModes_RoundTripAndModeSelection, test_field_encoders.cpp:590-674.
    Yields (name, info, data, expected per-chunk mode bytes or None)."""
    n = 32 * 1024 + 19
    i = np.arange(n, dtype=np.int64)
    yield "ref_u32_linear", *int_only((100000 + i * 3).astype(np.uint32), F.UINT32), [3, 3]
    yield "ref_u32_mod4", *int_only((i % 4).astype(np.uint32), F.UINT32), [1, 1]
    yield "ref_u16_steps", *int_only(((i // 256) % 8).astype(np.uint16), F.UINT16), [2, 2]
    diff = np.where((i // 64) % 2 == 0, 3, 7)
    yield "ref_u32_two_slopes", *int_only((1000 + np.cumsum(diff)).astype(np.uint32), F.UINT32), [3, 3]
    yield "ref_i32_descending", *int_only((200000 - i * 5).astype(np.int32), F.INT32), [3, 3]
    rs = np.random.RandomState(12345)
    yield "ref_u32_random16", *int_only(rs.randint(0, 0x10000, size=n).astype(np.uint32), F.UINT32), None


def probe_boundaries():
    """PointcloudV5_AdaptiveProbeBoundaries_RoundTrip, test_field_encoders.cpp:676-693."""
    for n in (4095, 4096, 4097, 32 * 1024, 32 * 1024 + 7):
        i = np.arange(n, dtype=np.int64)
        yield f"probe_{n}", *int_only((1000 + i * 3).astype(np.uint32), F.UINT32)


def xyzi_struct_4133():
    """PointcloudV5_LossyFloatOnlyRoundTrip, test_field_encoders.cpp:695-769: 4133 XYZI float points."""
    n = 4133
    i = np.arange(n, dtype=np.float32)
    cols = {"x": (0.001 * i).astype(np.float32), "y": (1.0 + 0.002 * i).astype(np.float32),
            "z": (-2.0 + 0.003 * i).astype(np.float32), "intensity": (i % 251).astype(np.float32)}
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001),
              ("intensity", 12, F.FLOAT32, 0.001)]
    info = make_info(fields, 16, n)
    return info, pack(info, cols, n)


def header_test_struct(n=1000, version=5):
    """DefaultV5AndExplicitV4RoundTrip, test_header.cpp:142-163: XYZI(float) + ring(u16) + time(u32)."""
    rs = np.random.RandomState(7)
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001),
              ("intensity", 12, F.FLOAT32, 0.001), ("ring", 16, F.UINT16, None), ("time", 20, F.UINT32, None)]
    info = make_info(fields, 24, n, version=version)
    i = np.arange(n)
    cols = {"x": rs.uniform(-10, 10, n).astype(np.float32), "y": rs.uniform(-10, 10, n).astype(np.float32),
            "z": rs.uniform(-2, 2, n).astype(np.float32), "intensity": rs.randint(0, 255, n).astype(np.float32),
            "ring": (i % 16).astype(np.uint16), "time": (i * 1000).astype(np.uint32)}
    return info, pack(info, cols, n)


def mixed_schema(n=40000, version=5, enc=EncodingOptions.LOSSY, seed=11):
    """Every regular codec the HIP path has a kernel for, unaligned offsets, padding bytes."""
    rs = np.random.RandomState(seed)
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.002), ("z", 8, F.FLOAT32, 0.0005),
              ("flag", 12, F.UINT8, None), ("rgb", 13, F.FLOAT32, None), ("temp", 17, F.FLOAT32, 0.01),
              ("stamp", 21, F.FLOAT64, 1e-6), ("label", 29, F.INT16, None), ("id", 31, F.UINT64, None),
              ("s8", 39, F.INT8, None), ("count", 41, F.INT32, None)]
    step = 47
    info = make_info(fields, step, n, enc=enc, version=version)
    i = np.arange(n)
    t = np.cumsum(rs.uniform(1e-5, 3e-5, n)) + 1.7e9
    cols = {"x": np.cumsum(rs.normal(0, 0.01, n)).astype(np.float32),
            "y": rs.uniform(-50, 50, n).astype(np.float32),
            "z": (np.sin(i / 100.0) * 3).astype(np.float32),
            "flag": rs.randint(0, 256, n).astype(np.uint8),
            "rgb": rs.randint(0, 2**32, n, dtype=np.uint32).view(np.float32),
            "temp": rs.uniform(-40, 90, n).astype(np.float32),
            "stamp": t,
            "label": rs.randint(-5, 5, n).astype(np.int16),
            "id": (rs.randint(0, 2**63 - 1, n, dtype=np.int64).astype(np.uint64) * 2 + 1),
            "s8": rs.randint(-128, 128, n).astype(np.int8),
            "count": (i // 7 - 3000).astype(np.int32)}
    cols["temp"][rs.randint(0, n, 200)] = np.nan
    cols["stamp"][rs.randint(0, n, 50)] = np.nan
    cols["x"][rs.randint(0, n, 100)] = np.nan
    return info, pack(info, cols, n)


def five_floats(n=20000):
    """5 leading lossy floats -> five scalar FieldEncoderFloat_Lossy (half-away-from-zero), SURVEY A.8 row 2."""
    rs = np.random.RandomState(13)
    fields = [(c, 4 * k, F.FLOAT32, 0.001 * (k + 1)) for k, c in enumerate("abcde")]
    info = make_info(fields, 20, n)
    cols = {c: np.cumsum(rs.normal(0, 0.05, n)).astype(np.float32) for c in "abcde"}
    cols["a"][::7] = np.round(cols["a"][::7], 2) + 0.0005  # half ticks
    cols["c"][rs.randint(0, n, 30)] = np.nan
    cols["d"][rs.randint(0, n, 5)] = np.inf
    return info, pack(info, cols, n)


def two_floats_then_ints(n=33000):
    """Only 2 leading lossy floats (no FloatN fusion) + adaptive ints of every width."""
    rs = np.random.RandomState(17)
    fields = [("u", 0, F.FLOAT32, 0.01), ("v", 4, F.FLOAT32, 0.01), ("a16", 8, F.INT16, None),
              ("b32", 10, F.INT32, None), ("c64", 14, F.INT64, None), ("d64", 22, F.UINT64, None),
              ("e16", 30, F.UINT16, None)]
    info = make_info(fields, 32, n)
    i = np.arange(n, dtype=np.int64)
    cols = {"u": rs.uniform(0, 1, n).astype(np.float32), "v": rs.uniform(0, 1, n).astype(np.float32),
            "a16": (rs.randint(0, 5, n) * 1000 - 2000).astype(np.int16),          # palette
            "b32": (-(i * 11) + 5).astype(np.int32),                               # delta-rle
            "c64": np.repeat(rs.randint(-2**62, 2**62, n // 500 + 1, dtype=np.int64), 500)[:n],  # rle, 64 bit
            "d64": rs.randint(0, 2**63 - 1, n, dtype=np.int64).astype(np.uint64) * 2,  # delta-varint, 10-byte tokens
            "e16": rs.randint(0, 65536, n).astype(np.uint16)}
    return info, pack(info, cols, n)


def palette_stress(kind, n=70000, seed=23):
    """Palette committed on the first 4096 values, then chunks that stress the table."""
    rs = np.random.RandomState(seed)
    if kind == "grows_u16":        # few values in the probe window, every value afterwards
        v = np.concatenate([rs.randint(0, 4, 4096), rs.randint(0, 65536, n - 4096)]).astype(np.uint16)
        return int_only(v, F.UINT16)
    if kind == "all_distinct_u32":  # > 6144 distinct values per chunk: multi-partition table path
        v = np.concatenate([rs.randint(0, 3, 4096) * 1000003, rs.permutation(n - 4096) * 7919 + 17]).astype(np.uint32)
        return int_only(v, F.UINT32)
    if kind == "wide_u64":
        pal = rs.randint(0, 2**63 - 1, 300, dtype=np.int64).astype(np.uint64) * 2 + 1
        return int_only(pal[rs.randint(0, 300, n)], F.UINT64)
    if kind == "single_value":
        return int_only(np.full(n, 77, dtype=np.uint16), F.UINT16)
    if kind == "two_values_i16":
        return int_only(np.where(rs.randint(0, 2, n) == 0, -1, 12345).astype(np.int16), F.INT16)
    if kind == "u5000_u32":         # 5000 distinct values: fits one LDS table pass
        pal = rs.randint(0, 2**32, 5000, dtype=np.uint32)
        return int_only(pal[rs.randint(0, 5000, n)], F.UINT32)
    raise KeyError(kind)


def rle_stress(kind, n=70000, seed=29):
    rs = np.random.RandomState(seed)
    if kind == "constant":
        return int_only(np.full(n, 0xBEEF, dtype=np.uint16), F.UINT16)
    if kind == "long_and_short":   # runs crossing tile (1024) and chunk (32768) boundaries, plus singletons
        lens = np.concatenate([rs.randint(1, 4, 3000), rs.randint(900, 3000, 30)])
        rs.shuffle(lens)
        vals = rs.randint(0, 1 << 31, len(lens))
        v = np.repeat(vals, lens)[:n]
        if len(v) < n:
            v = np.concatenate([v, np.full(n - len(v), 5)])
        return int_only(v.astype(np.uint32), F.UINT32)
    if kind == "alternating":
        v = np.concatenate([np.full(4096, 9), np.arange(n - 4096) % 2]).astype(np.uint16)
        return int_only(v, F.UINT16)
    if kind == "delta_runs_i64":   # delta-rle with 64-bit slopes
        slopes = rs.randint(-2**40, 2**40, n // 300 + 1, dtype=np.int64)
        d = np.repeat(slopes, 300)[:n]
        return int_only(np.cumsum(d).astype(np.int64), F.INT64)
    if kind == "drle_then_noise":  # DeltaRle committed, later values random: 11-byte runs
        v = np.concatenate([np.arange(5000) * 3, rs.randint(0, 2**32, n - 5000)]).astype(np.uint32)
        return int_only(v, F.UINT32)
    raise KeyError(kind)


def float_specials(n=6000, seed=3, lanes=3):
    rs = np.random.RandomState(seed)
    pts = rs.uniform(-100, 100, size=(n, lanes)).astype(np.float32)
    for val in (np.nan, np.inf, -np.inf, 3e9, -3e9, 2147483.648, -2147483.648, 1e-42, -0.0, 2147483.5):
        pts[rs.randint(0, n, 40), rs.randint(0, lanes, 40)] = val
    pts[10:40] = np.round(pts[10:40]) + 0.0005     # exact half ticks at 1 mm (rounding mode)
    pts[40:70] = np.round(pts[40:70]) - 0.0015
    pts[100:110] = np.nan                          # whole NaN points, consecutive
    fields = [("xyzw"[k], 4 * k, F.FLOAT32, 0.001) for k in range(lanes)]
    info = make_info(fields, 4 * lanes, n)
    return info, pts.view(np.uint8).reshape(-1)


def float_domain_boundary(n=30000, seed=53, lanes=3, with_u16=False):
    """Values straddling the |ticks| = 2^21 limit of the kernels' float-domain token path (2097.152 m at 1 mm), mixed
    with quiet stretches, signed zeros and sub-tick values, so wave rows alternate between the two code paths."""
    rs = np.random.RandomState(seed)
    pts = rs.uniform(-2300, 2300, size=(n, lanes)).astype(np.float32)
    pts[: n // 3] = np.cumsum(rs.normal(0, 0.01, size=(n // 3, lanes)), axis=0).astype(np.float32)
    edge = np.float32(2097.152)
    for k, val in enumerate((edge, -edge, np.nextafter(edge, np.float32(0)), np.nextafter(-edge, np.float32(0)),
                             np.nextafter(edge, np.float32(1e9)), np.float32(2097.1515), np.float32(-2097.1525),
                             np.float32(-0.0), np.float32(0.0), np.float32(-0.0004), np.float32(0.0004),
                             np.float32(4194.304), np.float32(-4194.303))):
        pts[rs.randint(n // 3, n, 60), rs.randint(0, lanes, 60)] = val
        pts[2000 + 7 * k, :] = val                       # also inside the quiet stretch: one rare row among common ones
    pts[5000:5200] = np.float32(2097.151)                # constant run just below the limit ...
    pts[5200:5400] = np.float32(-2097.151)               # ... and a 2^22-tick jump between two in-range values
    fields = [("xyzw"[k], 4 * k, F.FLOAT32, 0.001) for k in range(lanes)]
    step = 4 * lanes
    cols = {"xyzw"[k]: pts[:, k].copy() for k in range(lanes)}
    if with_u16:
        fields.append(("i", step, F.UINT16, None))
        cols["i"] = (rs.randint(0, 200, n) * 3).astype(np.uint16)
        step += 4
    info = make_info(fields, step, n)
    return info, pack(info, cols, n)


def region_overflow(n=70000, seed=71, lanes=3, with_u16=False, pad=0):
    """Round 5: the piece kernel's LDS region holds 3 bytes per token; pieces (378 / 504 points) whose tokens are larger
    on average are rewritten by its slow path. Stretches of every length around a piece -- quiet, 4-byte tokens (noise
    over kilometres), 5-byte tokens (|ticks| beyond 2^27, int32 wrap-around), NaN / Inf inside the noisy stretches --
    so that overflowing and fitting pieces alternate inside a workgroup and across chunk boundaries."""
    rs = np.random.RandomState(seed)
    pts = np.cumsum(rs.normal(0, 0.01, size=(n, lanes)), axis=0).astype(np.float32)
    pos = 100
    k = 0
    while pos < n - 3000:
        length = int(rs.choice([30, 200, 378, 504, 505, 800, 1600, 2500]))
        kind = k % 4
        if kind == 0:
            pts[pos:pos + length] = rs.uniform(-3000, 3000, size=(length, lanes))          # 4-byte tokens
        elif kind == 1:
            pts[pos:pos + length] = rs.uniform(-2.0e6, 2.0e6, size=(length, lanes))        # 5-byte tokens, wrap-around
        elif kind == 2:
            pts[pos:pos + length] = rs.uniform(-900, 900, size=(length, lanes))            # 3-byte tokens: just fits
        else:
            blk = rs.uniform(-5000, 5000, size=(length, lanes)).astype(np.float32)
            blk[rs.randint(0, length, max(1, length // 9)), rs.randint(0, lanes, max(1, length // 9))] = np.nan
            blk[rs.randint(0, length, 3), rs.randint(0, lanes, 3)] = np.inf
            pts[pos:pos + length] = blk
        pos += length + int(rs.choice([0, 1, 63, 500, 1500]))
        k += 1
    pts[32768 - 200:32768 + 300] = rs.uniform(-4000, 4000, size=(500, lanes))               # across the chunk boundary
    fields = [("xyzw"[j], pad + 4 * j, F.FLOAT32, 0.001) for j in range(lanes)]
    step = pad + 4 * lanes
    cols = {"xyzw"[j]: pts[:, j].copy() for j in range(lanes)}
    if with_u16:
        fields.append(("i", step, F.UINT16, None))
        cols["i"] = (rs.randint(0, 200, n) * 3).astype(np.uint16)
        step += 2 if pad else 4
    info = make_info(fields, step, n)
    return info, pack(info, cols, n)


def raw_fields_between_varints(n_raw_f32, n_u8=0, n=45000, seed=81):
    """Round 5: x y z (lossy varints) + n_raw_f32 FLOAT32 fields without resolution (FieldEncoderCopy: raw bytes) + n_u8 UINT8
    fields -- point forms of 3 + 4 * n_raw_f32 + n_u8 states for the byte automaton of the decoder
    (stage1_decode_automaton.h: at most 8 states in a dword, at most 16 in a 64-bit word, beyond that the FORM kernel)."""
    rs = np.random.RandomState(seed)
    _, xyz = synth.lidar_xyz(n, seed=seed)
    p = xyz.view(np.float32).reshape(n, 3).copy()
    p[rs.randint(0, n, 40)] = np.nan
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001)]
    cols = {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2]}
    off = 12
    for k in range(n_raw_f32):
        fields.append((f"r{k}", off, F.FLOAT32, None))
        cols[f"r{k}"] = rs.randint(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)  # any bytes, MSBs included
        off += 4
    for k in range(n_u8):
        fields.append((f"b{k}", off, F.UINT8, None))
        cols[f"b{k}"] = rs.randint(0, 256, n).astype(np.uint8)
        off += 1
    info = make_info(fields, off + (4 - off % 4) % 4, n)
    return info, pack(info, cols, n)


def padded_fourth_lane(kind, n=90000, seed=61):
    """Real-world layouts whose fused FloatN encoder has its 4th lane one dword further: PCL PointXYZI (x y z pad
    intensity@16, 32-byte points) and an Ouster-style 48-byte point with five integer channels behind the floats."""
    rs = np.random.RandomState(seed)
    _, xyz = synth.lidar_xyz(n, seed=seed)
    p = xyz.view(np.float32).reshape(n, 3).copy()
    p[rs.randint(0, n, 50)] = np.nan
    inten = rs.randint(0, 256, n).astype(np.float32)
    inten[rs.randint(0, n, 30)] = np.nan
    if kind == "pcl_xyzi":
        fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001),
                  ("intensity", 16, F.FLOAT32, 0.01)]
        info = make_info(fields, 32, n)
        return info, pack(info, {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": inten}, n)
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001),
              ("intensity", 16, F.FLOAT32, 0.001), ("t", 20, F.UINT32, None), ("reflectivity", 24, F.UINT16, None),
              ("ring", 26, F.UINT16, None), ("ambient", 28, F.UINT16, None), ("range", 32, F.UINT32, None)]
    info = make_info(fields, 48, n)
    cols = {"x": p[:, 0], "y": p[:, 1], "z": p[:, 2], "intensity": inten, "t": (np.arange(n) * 97).astype(np.uint32),
            "reflectivity": rs.randint(0, 256, n).astype(np.uint16), "ring": (np.arange(n) % 64).astype(np.uint16),
            "ambient": rs.randint(0, 5000, n).astype(np.uint16),
            "range": (np.nan_to_num(np.linalg.norm(p, axis=1)) * 1000).astype(np.uint32)}
    return info, pack(info, cols, n)


def stride_variants():
    """Same XYZ+u16 content at awkward strides / offsets (unaligned loads, padding untouched)."""
    out = []
    n = 50000
    rs = np.random.RandomState(31)
    base = {"x": np.cumsum(rs.normal(0, 0.02, n)).astype(np.float32), "y": rs.uniform(-5, 5, n).astype(np.float32),
            "z": rs.uniform(-1, 1, n).astype(np.float32), "i": (rs.randint(0, 64, n) * 4).astype(np.uint16)}
    for name, offs, step in (("step14", (0, 4, 8, 12), 14), ("step18_off2", (2, 6, 10, 14), 18),
                             ("step32", (0, 4, 8, 16), 32), ("step19_odd", (1, 5, 9, 15), 19),
                             ("step64", (8, 12, 16, 40), 64), ("step200", (100, 104, 108, 190), 200)):
        fields = [("x", offs[0], F.FLOAT32, 0.001), ("y", offs[1], F.FLOAT32, 0.001),
                  ("z", offs[2], F.FLOAT32, 0.001), ("i", offs[3], F.UINT16, None)]
        info = make_info(fields, step, n)
        out.append((name, info, pack(info, base, n)))
    return out


def kat_vectors():
    """known-answer vectors produced by the compiled reference (res = 1.0 unless stated).
    Returns (name, info, data, expected stage-1 payload bytes of the single chunk)."""
    def f32(rows, res=1.0):
        a = np.array(rows, dtype=np.float32)
        lanes = a.shape[1]
        info = make_info([(f"f{k}", 4 * k, F.FLOAT32, res) for k in range(lanes)], 4 * lanes, a.shape[0])
        return info, a.view(np.uint8).reshape(-1)

    inf, nan = np.inf, np.nan
    yield ("kat_half_even", *f32([(0.5, 1.5, 2.5), (-0.5, -1.5, -2.5)]), bytes.fromhex("010505010808"))
    yield ("kat_5_scalar_half_away", *f32([(0.5, 1.5, 2.5, -0.5, -1.5)]), bytes.fromhex("0305070204"))
    yield ("kat_overflow", *f32([(inf, -inf, 3e9), (1, 1, 1)]),
           bytes.fromhex("8080808010" * 3 + "feffffff0f" * 3))
    yield ("kat_nan", *f32([(nan, 5, 5), (7, 5, 5), (8, 6, 5)]), bytes.fromhex("000b0b0f0101030301"))
    yield ("kat_4lane", *f32([(1, 2, 3, 4), (2, 2, 2, 2)]), bytes.fromhex("0305070903010204"))
    dt = np.dtype({"names": ["x", "y", "z", "i"], "formats": ["<f4", "<f4", "<f4", "<u2"],
                   "offsets": [0, 4, 8, 12], "itemsize": 16})
    pts = np.zeros(4, dtype=dt)
    rows = [(0.0005, 0.0015, 0.0025, 100), (1, 2, 3, 101), (-1, -2.5, 1e-4, 102), (0.5, 0.25, 0.125, 103)]
    for k, r in enumerate(rows):
        pts[k] = r
    info = synth.xyzi_info(4)
    yield ("kat_xyzi_u16", info, pts.view(np.uint8).reshape(-1),
           bytes.fromhex("010305d10f9f1fed2ea01fa846f02eb917fd2afb01" + "00c901030303"))


def ouster_like(n=40000, seed=37, enc=EncodingOptions.LOSSY):
    """The layout of the reference's samples/dds_message.bin (test_ros_msg.cpp:110-125): XYZI float32 + ring u16 +
    FLOAT64 time stamp WITHOUT resolution -> Gorilla codec inside the per-point stream, 26-byte unaligned points."""
    rs = np.random.RandomState(seed)
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001),
              ("intensity", 12, F.FLOAT32, 0.001), ("ring", 16, F.UINT16, None), ("timestamp", 18, F.FLOAT64, None)]
    info = make_info(fields, 26, n, enc=enc)
    i = np.arange(n)
    t = 1.7e9 + np.cumsum(rs.uniform(0.8e-6, 1.2e-6, n))
    t[1000:1100] = t[1000]                       # repeated stamps (xor == 0)
    t[2000:2010] = rs.uniform(0, 1e300, 10)      # wild exponents: window resets, > 31 leading zeros nowhere
    t[3000:3005] = 0.0
    t[3005] = np.nan
    t[4000:4064] = np.frombuffer(rs.bytes(64 * 8), dtype=np.float64)  # random bit patterns
    cols = {"x": np.cumsum(rs.normal(0, 0.01, n)).astype(np.float32), "y": rs.uniform(-5, 5, n).astype(np.float32),
            "z": rs.uniform(-1, 1, n).astype(np.float32), "intensity": rs.randint(0, 255, n).astype(np.float32),
            "ring": (i % 64).astype(np.uint16), "timestamp": t}
    return info, pack(info, cols, n)


def gorilla_pair(n=35000, seed=41):
    """Two Gorilla-coded doubles (one slowly varying, one with tiny integer steps -> long shared windows) between
    other fields, LOSSLESS mode (FLOAT32 -> XOR)."""
    rs = np.random.RandomState(seed)
    fields = [("a", 0, F.FLOAT64, None), ("f", 8, F.FLOAT32, None), ("b", 12, F.FLOAT64, None), ("k", 20, F.UINT8, None)]
    info = make_info(fields, 21, n, enc=EncodingOptions.LOSSLESS)
    cols = {"a": np.sin(np.arange(n) / 500.0) * 1000.0, "f": rs.uniform(-1, 1, n).astype(np.float32),
            "b": np.floor(np.arange(n) / 7.0) * 0.5, "k": rs.randint(0, 256, n).astype(np.uint8)}
    return info, pack(info, cols, n)


_FT_SIZE = {F.INT8: 1, F.UINT8: 1, F.INT16: 2, F.UINT16: 2, F.INT32: 4, F.UINT32: 4, F.FLOAT32: 4, F.FLOAT64: 8,
            F.INT64: 8, F.UINT64: 8}
_FT_NP = {F.INT8: np.int8, F.UINT8: np.uint8, F.INT16: np.int16, F.UINT16: np.uint16, F.INT32: np.int32, F.UINT32: np.uint32,
          F.FLOAT32: np.float32, F.FLOAT64: np.float64, F.INT64: np.int64, F.UINT64: np.uint64}


def very_wide_schema(seed):
    """Round 5: schemas beyond the launch-argument plan of the ordinary kernels -- 65 to 200 fields (so more than 64
    per-point tokens and / or more than 64 adaptive integer fields, sometimes more than 64 Gorilla-coded doubles) and
    points of 1 to 4 KiB. The reference has no such limits (src/codec_common.cpp:116-153, src/v5_codec.cpp:719-740);
    the library's WIDE route (stage1_wide.h) takes them. Fixed seeds; every encoding option and wire version."""
    rs = np.random.RandomState(seed)
    n = int(rs.choice([1, 100, 4097, 9000])) if seed % 10 else 33000        # every tenth seed: two chunks
    n_fields = int(rs.choice([65, 70, 100, 130, 200])) if n < 33000 else int(rs.choice([65, 80]))
    flavour = int(rs.randint(0, 4))   # 0 mixed, 1 mostly integers (adaptive fields), 2 mostly doubles (Gorilla), 3 mostly floats
    lead_floats = int(rs.choice([0, 2, 3, 4, 5]))
    types = []
    for i in range(n_fields):
        if i < lead_floats:
            types.append(F.FLOAT32)
        elif flavour == 1 and rs.rand() < 0.85:
            types.append(F(int(rs.choice([3, 4, 5, 6, 9, 10]))))
        elif flavour == 2 and rs.rand() < 0.8:
            types.append(F.FLOAT64)
        elif flavour == 3 and rs.rand() < 0.8:
            types.append(F.FLOAT32)
        else:
            types.append(F(int(rs.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10]))))
    fields, off, cols = [], int(rs.choice([0, 0, 1, 4])), {}
    for i, t in enumerate(types):
        res = None
        if t == F.FLOAT32 and (i < lead_floats or rs.rand() < 0.5):
            res = float(rs.choice([0.001, 0.01, 1.0]))
        if t == F.FLOAT64 and rs.rand() < (0.15 if flavour == 2 else 0.5):
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
            v = v.astype(_FT_NP[t])
            if rs.rand() < 0.3 and n > 10:
                v[rs.randint(0, n, max(1, n // 50))] = np.nan
        else:
            ii = np.iinfo(_FT_NP[t])
            if kind == 0:
                v = rs.randint(0, 7, n) * 3
            elif kind == 1:
                v = np.arange(n) % 50
            elif kind == 2:
                v = np.repeat(rs.randint(0, 100, n // 300 + 1), 300)[:n]
            else:
                v = rs.randint(max(ii.min, -2**62), min(ii.max, 2**62), n, dtype=np.int64)
            v = v.astype(_FT_NP[t])
        cols[name] = v
        off += _FT_SIZE[t] + int(rs.choice([0, 0, 0, 1, 2, 4, 16]))
    step = max(off + int(rs.choice([0, 3, 8])), int(rs.choice([1025, 1500, 2048, 3000, 4096])))
    enc = EncodingOptions(int(rs.choice([0, 1, 1, 1, 2])))
    version = int(rs.choice([4, 5, 5, 5]))
    info = make_info(fields, step, n, enc=enc, version=version)
    return info, pack(info, cols, n)


VERY_WIDE_SEEDS = list(range(9000, 9040))


def encode_cases(small=False):
    """(name, info, data) for every schema family; `small` trims sizes for the CPU-only suite."""
    out = []
    for name, info, data, _modes in reference_int_sequences():
        out.append((name, info, data))
    for name, info, data in probe_boundaries():
        out.append((name, info, data))
    out.append(("xyzi_struct_4133", *xyzi_struct_4133()))
    out.append(("header_struct_v5", *header_test_struct(1000, 5)))
    out.append(("header_struct_v4", *header_test_struct(1000, 4)))
    out.append(("c2_xyzi", *synth.lidar_xyzi(100000 if small else 300000)))
    out.append(("c3_depthcam", *(synth.depthcam_xyzrgba(320, 240) if small else synth.depthcam_xyzrgba(640, 400))))
    out.append(("c4_velodyne", *synth.velodyne_xyzir(130048)))
    out.append(("mixed_v5", *mixed_schema(40000, 5)))
    out.append(("mixed_v4", *mixed_schema(40000, 4)))
    out.append(("mixed_lossless", *mixed_schema_lossless(30000)))
    out.append(("mixed_none", *mixed_schema(20000, 5, EncodingOptions.NONE)))
    for name, case in lossless_float_clouds(small):
        out.append((name, *case))
    out.append(("ouster_like_gorilla", *ouster_like()))
    out.append(("gorilla_pair_lossless", *gorilla_pair()))
    out.append(("five_floats", *five_floats()))
    out.append(("two_floats_then_ints", *two_floats_then_ints()))
    for kind in ("grows_u16", "all_distinct_u32", "wide_u64", "single_value", "two_values_i16", "u5000_u32"):
        out.append((f"palette_{kind}", *palette_stress(kind)))
    for kind in ("constant", "long_and_short", "alternating", "delta_runs_i64", "drle_then_noise"):
        out.append((f"rle_{kind}", *rle_stress(kind)))
    out.append(("float_specials3", *float_specials(lanes=3)))
    out.append(("float_specials4", *float_specials(lanes=4, seed=4)))
    out.append(("float_boundary3", *float_domain_boundary(lanes=3)))
    out.append(("float_boundary4", *float_domain_boundary(lanes=4, seed=54)))
    out.append(("float_boundary3_u16", *float_domain_boundary(lanes=3, seed=55, with_u16=True)))
    out.append(("region_overflow3", *region_overflow(lanes=3)))
    out.append(("region_overflow4", *region_overflow(lanes=4, seed=72)))
    out.append(("region_overflow3_u16", *region_overflow(lanes=3, seed=73, with_u16=True)))
    out.append(("region_overflow3_u16_unaligned", *region_overflow(lanes=3, seed=74, with_u16=True, pad=1)))
    out.append(("region_overflow4_unaligned", *region_overflow(n=40000, lanes=4, seed=75, pad=2)))
    out.append(("raw_form_7_states", *raw_fields_between_varints(1)))
    out.append(("raw_form_8_states", *raw_fields_between_varints(1, 1, seed=82)))
    out.append(("raw_form_11_states", *raw_fields_between_varints(2, seed=83)))
    out.append(("raw_form_16_states", *raw_fields_between_varints(3, 1, seed=84)))
    out.append(("raw_form_17_states", *raw_fields_between_varints(3, 2, n=20000, seed=85)))
    out.append(("pcl_xyzi_step32", *padded_fourth_lane("pcl_xyzi")))
    out.append(("ouster_step48", *padded_fourth_lane("ouster")))
    out.extend(stride_variants())
    return out


def lossless_float_clouds(small=False):
    """EncodingOptions::LOSSLESS clouds made of floats only: every per-point encoder is FieldEncoderFloat_XOR (fixed-size
    tokens) -- the layouts k_encode_fixed / k_decode_fixed take, with their 16-byte fast path (x y z intensity), the general
    path (12-byte points, doubles, a padded stride) and chunk tails that are no multiple of a tile or of a sub-stream."""
    rs = np.random.RandomState(77)
    out = []
    n = 40001 if small else 70001
    f4 = [("x", 0, F.FLOAT32, None), ("y", 4, F.FLOAT32, None), ("z", 8, F.FLOAT32, None), ("intensity", 12, F.FLOAT32, None)]
    info = make_info(f4, 16, n, enc=EncodingOptions.LOSSLESS)
    cols = {"x": np.cumsum(rs.normal(0, 0.01, n)).astype(np.float32), "y": rs.uniform(-5, 5, n).astype(np.float32),
            "z": rs.uniform(-1, 1, n).astype(np.float32), "intensity": rs.randint(0, 255, n).astype(np.float32)}
    cols["y"][100:110] = np.nan
    cols["z"][5] = np.inf
    out.append(("lossless_xyzi", (info, pack(info, cols, n))))
    n3 = 32768 + 1
    f3 = [("x", 0, F.FLOAT32, None), ("y", 4, F.FLOAT32, None), ("z", 8, F.FLOAT32, None)]
    info3 = make_info(f3, 12, n3, enc=EncodingOptions.LOSSLESS)
    out.append(("lossless_xyz12", (info3, pack(info3, {k: cols[k][:n3] for k in ("x", "y", "z")}, n3))))
    nd = 33000
    fd = [("a", 0, F.FLOAT64, 1e-6), ("f", 8, F.FLOAT32, None), ("b", 16, F.FLOAT64, 1e-9)]   # padded 28-byte stride, a gap at 12
    infod = make_info(fd, 28, nd, enc=EncodingOptions.LOSSLESS)
    out.append(("lossless_f64_padded", (infod, pack(infod, {"a": np.cumsum(rs.uniform(0, 1e-3, nd)), "f": rs.uniform(-1, 1, nd).astype(np.float32),
                                                             "b": np.sin(np.arange(nd) / 300.0)}, nd))))
    return out


def mixed_schema_lossless(n=30000):
    """LOSSLESS: FLOAT32 -> XOR, ints -> FieldEncoderInt, FLOAT64 with a resolution -> XOR<double>
    (a FLOAT64 without resolution would select Gorilla, which the HIP path rejects)."""
    rs = np.random.RandomState(19)
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, None), ("t", 8, F.FLOAT64, 1e-6),
              ("k", 16, F.UINT16, None), ("b", 18, F.UINT8, None)]
    info = make_info(fields, 20, n, enc=EncodingOptions.LOSSLESS)
    cols = {"x": rs.uniform(-1, 1, n).astype(np.float32), "y": np.cumsum(rs.normal(0, 1, n)).astype(np.float32),
            "t": np.cumsum(rs.uniform(0, 1e-3, n)), "k": rs.randint(0, 100, n).astype(np.uint16),
            "b": rs.randint(0, 256, n).astype(np.uint8)}
    return info, pack(info, cols, n)
