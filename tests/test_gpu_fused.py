"""GPU parity of the piece kernel (cloudini_amd/csrc/stage1_fused.h: one wave per 504/378-point piece) and of k_finish
(stage1_finish.h: sections, chunk sizes and placement in one launch) against the oracle and against the tile-kernel
pipeline, on schemas that exercise every section mode, piece / chunk boundaries, padding pieces, forced modes, ragged
batches and repeated calls on one codec."""
import numpy as np
import pytest

import cases
from cloudini_amd import synth
from test_gpu_encode import check_encode

pytestmark = pytest.mark.gpu
F = cases.F


def xyz_plus(n, columns, seed=5, step=None, lanes=3):
    """lidar-like XYZ (or XYZ + f32 intensity) followed by integer columns: [(name, ftype, array)]."""
    rs = np.random.RandomState(seed)
    t = np.arange(n, dtype=np.float32)
    cols = {"x": (20 + 5 * np.sin(t / 97) + rs.normal(0, 0.002, n)).astype(np.float32),
            "y": (3 * np.cos(t / 61) + rs.normal(0, 0.002, n)).astype(np.float32),
            "z": (0.001 * t).astype(np.float32)}
    fields = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001)]
    off = 12
    if lanes == 4:
        cols["i"] = rs.randint(0, 256, n).astype(np.float32)
        fields.append(("i", 12, F.FLOAT32, 0.001))
        off = 16
    for name, ftype, arr in columns:
        cols[name] = arr
        fields.append((name, off, ftype, None))
        off += arr.dtype.itemsize
    step = step or off
    info = cases.make_info(fields, step, n)
    return info, cases.pack(info, cols, n)


def mode_columns(n, seed=9):
    rs = np.random.RandomState(seed)
    i = np.arange(n, dtype=np.int64)
    return {
        "palette": (rs.randint(0, 256, n) * 16).astype(np.uint16),
        "rle": ((i // 256) % 8).astype(np.uint16),
        "rle_long": ((i // 20000) % 3 * 1000).astype(np.uint16),
        "drle": (i % 128).astype(np.uint16),
        "drle_long": (i * 2).astype(np.uint16),
        "delta": rs.randint(0, 65536, n).astype(np.uint16),
        "delta_i16": np.cumsum(rs.randint(-40, 41, n)).astype(np.int16),
        "const": np.full(n, 77, np.uint16),
        "runs127": np.repeat(rs.randint(0, 50, n // 127 + 1), 127)[:n].astype(np.uint16),
        "runs128": np.repeat(rs.randint(0, 50, n // 128 + 1), 128)[:n].astype(np.uint16),
        "runs130": np.repeat(rs.randint(0, 50, n // 130 + 1), 130)[:n].astype(np.uint16),
    }


@pytest.mark.parametrize("kind", sorted(mode_columns(10).keys()))
def test_every_section_mode_behind_xyz(oracle, kind):
    n = 32768 * 2 + 777
    col = mode_columns(n)[kind]
    ftype = F.INT16 if col.dtype == np.int16 else F.UINT16
    info, data = xyz_plus(n, [("v", ftype, col)])
    from cloudini_amd import native
    codec = native.Codec(native.Plan(info))
    assert codec.pipeline(0) == 2  # the piece kernel is the default
    codec.close()
    check_encode(oracle, info, [data])


@pytest.mark.parametrize("n", [1, 2, 62, 63, 64, 503, 504, 505, 1007, 1008, 1009, 2015, 2016, 2017, 4095, 4096, 4097,
                               32255, 32256, 32257, 32767, 32768, 32769, 33271, 33272, 33273, 65536, 100000])
def test_piece_and_chunk_boundaries(oracle, n):
    cols = mode_columns(n, seed=n)
    info, data = xyz_plus(n, [("a", F.UINT16, cols["palette"]), ("b", F.UINT16, cols["drle"])], seed=n)
    check_encode(oracle, info, [data])


@pytest.mark.parametrize("n", [1, 377, 378, 379, 756, 32768, 32769, 70001])
def test_four_lane_pieces(oracle, n):
    cols = mode_columns(n, seed=n + 1)
    info, data = xyz_plus(n, [("ring", F.UINT16, cols["drle"]), ("lab", F.UINT16, cols["rle"])], seed=n, lanes=4)
    check_encode(oracle, info, [data])


def test_many_fields_all_modes(oracle):
    n = 70000
    cols = mode_columns(n, seed=21)
    info, data = xyz_plus(n, [(k, F.INT16 if v.dtype == np.int16 else F.UINT16, v) for k, v in sorted(cols.items())])
    check_encode(oracle, info, [data])


def test_padded_and_unaligned_layouts(oracle):
    n = 50000
    cols = mode_columns(n, seed=4)
    # 16-byte XYZI (C2), 18-byte packed XYZI+ring (C4), 32-byte stride with the u16 deep in the padding
    check_encode(oracle, *[(i, [d]) for i, d in [synth.lidar_xyzi(n, seed=8)]][0])
    check_encode(oracle, *[(i, [d]) for i, d in [synth.velodyne_xyzir(n, seed=8)]][0])
    info, data = xyz_plus(n, [("v", F.UINT16, cols["palette"])], step=32)
    check_encode(oracle, info, [data])
    info, data = xyz_plus(n, [("v", F.UINT16, cols["rle"]), ("w", F.UINT16, cols["delta"])], step=19)
    check_encode(oracle, info, [data])


def test_special_values_and_rare_rows(oracle):
    n = 40000
    rs = np.random.RandomState(3)
    info, data = xyz_plus(n, [("v", F.UINT16, mode_columns(n)["palette"])])
    pts = data.view(np.uint8).reshape(n, info.point_step)
    xyz = pts[:, :12].copy().view(np.float32).reshape(n, 3)
    xyz[rs.randint(0, n, 600), rs.randint(0, 3, 600)] = np.nan
    xyz[rs.randint(0, n, 60), rs.randint(0, 3, 60)] = np.inf
    xyz[rs.randint(0, n, 60), rs.randint(0, 3, 60)] = -np.inf
    xyz[rs.randint(0, n, 60), rs.randint(0, 3, 60)] = 3e9
    xyz[rs.randint(0, n, 60), rs.randint(0, 3, 60)] = -2.5e6
    xyz[503:506] = np.nan                       # across a piece boundary
    xyz[32767:32770, 1] = np.inf                # across a chunk boundary
    pts[:, :12] = xyz.view(np.uint8).reshape(n, 12)
    check_encode(oracle, info, [pts.reshape(-1)])
    # incompressible: every token 4-5 bytes (the LDS region is sized for exactly this)
    big = rs.uniform(-2e6, 2e6, size=(n, 3)).astype(np.float32)
    pts[:, :12] = big.view(np.uint8).reshape(n, 12)
    check_encode(oracle, info, [pts.reshape(-1)])
    huge = (rs.randint(0, 2, size=(n, 3)) * 4e9 - 2e9).astype(np.float32)   # alternating +-2^31: 5-byte tokens everywhere
    pts[:, :12] = huge.view(np.uint8).reshape(n, 12)
    check_encode(oracle, info, [pts.reshape(-1)])


def test_ragged_batches_and_empty_clouds(oracle):
    clouds = []
    info = None
    for k, n in enumerate([5, 40000, 0, 32768, 1, 0, 504, 66000]):
        info, data = synth.lidar_xyzi(n, seed=100 + k)
        clouds.append(data)
    check_encode(oracle, info, clouds)
    check_encode(oracle, info, [clouds[2]])          # a batch of one empty cloud
    check_encode(oracle, info, [clouds[2], clouds[5]])


def test_one_codec_many_calls(oracle):
    """The look-back records and control words must be clean for every call, whatever the previous call did."""
    from cloudini_amd import native
    info, _ = synth.lidar_xyzi(10)
    codec = native.Codec(native.Plan(info))
    assert codec.pipeline(2) == 2
    rs = np.random.RandomState(11)
    for it in range(12):
        clouds = []
        for _k in range(int(rs.randint(1, 5))):
            n = int(rs.choice([0, 1, 300, 504, 5000, 32768, 40000, 70000]))
            _info, data = synth.lidar_xyzi(n, seed=int(rs.randint(0, 1000)))
            if it % 3 == 1 and n:  # a different value population in the same chunk slots
                pts = data.reshape(n, 16).copy()
                pts[:, 12:14] = rs.randint(0, 65536, n).astype(np.uint16).view(np.uint8).reshape(n, 2)
                data = pts.reshape(-1)
            clouds.append(data)
        streams, _cs, modes = codec.encode_host(clouds)
        for k, cloud in enumerate(clouds):
            want, want_modes = oracle.encode_stage1(info, cloud, return_modes=True)
            assert np.array_equal(streams[k], want), (it, k)
            assert list(modes[k]) == list(want_modes)
    codec.close()


def test_forced_modes_chunk_ranges(oracle):
    """Chunk ranges of one cloud with the modes committed elsewhere (cldn_hip_codec_force_modes): concatenation equals
    the whole cloud's stream."""
    from cloudini_amd import native
    info, data = synth.lidar_xyzi(200000, seed=2)
    want, want_modes = oracle.encode_stage1(info, data, return_modes=True)
    codec = native.Codec(native.Plan(info))
    assert codec.pipeline(2) == 2
    codec.force_modes(want_modes)
    step = info.point_step
    parts = []
    for p0, p1 in ((0, 2 * 32768), (2 * 32768, 5 * 32768), (5 * 32768, 200000)):
        s, _, _ = codec.encode_host([data[p0 * step:p1 * step]])
        parts.append(s[0])
    assert np.array_equal(np.concatenate(parts), want)
    codec.force_modes(None)
    codec.close()


def test_wide_integer_fields(oracle):
    info, data = synth.depthcam_xyzrgba(320, 240)
    check_encode(oracle, info, [data])


# ---- one more regular op behind the FloatN lanes (TAIL instantiations of the piece kernel) ------------------------

def _tail_layout(kind, n, seed=3):
    rs = np.random.RandomState(seed)
    t = np.arange(n, dtype=np.float32)
    cols = {"x": (20 + 5 * np.sin(t / 97) + rs.normal(0, 0.002, n)).astype(np.float32),
            "y": (3 * np.cos(t / 61) + rs.normal(0, 0.002, n)).astype(np.float32),
            "z": (0.001 * t).astype(np.float32),
            "i": rs.randint(0, 256, n).astype(np.float32)}
    xyz = [("x", 0, F.FLOAT32, 0.001), ("y", 4, F.FLOAT32, 0.001), ("z", 8, F.FLOAT32, 0.001)]
    stamp = 1.7e9 + np.arange(n) * 1e-5 + rs.normal(0, 1e-7, n)
    if kind == "xyz_rgb_copy_step16":          # PCL PointXYZRGB: packed rgb as a FLOAT32 without resolution
        fields, step = xyz + [("rgb", 12, F.FLOAT32, None)], 16
        cols["rgb"] = rs.randint(0, 1 << 24, n).astype(np.uint32).view(np.float32)
    elif kind == "xyz_u8_copy_step13":          # packed, unaligned points, 1-byte tail
        fields, step = xyz + [("flag", 12, F.UINT8, None)], 13
        cols["flag"] = rs.randint(0, 256, n).astype(np.uint8)
    elif kind in ("dds_gorilla_step26", "dds_stamp_1us_step26"):   # the reference's samples/dds_message.bin layout
        res = None if kind == "dds_gorilla_step26" else 1e-6
        fields = xyz + [("i", 12, F.FLOAT32, 0.001), ("ring", 16, F.UINT16, None), ("timestamp", 18, F.FLOAT64, res)]
        step = 26
        cols["ring"] = (np.arange(n) % 64).astype(np.uint16)
        cols["timestamp"] = stamp.copy()
        cols["timestamp"][::5003] = np.nan
        cols["timestamp"][7::9001] = 1e300      # beyond int64 after scaling: the reference's cvttsd2si result
    elif kind == "xyz_f64_1us_step24":          # aligned, 8 dwords
        fields, step = xyz + [("timestamp", 16, F.FLOAT64, 1e-6)], 24
        cols["timestamp"] = stamp
    elif kind == "xyz_gorilla_step20":          # f64 directly behind the lanes, unaligned step
        fields, step = xyz + [("timestamp", 12, F.FLOAT64, None)], 20
        cols["timestamp"] = np.where(np.arange(n) % 100 < 50, stamp, np.round(stamp))  # repeats -> '0' bits, new windows
    elif kind == "xyz_pad_i_u8_step32":         # PCL layout x y z <pad> intensity, then a 1-byte copy field
        fields = xyz + [("i", 16, F.FLOAT32, 0.001), ("flag", 20, F.UINT8, None)]
        step = 32
        cols["flag"] = rs.randint(0, 256, n).astype(np.uint8)
    elif kind == "xyzi_ring_scalar_f32_step22":  # a lossy float behind an integer field is not fused: Float_Lossy<float>
        fields = xyz + [("i", 12, F.FLOAT32, 0.001), ("ring", 16, F.UINT16, None), ("temp", 18, F.FLOAT32, 0.01)]
        step = 22
        cols["ring"] = (np.arange(n) % 32).astype(np.uint16)
        cols["temp"] = (rs.normal(30, 5, n)).astype(np.float32)
        cols["temp"][::777] = np.nan
    else:
        raise KeyError(kind)
    cols = {k: v for k, v in cols.items() if k in [f[0] for f in fields]}
    info = cases.make_info(fields, step, n)
    return info, cases.pack(info, cols, n)


TAIL_KINDS = ["xyz_rgb_copy_step16", "xyz_u8_copy_step13", "dds_gorilla_step26", "dds_stamp_1us_step26",
              "xyz_f64_1us_step24", "xyz_gorilla_step20", "xyz_pad_i_u8_step32", "xyzi_ring_scalar_f32_step22"]


@pytest.mark.parametrize("kind", TAIL_KINDS)
def test_tail_op_behind_the_floatn_lanes(oracle, kind):
    """Layouts whose regular stream is the fused FloatN encoder plus ONE more per-point encoder (FieldEncoderCopy,
    Float_Lossy<float/double>, Float_Gorilla<double>: include/cloudini_lib/field_encoder.hpp:56-60, :342-357, :156-312)
    take the piece kernel with the tail token appended to every point; the tile/generic pipeline must agree."""
    from cloudini_amd import native
    info, data = _tail_layout(kind, 32768 * 2 + 777)
    codec = native.Codec(native.Plan(info))
    assert codec.pipeline(2) == 2, "the piece kernel must take this layout"
    codec.close()
    streams = check_encode(oracle, info, [data])
    # ragged batch: chunk starts inside the batch, a cloud shorter than a piece, an empty one
    step = info.point_step
    parts = [data[: 40000 * step], data[40000 * step: 40100 * step], data[:0], data[40100 * step:]]
    check_encode(oracle, info, parts)
    # and the way back (these layouts decode through k_decode_varint<8> or, with raw / Gorilla bytes in the point stream,
    # through the serial decoder): bit for bit the oracle's decode, untouched bytes included
    n = data.size // step
    codec = native.Codec(native.Plan(info))
    out = np.full(data.size, 0x5D, dtype=np.uint8)
    got = codec.decode_host([streams[0]], [n], out=out)[0]
    assert np.array_equal(got, oracle.decode_stage1(info, streams[0], n, fill=0x5D))
    codec.close()


# ---- chunk-table output (cldn_hip_encode_stage1_chunks / cldn_hip_frame_chunks) ------------------------------------------

def _read_chunk_table(table, torch):
    """Gather every chunk's payload from the codec's workspace (device pointers) through torch: list of numpy byte arrays."""
    import ctypes as C
    n = table.n_chunks
    spc = table.segments_per_chunk

    def dev_bytes(ptr, nbytes):
        out = torch.empty(max(1, nbytes), dtype=torch.uint8, device="cuda:0")
        assert torch.cuda.current_device() == 0
        rc = C.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(nbytes), 3)  # D2D
        assert rc == 0
        return out[:nbytes].cpu().numpy()
    segs = dev_bytes(table.segments, n * spc * 8).view(np.uint32).reshape(n, spc, 2) if n else np.zeros((0, spc, 2), np.uint32)
    sizes = dev_bytes(table.chunk_sizes, n * 4).view(np.uint32) if n else np.zeros(0, np.uint32)
    flag = int(dev_bytes(table.not_contiguous, 4).view(np.uint32)[0])
    payloads = []
    for c in range(n):
        parts = []
        for off, size in segs[c]:
            if size:
                parts.append(dev_bytes(table.payload_base + c * table.chunk_stride + int(off), int(size)))
        payloads.append(np.concatenate(parts) if parts else np.zeros(0, np.uint8))
        assert payloads[-1].size == int(sizes[c])
    return payloads, segs, flag


@pytest.mark.parametrize("case", ["xyzi", "xyz", "depth_rgba", "velodyne", "two_fields", "ragged"])
def test_chunk_table_holds_the_reference_payloads(oracle, case):
    """Stage 1 without the framing: the chunk table's payloads are the reference's chunk payloads; schemas with at most one
    adaptive field leave every payload as one run of its slot; cldn_hip_frame_chunks writes the framed streams."""
    import torch
    from cloudini_amd import native
    if case == "xyzi":
        info, clouds = synth.lidar_xyzi(70000, seed=3)[0], [synth.lidar_xyzi(70000, seed=3)[1], synth.lidar_xyzi(40000, seed=4)[1]]
    elif case == "xyz":
        info, clouds = synth.lidar_xyz(100000, seed=3)[0], [synth.lidar_xyz(100000, seed=3)[1]]
    elif case == "depth_rgba":
        info, clouds = synth.depthcam_xyzrgba(320, 240, seed=2)[0], [synth.depthcam_xyzrgba(320, 240, seed=2)[1]] * 2
    elif case == "velodyne":
        info, clouds = synth.velodyne_xyzir(130048, seed=3)[0], [synth.velodyne_xyzir(130048, seed=3)[1]]
    elif case == "two_fields":
        n = 70000
        cols = mode_columns(n)
        info, data = xyz_plus(n, [("a", F.UINT16, cols["palette"]), ("b", F.UINT16, cols["drle"])])
        clouds = [data]
    else:
        info = synth.lidar_xyzi(10)[0]
        clouds = [synth.lidar_xyzi(k, seed=20 + k)[1] for k in (0, 5, 40000, 0, 32768)]
    dev = torch.device("cuda", 0)
    codec = native.Codec(native.Plan(info))
    step = info.point_step
    npts = np.array([c.size // step for c in clouds], dtype=np.uint64)
    host = np.concatenate(clouds) if sum(c.size for c in clouds) else np.zeros(1, np.uint8)
    d_in = torch.from_numpy(host).to(dev)
    for _round in range(2):  # (the second call runs with the mode hint of the first)
        table = codec.encode_chunks_device(d_in.data_ptr(), npts)
        codec.synchronize()
        codec.status()
        payloads, segs, not_contiguous = _read_chunk_table(table, torch)
        want_streams = [oracle.encode_stage1(info, c) for c in clouds]
        want_payloads = []
        for s in want_streams:
            o = 0
            while o < s.size:
                size = int.from_bytes(s[o:o + 4].tobytes(), "little")
                want_payloads.append(s[o + 4:o + 4 + size])
                o += 4 + size
        assert len(payloads) == len(want_payloads)
        for k, (a, b) in enumerate(zip(payloads, want_payloads)):
            assert a.size == b.size and np.array_equal(a, b), (case, k)
        if case != "two_fields":
            assert not_contiguous == 0, case
            for c in range(len(payloads)):  # one run: every non-empty segment starts where the one before ends
                nz = [(int(o), int(sz)) for o, sz in segs[c] if sz]
                assert all(nz[i][0] + nz[i][1] == nz[i + 1][0] for i in range(len(nz) - 1))
        # framing of that table = the framed streams
        cap = int(sum(codec.plan.stage1_bound(int(n)) for n in npts))
        d_out = torch.empty(max(1, cap), dtype=torch.uint8, device=dev)
        d_off = torch.zeros(len(clouds) + 1, dtype=torch.int64, device=dev)
        codec.frame_chunks_device(d_out.data_ptr(), cap, d_off.data_ptr())
        codec.synchronize()
        codec.status()
        offs = d_off.cpu().numpy()
        for k, s in enumerate(want_streams):
            got = d_out[int(offs[k]):int(offs[k + 1])].cpu().numpy()
            assert got.size == s.size and np.array_equal(got, s), (case, k)
    # an ordinary call afterwards takes the table away
    codec.encode_host(clouds)
    with pytest.raises(native.CloudiniHipError):
        codec.frame_chunks_device(0, 1 << 40)
    codec.close()


def test_wide_route_through_the_other_entry_points(oracle):
    """Round 5: a schema beyond the launch-argument plan (WIDE route, stage1_wide.h) through the entry points the fuzz family
    does not take: device-resident calls, the chunk table + cldn_hip_frame_chunks, LZ4 blocks on the device (both parameter
    sets), chunk ranges with forced modes -- the same bytes as the one-call host path, which test_gpu_fuzz.py pins to the
    oracle."""
    import ctypes as C
    import torch
    import cases
    from cloudini_amd import native
    info, data = cases.very_wide_schema(9010)          # 33000 points: two chunks
    step = info.point_step
    n = data.size // step
    want, want_modes = oracle.encode_stage1(info, data, return_modes=True)
    dev = torch.device("cuda", 0)
    codec = native.Codec(native.Plan(info))
    # device-resident encode + decode with the encoder's chunk sizes
    d_in = torch.from_numpy(data).to(dev)
    cap = codec.plan.stage1_bound(n)
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(2, dtype=torch.int64, device=dev)
    d_sizes = torch.zeros(2, dtype=torch.int32, device=dev)
    cp = np.array([n], dtype=np.uint64)
    codec.encode_device(d_in.data_ptr(), cp, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)
    codec.synchronize()
    codec.status()
    offs = d_off.cpu().numpy().astype(np.uint64)
    assert int(offs[1]) == want.size and np.array_equal(d_out[: want.size].cpu().numpy(), want)
    d_dec = torch.full((data.size,), 0xC3, dtype=torch.uint8, device=dev)
    codec.decode_device(d_out.data_ptr(), offs, cp, d_dec.data_ptr(), data.size, d_sizes.data_ptr())
    codec.synchronize()
    codec.status()
    assert np.array_equal(d_dec.cpu().numpy(), oracle.decode_stage1(info, want, n, fill=0xC3))
    # chunk table, then its framing
    table = codec.encode_chunks_device(d_in.data_ptr(), cp)
    codec.synchronize()
    codec.status()
    payloads, _segs, not_contiguous = _read_chunk_table(table, torch)
    assert not_contiguous == 0 and table.segments_per_chunk == 1
    o, k = 0, 0
    while o < want.size:
        size = int.from_bytes(want[o:o + 4].tobytes(), "little")
        assert np.array_equal(payloads[k], want[o + 4:o + 4 + size])
        o += 4 + size
        k += 1
    assert k == len(payloads) == 2
    codec.frame_chunks_device(d_out.data_ptr(), cap, d_off.data_ptr())
    codec.synchronize()
    codec.status()
    assert np.array_equal(d_out[: want.size].cpu().numpy(), want)
    # chunk ranges with the modes committed on the head
    codec.force_modes(want_modes)
    a, _, _ = codec.encode_host([data[: 32768 * step]])
    b, _, _ = codec.encode_host([data[32768 * step:]])
    assert np.array_equal(np.concatenate([a[0], b[0]]), want)
    codec.force_modes(None)
    # stage 2 on the device: blocks equal the serial model's and decode to the payloads
    lz4 = C.CDLL("/usr/lib/x86_64-linux-gnu/liblz4.so.1")
    for stage2, params in ((1, (8192, 11, 1024)), (2, (4096, 10, 512))):
        codec.set_stage2(stage2)
        streams, _sz, _m = codec.encode_host([data])
        o, k = 0, 0
        while o < streams[0].size:
            size = int.from_bytes(streams[0][o:o + 4].tobytes(), "little")
            block = np.ascontiguousarray(streams[0][o + 4:o + 4 + size])
            assert np.array_equal(block, oracle.lz4_model(payloads[k].tobytes(), *params)), (stage2, k)
            out = C.create_string_buffer(max(1, payloads[k].size))
            assert lz4.LZ4_decompress_safe(block.ctypes.data_as(C.c_char_p), out, int(block.size), int(payloads[k].size)) == payloads[k].size
            assert out.raw[: payloads[k].size] == payloads[k].tobytes()
            o += 4 + size
            k += 1
        assert k == 2
    codec.set_stage2(0)
    codec.close()
