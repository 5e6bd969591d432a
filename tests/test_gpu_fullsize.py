"""BASELINE.json's configurations at their full sizes: the oracle is fast enough for a direct
byte comparison of single clouds; whole batches are checked through properties that do not need it (a batch is the
concatenation of its clouds' streams; decode(encode(x)) is within half a tick; chunk sizes add up)."""
import numpy as np
import pytest

from cloudini_amd import synth

pytestmark = pytest.mark.gpu


def _roundtrip_tolerance(info, data, decoded):
    step = info.point_step
    n = data.size // step
    a = data.reshape(n, step)
    b = decoded.reshape(n, step)
    for f in info.fields:
        if int(f.type) == 7 and f.resolution is not None:
            x = a[:, f.offset:f.offset + 4].copy().view(np.float32).reshape(-1)
            y = b[:, f.offset:f.offset + 4].copy().view(np.float32).reshape(-1)
            nan = np.isnan(x)
            assert np.array_equal(nan, np.isnan(y))
            # the reference's own tolerance: res * 1.0001 .. 1.1 (test_field_encoders.cpp:129, :762); half a tick plus
            # float rounding holds here
            assert np.all(np.abs(x[~nan] - y[~nan]) <= f.resolution * 0.5001 + 1e-6 * np.abs(x[~nan]))
        else:
            size = {1: 1, 2: 1, 3: 2, 4: 2, 5: 4, 6: 4, 7: 4, 8: 8, 9: 8, 10: 8}[int(f.type)]
            assert np.array_equal(a[:, f.offset:f.offset + size], b[:, f.offset:f.offset + size])


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c5"])
def test_single_cloud_configs_match_the_oracle(oracle, name):
    from cloudini_amd import native
    info, data = {"c1": lambda: synth.lidar_xyz(65536), "c2": lambda: synth.lidar_xyzi(1_000_000),
                  "c3": lambda: synth.depthcam_xyzrgba(1280, 800), "c5": lambda: synth.lidar_xyz(10_000_000)}[name]()
    n = data.size // info.point_step
    codec = native.Codec(native.Plan(info))
    streams, chunk_sizes, modes = codec.encode_host([data])
    want, want_modes = oracle.encode_stage1(info, data, return_modes=True)
    assert np.array_equal(streams[0], want)
    assert list(modes[0]) == list(want_modes) or not codec.plan.adaptive_fields
    n_chunks = (n + 32767) // 32768
    assert int(np.sum(chunk_sizes[:n_chunks].astype(np.int64))) + 4 * n_chunks == len(want)
    out = np.full(data.size, 0x6B, dtype=np.uint8)
    got = codec.decode_host([want], [n], out=out)[0]
    assert np.array_equal(got, oracle.decode_stage1(info, want, n, fill=0x6B))
    _roundtrip_tolerance(info, data, got)
    assert codec.decode_stats()[2] == 0  # no chunk needed the serial decoder
    codec.close()


def test_c4_batch_of_256_clouds(oracle):
    """256 x 130048-point packed 18-byte clouds in one call: a few clouds against the oracle, the rest through the
    batch property (equal input clouds give equal streams; offsets are the running sum)."""
    from cloudini_amd import native
    distinct = [synth.velodyne_xyzir(130048, seed=42 + k) for k in range(4)]
    info = distinct[0][0]
    clouds = [distinct[k % 4][1] for k in range(256)]
    codec = native.Codec(native.Plan(info))
    streams, _sizes, modes = codec.encode_host(clouds)
    for k in range(4):
        want, want_modes = oracle.encode_stage1(info, clouds[k], return_modes=True)
        assert np.array_equal(streams[k], want)
        assert list(modes[k]) == list(want_modes)
    for k in range(4, 256):
        assert np.array_equal(streams[k], streams[k % 4]), k
    counts = [130048] * 256
    out = np.full(sum(len(c) for c in clouds), 0x19, dtype=np.uint8)
    decoded = codec.decode_host(streams, counts, out=out)
    for k in (0, 1, 2, 3, 255):
        _roundtrip_tolerance(info, clouds[k], decoded[k])
    assert np.array_equal(decoded[200], decoded[200 % 4])
    codec.close()


FULL = {"c1": lambda: synth.lidar_xyz(65536), "c2": lambda: synth.lidar_xyzi(1_000_000),
        "c3": lambda: synth.depthcam_xyzrgba(1280, 800), "c4": lambda: synth.velodyne_xyzir(130048),
        "c5": lambda: synth.lidar_xyz(10_000_000)}


@pytest.mark.parametrize("name", sorted(FULL))
def test_full_size_configs_match_the_reference_itself(reflib, name):
    """BASELINE configs at full size against the compiled reference (oracle/_ref travels with the repository), not via
    the oracle: the framed stage-1 stream the HIP codec produces must be the bytes PointcloudEncoder::encode writes after
    its header with CompressionOption::NONE, and the HIP decode of that stream must be the reference's decode."""
    from cloudini_amd import native
    info, data = FULL[name]()
    n = data.size // info.point_step
    codec = native.Codec(native.Plan(info))
    want = reflib.encode_stage1(info, data)
    for mode in (2, 1):  # piece kernel (where the schema allows it) and tile kernel
        codec.pipeline(mode)
        got = codec.encode_host([data])[0][0]
        assert len(got) == len(want) and np.array_equal(got, want), (name, mode)
    full = reflib.encode(info, data)
    ref_dec, _yaml = reflib.decode(full, data.size, fill=0x3C)
    out = np.full(data.size, 0x3C, dtype=np.uint8)
    assert np.array_equal(codec.decode_host([want], [n], out=out)[0], ref_dec)
    codec.close()


def test_batch_beyond_4_gib(reflib):
    """One call over more than 2^32 input bytes (missing 4): 30 x 10 M-point XYZI clouds = 4.8 GB of points
    (the reference has no such limit: size_t throughout, src/cloudini.cpp:522-560). Four distinct clouds against the compiled
    reference, the other 26 as copies of them; encode and decode device resident (the batch never exists in host memory).
    Byte offsets beyond 2^32 occur in the input, in the decoded output and in the chunk table's first_point * point_step."""
    import torch
    from cloudini_amd import native
    dev = torch.device("cuda", 0)
    n = 10_000_000
    n_clouds, n_distinct = 30, 4
    made = [synth.lidar_xyzi(n, seed=42 + k) for k in range(n_distinct)]
    info = made[0][0]
    step = info.point_step
    assert n_clouds * n * step > (1 << 32)
    want = [reflib.encode_stage1(info, d) for _i, d in made]
    codec = native.Codec(native.Plan(info), device=0, stream=torch.cuda.current_stream(dev).cuda_stream)
    d_distinct = [torch.from_numpy(d).to(dev) for _i, d in made]
    d_points = torch.empty(n_clouds * n * step, dtype=torch.uint8, device=dev)
    for k in range(n_clouds):
        d_points[k * n * step:(k + 1) * n * step] = d_distinct[k % n_distinct]
    cloud_points = np.full(n_clouds, n, dtype=np.uint64)
    cap = codec.plan.stage1_bound(n) * n_clouds
    d_out = torch.empty(cap, dtype=torch.uint8, device=dev)
    d_off = torch.zeros(n_clouds + 1, dtype=torch.int64, device=dev)
    n_chunks = n_clouds * ((n + 32767) // 32768)
    d_sizes = torch.zeros(n_chunks, dtype=torch.int32, device=dev)
    codec.encode_device(d_points.data_ptr(), cloud_points, d_out.data_ptr(), cap, d_off.data_ptr(), d_sizes.data_ptr(), 0)
    codec.synchronize()
    codec.status()
    offs = d_off.cpu().numpy().astype(np.uint64)
    assert int(offs[-1]) == sum(len(want[k % n_distinct]) for k in range(n_clouds))
    d_want = [torch.from_numpy(w).to(dev) for w in want]
    for k in range(n_clouds):
        a, b = int(offs[k]), int(offs[k + 1])
        assert b - a == len(want[k % n_distinct]), k
        assert torch.equal(d_out[a:b], d_want[k % n_distinct]), k
    # and back: the decode of the batch against the reference's decode of the four distinct streams
    ref_dec = []
    for k in range(n_distinct):
        full = reflib.encode(info, made[k][1])
        ref_dec.append(torch.from_numpy(reflib.decode(full, n * step, fill=0x00)[0]).to(dev))
    d_dec = torch.zeros(n_clouds * n * step, dtype=torch.uint8, device=dev)   # (KEEP mode over zeros = the reference over fill 0)
    codec.decode_device(d_out.data_ptr(), offs, cloud_points, d_dec.data_ptr(), n_clouds * n * step, d_sizes.data_ptr())
    codec.synchronize()
    codec.status()
    for k in range(n_clouds):
        assert torch.equal(d_dec[k * n * step:(k + 1) * n * step], ref_dec[k % n_distinct]), k
    assert codec.decode_stats()[2] == 0
    codec.close()
