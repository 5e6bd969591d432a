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
