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
