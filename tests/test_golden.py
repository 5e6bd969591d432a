"""Golden fixtures produced by the real reference (tests/golden/make_golden.py): they pin the CPU oracle without
/root/reference, and -- on the GPU box -- the HIP path end to end (header + framed stage-1 chunks, and decode)."""
import os

import numpy as np
import pytest

from cloudini_amd import api
from cloudini_amd.schema import CompressionOption

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")


def _load():
    z = np.load(GOLDEN)
    out = []
    for name in z["names"]:
        name = str(name)
        yaml = z[name + "/yaml"].tobytes().decode()
        info = api.parse_yaml_info(yaml, int(z[name + "/version"][0]))
        info.use_threads = False
        out.append((name, info, yaml, z[name + "/input"], z[name + "/stream"], z[name + "/decoded_fill5a"]))
    return out


CASES = _load()
IDS = [c[0] for c in CASES]


def _header_len(stream):
    return stream.tobytes().index(b"\0") + 1


@pytest.mark.parametrize("name,info,yaml,data,stream,decoded", CASES, ids=IDS)
def test_oracle_encode_matches_golden(oracle, name, info, yaml, data, stream, decoded):
    assert info.compression_opt == CompressionOption.NONE
    got = oracle.encode_stage1(info, data)
    assert np.array_equal(got, stream[_header_len(stream):])


@pytest.mark.parametrize("name,info,yaml,data,stream,decoded", CASES, ids=IDS)
def test_oracle_decode_matches_golden(oracle, name, info, yaml, data, stream, decoded):
    n = len(data) // info.point_step
    got = oracle.decode_stage1(info, stream[_header_len(stream):], n, fill=0x5A)
    assert np.array_equal(got, decoded)


@pytest.mark.gpu
@pytest.mark.parametrize("name,info,yaml,data,stream,decoded", CASES, ids=IDS)
def test_hip_encoder_matches_golden_stream(name, info, yaml, data, stream, decoded):
    """PointcloudEncoder(info).encode(...) == the reference's bytes, header included."""
    got = api.PointcloudEncoder(info).encode(data)
    assert len(got) == len(stream)
    assert np.array_equal(got, stream)


@pytest.mark.gpu
@pytest.mark.parametrize("name,info,yaml,data,stream,decoded", CASES, ids=IDS)
def test_hip_decoder_matches_golden_decode(name, info, yaml, data, stream, decoded):
    got, hdr = api.PointcloudDecoder().decode_stream(stream, fill=0x5A)
    assert np.array_equal(got, decoded)
    assert hdr.point_step == info.point_step and len(hdr.fields) == len(info.fields)
