"""CPU: decoder hardening of the oracle beyond what the reference itself survives (the reference's own bound test
wraps on these inputs and writes out of bounds, so it cannot be the checker here)."""
import numpy as np
import pytest

import cases


@pytest.mark.parametrize("mode", [2, 3])
def test_oracle_rejects_run_length_that_wraps_the_bound_check(oracle, mode):
    n = 200
    info, _data = cases.int_only(np.zeros(n, dtype=np.uint16), cases.F.UINT16)
    huge = bytes([0xFF] * 9 + [0x01])   # uvarint 2^64 - 1
    if mode == 2:
        body = bytes([2]) + (2).to_bytes(4, "little") + bytes([7, 0, 1]) + bytes([9, 0]) + huge
    else:
        body = bytes([3]) + (2).to_bytes(4, "little") + bytes([0x03, 1]) + bytes([0x03]) + huge
    s = np.frombuffer(len(body).to_bytes(4, "little") + body, dtype=np.uint8).copy()
    with pytest.raises(Exception):
        oracle.decode_stage1(info, s, n)


def test_oracle_rejects_a_gorilla_window_of_more_than_64_bits(oracle):
    """A '11' token of FieldDecoderFloat_Gorilla whose leading + meaningful bits exceed 64: the reference shifts by
    uint8_t(64 - leading - meaningful) -- undefined behaviour, so it cannot be the checker -- and the oracle (and with it
    every GPU decoder) rejects the stream. Found as the one disagreement of tools/dev/oracle_vs_ref_campaign.py's million
    damaged streams (seed 695298: one bit turns a '10' token into such a '11' token)."""
    from cloudini_amd.schema import FieldType as F
    n = 50
    v = (np.cumsum(np.random.RandomState(3).normal(0, 1e-3, n)) + 5.0).astype(np.float64)
    info = cases.make_info([("t", 0, F.FLOAT64, None)], 8, n)
    s = oracle.encode_stage1(info, cases.pack(info, {"t": v}, n)).copy()
    assert np.array_equal(oracle.decode_stage1(info, s, n), cases.pack(info, {"t": v}, n))
    # the second token of a chunk opens the first window ('11', 5 bits of leading zeros, 6 bits of meaningful - 1): the payload's
    # bytes 8.. hold it. Rewrite its header to leading = 31, meaningful = 64
    tok = 4 + 8
    assert s[tok] & 3 == 3
    hdr = 3 | (31 << 2) | (63 << 7)
    s[tok] = hdr & 0xff
    s[tok + 1] = (int(s[tok + 1]) & 0xe0) | ((hdr >> 8) & 0x1f)
    with pytest.raises(Exception):
        oracle.decode_stage1(info, s, n)
