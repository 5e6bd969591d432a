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
