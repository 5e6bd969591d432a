"""cloudini_amd/csrc/stage1_math.h -- the exact per-value arithmetic the HIP kernels run -- compiled with g++ and
diffed against the oracle on the CPU (no GPU needed): varint tokens, lengths, token concatenation, quantisation."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mathlib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("math") / "libdevmath.so")
    subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off",
                    "-I" + os.path.join(ROOT, "cloudini_amd", "csrc"), os.path.join(ROOT, "tests", "cpu_math_shim.cpp"),
                    "-o", so], check=True)
    L = C.CDLL(so)
    L.m_varint64.argtypes = [C.c_int64, C.c_char_p]
    L.m_varint32.argtypes = [C.c_int32, C.c_char_p]
    L.m_uvarint32.argtypes = [C.c_uint32, C.c_char_p]
    L.m_varint64_len.argtypes = [C.c_int64]
    L.m_varint32_len.argtypes = [C.c_int32]
    L.m_concat.argtypes = [C.c_int64, C.c_int64, C.c_char_p]
    L.m_quant_rne_i32.argtypes = [C.c_float, C.c_float]
    L.m_quant_rne_i32.restype = C.c_int32
    L.m_quant_away_f32.argtypes = [C.c_float, C.c_float]
    L.m_quant_away_f32.restype = C.c_int64
    L.m_quant_away_f64.argtypes = [C.c_double, C.c_double]
    L.m_quant_away_f64.restype = C.c_int64
    L.m_int_as_i64.argtypes = [C.c_uint64, C.c_uint32]
    L.m_int_as_i64.restype = C.c_int64
    return L


def _values():
    rs = np.random.RandomState(1)
    vals = [0, 1, -1, 63, -64, 64, -65, 2**31 - 1, -2**31, 2**31, 2**62, -2**62, 2**63 - 1, -2**63, -2**63 + 1]
    for sh in range(64):
        for d in (-1, 0, 1):
            vals += [(1 << sh) + d, -(1 << sh) + d]
    vals += [int(x) for x in rs.randint(-2**62, 2**62, size=5000, dtype=np.int64)]
    vals += [int(x) >> int(s) for x, s in zip(rs.randint(-2**62, 2**62, size=5000, dtype=np.int64), rs.randint(0, 62, size=5000))]
    return [v for v in vals if -2**63 <= v < 2**63]


def test_varint_tokens_match_oracle(mathlib, oracle):
    buf = C.create_string_buffer(16)
    for v in _values():
        want = oracle.encode_varint64(v)
        n = mathlib.m_varint64(v, buf)
        assert buf.raw[:n] == want and mathlib.m_varint64_len(v) == len(want), v
        if -2**31 <= v < 2**31:
            n = mathlib.m_varint32(v, buf)
            assert buf.raw[:n] == want and mathlib.m_varint32_len(v) == len(want), v


def test_uvarint_and_groups7(mathlib):
    buf = C.create_string_buffer(16)
    for b in range(1, 65):
        assert mathlib.m_groups7(b) == (b + 6) // 7
    for v in [0, 1, 127, 128, 16383, 16384, 32768, 2**21 - 1, 2**21, 2**28 - 1, 2**28, 2**32 - 1]:
        n = mathlib.m_uvarint32(v, buf)
        want, x = bytearray(), v
        while x > 0x7F:
            want.append((x & 0x7F) | 0x80)
            x >>= 7
        want.append(x)
        assert buf.raw[:n] == bytes(want), v


def test_token_concat(mathlib, oracle):
    buf = C.create_string_buffer(16)
    rs = np.random.RandomState(2)
    for _ in range(3000):
        a = int(rs.randint(-2**40, 2**40)) >> int(rs.randint(0, 40))
        b = int(rs.randint(-2**40, 2**40)) >> int(rs.randint(0, 40))
        want = oracle.encode_varint64(a) + oracle.encode_varint64(b)
        n = mathlib.m_concat(a, b, buf)
        assert n == len(want) and buf.raw[:n] == want


def test_quantisation_matches_x86_semantics(mathlib):
    m = np.float32(1.0) / np.float32(0.001)
    # round half to even on exact .5 products, cvtps2dq "integer indefinite" on overflow / NaN / inf
    assert mathlib.m_quant_rne_i32(0.5, 1.0) == 0 and mathlib.m_quant_rne_i32(1.5, 1.0) == 2
    assert mathlib.m_quant_rne_i32(2.5, 1.0) == 2 and mathlib.m_quant_rne_i32(-2.5, 1.0) == -2
    for v in (float("inf"), float("-inf"), float("nan"), 3e9, -3e9, 2147484.0):
        assert mathlib.m_quant_rne_i32(v, float(m)) == -2**31, v
    assert mathlib.m_quant_rne_i32(-2147483.5, float(m)) in (-2147483520, -2147483392, -2**31)  # representable range edge
    # scalar path: half away from zero, int64
    assert mathlib.m_quant_away_f32(0.5, 1.0) == 1 and mathlib.m_quant_away_f32(-0.5, 1.0) == -1
    assert mathlib.m_quant_away_f32(2.5, 1.0) == 3 and mathlib.m_quant_away_f64(-2.5, 1.0) == -3
    assert mathlib.m_quant_away_f64(1e300, 1.0) == -2**63 and mathlib.m_quant_away_f32(float("inf"), 1.0) == -2**63
    # against numpy on random data
    rs = np.random.RandomState(3)
    for v in rs.uniform(-5000, 5000, 2000).astype(np.float32):
        t = np.float32(v) * m
        assert mathlib.m_quant_rne_i32(float(v), float(m)) == int(np.rint(t))


def test_int_field_widening(mathlib):
    assert mathlib.m_int_as_i64(0xFFFF, 3) == -1 and mathlib.m_int_as_i64(0xFFFF, 4) == 65535
    assert mathlib.m_int_as_i64(0xFFFFFFFF, 5) == -1 and mathlib.m_int_as_i64(0xFFFFFFFF, 6) == 2**32 - 1
    assert mathlib.m_int_as_i64(2**64 - 1, 10) == -1 and mathlib.m_int_as_i64(2**63, 9) == -2**63
