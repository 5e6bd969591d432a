// stage1_math.h -- per-value arithmetic of the stage-1 codec, shared by every kernel.
//
// Everything here is a pure function of its arguments and is marked host+device so that the exact code the
// kernels run can also be compiled with g++ and checked on the CPU against the oracle
// (tests/test_device_math_cpu.py); the product only ever calls it from HIP kernels.
//
// Semantics being reproduced (reference, paths under /root/reference/cloudini_lib):
//   encodeVarint64            include/cloudini_lib/encoding_utils.hpp:55-67
//   cast_vector4f_to_vector4i include/cloudini_lib/intrinsics.hpp:288-300 (-msse4.1 branch)
//   FieldEncoderFloatN_Lossy  src/field_encoder.cpp:42-91
//   FieldEncoderFloat_Lossy   include/cloudini_lib/field_encoder.hpp:342-357
//   FieldEncoderInt           include/cloudini_lib/field_encoder.hpp:78-85
//   appendUVarint             src/v5_codec.cpp:160-174
#pragma once

#include <stdint.h>

#if defined(__HIPCC__)
#define CLDN_HD __host__ __device__ __forceinline__
#else
#define CLDN_HD static inline
#endif

namespace cldn {

// A token = the encoded bytes of one field of one point: at most 12 bytes, little-endian in w0..w2.
struct Tok {
  uint32_t w0, w1, w2;
  uint32_t len;
};

CLDN_HD uint32_t clz32(uint32_t x) {  // x != 0
  return (uint32_t)__builtin_clz(x);
}
CLDN_HD uint32_t clz64(uint64_t x) {  // x != 0
  return (uint32_t)__builtin_clzll(x);
}

// ceil(bits / 7) for bits in [1, 64]  ==  ((bits + 6) * 37) >> 8   (checked exhaustively on the CPU)
CLDN_HD uint32_t groups7(uint32_t bits) {
  return ((bits + 6u) * 37u) >> 8;
}

// 28 payload bits -> 4 bytes of 7 bits each (no continuation flags yet)
CLDN_HD uint32_t spread28(uint32_t x) {
  return (x & 0x7fu) | ((x & 0x3f80u) << 1) | ((x & 0x1fc000u) << 2) | ((x & 0xfe00000u) << 3);
}

// 0x80 in the low `c` bytes (c in [0,4])
CLDN_HD uint32_t cont_mask(uint32_t c) {
  const uint32_t full = 0x80808080u;
  return c >= 4u ? full : (full & ((1u << (8u * c)) - 1u));
}

CLDN_HD uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// zig-zag(+1) varint of a sign-extended int32 delta: u = zigzag(d) + 1 has at most 33 bits -> 1..5 bytes.
CLDN_HD uint32_t varint32_len(int32_t d) {
  const uint32_t zz = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
  if (zz == 0xffffffffu) return 5u;
  return groups7(32u - clz32(zz + 1u));
}

CLDN_HD Tok varint32_tok(int32_t d) {
  const uint32_t zz = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
  const uint64_t u = (uint64_t)zz + 1u;
  const uint32_t lo = (uint32_t)u;
  const uint32_t bits = (u >> 32) ? 33u : (32u - clz32(lo));
  const uint32_t len = groups7(bits);
  Tok t;
  t.w0 = spread28(lo & 0x0fffffffu) | cont_mask(len - 1u);
  t.w1 = (uint32_t)(u >> 28);  // 5 bits at most, last byte never carries a continuation flag
  t.w2 = 0;
  t.len = len;
  return t;
}

// encodeVarint64 for any int64 (1..10 bytes). x == INT64_MIN wraps u to 0 and emits the single byte 0x00,
// exactly as the reference's loop does.
CLDN_HD uint32_t varint64_len(int64_t x) {
  const uint64_t u = (((uint64_t)x << 1) ^ (uint64_t)(x >> 63)) + 1u;
  if (u == 0) return 1u;
  return groups7(64u - clz64(u));
}

CLDN_HD Tok varint64_tok(int64_t x) {
  const uint64_t u = (((uint64_t)x << 1) ^ (uint64_t)(x >> 63)) + 1u;
  Tok t;
  if (u == 0) {
    t.w0 = t.w1 = t.w2 = 0;
    t.len = 1;
    return t;
  }
  const uint32_t len = groups7(64u - clz64(u));
  const uint32_t c = len - 1u;  // bytes that carry a continuation flag
  const uint32_t hi = (uint32_t)(u >> 56);
  t.w0 = spread28((uint32_t)u & 0x0fffffffu) | cont_mask(min_u32(c, 4u));
  t.w1 = spread28((uint32_t)(u >> 28) & 0x0fffffffu) | cont_mask(c > 4u ? min_u32(c - 4u, 4u) : 0u);
  t.w2 = ((hi & 0x7fu) | ((hi >> 7) << 8)) | cont_mask(c > 8u ? (c - 8u) : 0u);
  t.len = len;
  return t;
}

// plain LEB128 of a run length (<= 32768 -> at most 3 bytes); general up to 2^32-1 (5 bytes)
CLDN_HD uint32_t uvarint32_len(uint32_t v) {
  return v == 0 ? 1u : groups7(32u - clz32(v));
}
CLDN_HD Tok uvarint32_tok(uint32_t v) {
  Tok t;
  const uint32_t len = uvarint32_len(v);
  t.w0 = spread28(v & 0x0fffffffu) | cont_mask(len - 1u);
  t.w1 = v >> 28;
  t.w2 = 0;
  t.len = len;
  return t;
}

CLDN_HD Tok raw_tok(uint64_t bits, uint32_t nbytes) {  // FieldEncoderCopy / XOR residual / palette value
  Tok t;
  t.w0 = (uint32_t)bits;
  t.w1 = (uint32_t)(bits >> 32);
  t.w2 = 0;
  t.len = nbytes;
  return t;
}

CLDN_HD Tok nan_tok() {  // the reserved marker byte 0x00
  Tok t;
  t.w0 = t.w1 = t.w2 = 0;
  t.len = 1;
  return t;
}

// Concatenate token b behind token a (a.len + b.len <= 12).
CLDN_HD Tok tok_concat(Tok a, Tok b) {
  // 96-bit shift of b by 8*a.len, OR into a
  const uint32_t sh = a.len * 8u;
  uint32_t s0, s1, s2;
  if (sh < 32u) {
    const uint64_t lo = ((uint64_t)b.w1 << 32 | b.w0) << sh;
    s0 = (uint32_t)lo;
    s1 = (uint32_t)(lo >> 32);
    s2 = (uint32_t)((((uint64_t)b.w2 << 32 | b.w1) << sh) >> 32);
  } else if (sh < 64u) {
    const uint32_t r = sh - 32u;
    const uint64_t lo = ((uint64_t)b.w1 << 32 | b.w0) << r;
    s0 = 0;
    s1 = (uint32_t)lo;
    s2 = (uint32_t)(lo >> 32);
  } else {
    s0 = 0;
    s1 = 0;
    s2 = b.w0 << (sh - 64u);
  }
  Tok t;
  t.w0 = a.w0 | s0;
  t.w1 = a.w1 | s1;
  t.w2 = a.w2 | s2;
  t.len = a.len + b.len;
  return t;
}

// ---- quantisation ------------------------------------------------------------------------------------

CLDN_HD bool is_nan_f32(float v) { return v != v; }
CLDN_HD bool is_nan_f64(double v) { return v != v; }

// FieldEncoderFloatN_Lossy lane: q = cvtps2dq(roundps(v * m, NEAREST)).
//   * one float32 rounding in the product, round-half-to-even to integer,
//   * x86 "integer indefinite" 0x80000000 when the rounded value is NaN or outside [-2^31, 2^31)
//     (gfx950's v_cvt_i32_f32 would saturate to 0x7fffffff / return 0 for NaN instead).
CLDN_HD int32_t quant_rne_i32(float v, float m) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float t = __fmul_rn(v, m);
  const float r = __builtin_rintf(t);  // v_rndne_f32
#else
  const float t = v * m;
  const float r = __builtin_rintf(t);  // default rounding mode: nearest-even
#endif
  return (r >= -2147483648.0f && r < 2147483648.0f) ? (int32_t)r : (int32_t)0x80000000;
}

// FieldEncoderFloat_Lossy<T>: q = (int64) std::round(v * m)  (half away from zero; the out-of-range cast is
// undefined in C++ -- we pin it to what x86-64 cvttss2si / cvttsd2si produce, INT64_MIN).
CLDN_HD int64_t quant_away_i64_f32(float v, float m) {
#if defined(__HIP_DEVICE_COMPILE__)
  const float t = __fmul_rn(v, m);
#else
  const float t = v * m;
#endif
  const float r = __builtin_roundf(t);
  return (r >= -9223372036854775808.0f && r < 9223372036854775808.0f) ? (int64_t)r
                                                                         : (int64_t)0x8000000000000000ull;
}
CLDN_HD int64_t quant_away_i64_f64(double v, double m) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double t = __dmul_rn(v, m);
#else
  const double t = v * m;
#endif
  const double r = __builtin_round(t);
  return (r >= -9223372036854775808.0 && r < 9223372036854775808.0) ? (int64_t)r
                                                                     : (int64_t)0x8000000000000000ull;
}

// ToInt64<T> for the six adaptive / FieldEncoderInt types (FieldType values 3,4,5,6,9,10)
CLDN_HD int64_t int_field_as_i64(uint64_t raw, uint32_t type) {
  switch (type) {
    case 3: return (int64_t)(int16_t)(uint16_t)raw;
    case 4: return (int64_t)(uint16_t)raw;
    case 5: return (int64_t)(int32_t)(uint32_t)raw;
    case 6: return (int64_t)(uint32_t)raw;
    default: return (int64_t)raw;  // INT64 / UINT64 (reinterpreted)
  }
}

// bitsForPaletteIndex, src/v5_codec.cpp:196-207
CLDN_HD uint32_t palette_bits(uint32_t unique_count) {
  return unique_count <= 1u ? 0u : (32u - clz32(unique_count - 1u));
}

// hashPaletteValue, src/v5_codec.cpp:326-333 (only its mixing quality matters, not its values)
CLDN_HD uint32_t hash_u64(uint64_t v) {
  v ^= v >> 30;
  v *= 0xbf58476d1ce4e5b9ull;
  v ^= v >> 27;
  v *= 0x94d049bb133111ebull;
  v ^= v >> 31;
  return (uint32_t)v;
}

}  // namespace cldn
