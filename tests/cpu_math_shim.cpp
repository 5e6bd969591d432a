// Compiles cloudini_amd/csrc/stage1_math.h (the exact per-value arithmetic the HIP kernels run) with g++ so
// that tests/test_device_math_cpu.py can diff it against the oracle without a GPU.
#include <cstdint>
#include <cstring>

#include "stage1_math.h"

using namespace cldn;

static int put(const Tok& t, uint8_t* out) {
  uint32_t w[3] = {t.w0, t.w1, t.w2};
  memcpy(out, w, 12);
  return (int)t.len;
}

extern "C" {
int m_varint32(int32_t d, uint8_t* out) { return put(varint32_tok(d), out); }
int m_varint32_len(int32_t d) { return (int)varint32_len(d); }
int m_varint64(int64_t d, uint8_t* out) { return put(varint64_tok(d), out); }
int m_varint64_len(int64_t d) { return (int)varint64_len(d); }
int m_uvarint32(uint32_t v, uint8_t* out) { return put(uvarint32_tok(v), out); }
int m_concat(int64_t a, int64_t b, uint8_t* out) {
  Tok ta = varint64_tok(a), tb = varint64_tok(b);
  if (ta.len + tb.len > 12) return -1;
  return put(tok_concat(ta, tb), out);
}
int m_groups7(uint32_t bits) { return (int)groups7(bits); }
int32_t m_quant_rne_i32(float v, float m) { return quant_rne_i32(v, m); }
int64_t m_quant_away_f32(float v, float m) { return quant_away_i64_f32(v, m); }
int64_t m_quant_away_f64(double v, double m) { return quant_away_i64_f64(v, m); }
int64_t m_int_as_i64(uint64_t raw, uint32_t type) { return int_field_as_i64(raw, type); }
uint32_t m_palette_bits(uint32_t u) { return palette_bits(u); }
}
