/* TEST INFRASTRUCTURE ONLY -- see cloudini_oracle.h for scope, citations and how parity is pinned.
 *
 * Scalar, single-threaded, deliberately simple. Every function names the reference lines it follows
 * (paths relative to /root/reference/cloudini_lib). Nothing here is used by the product path.
 */
#include "cloudini_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * varints -- include/cloudini_lib/encoding_utils.hpp
 * ---------------------------------------------------------------------------------------------- */

/* encodeVarint64, encoding_utils.hpp:55-67: zig-zag, +1 (0 is the NaN marker), LEB128 low group first. */
size_t orc_encode_varint64(int64_t value, uint8_t* out) {
  uint64_t val = ((uint64_t)value << 1) ^ (uint64_t)(value >> 63);
  val++;
  size_t n = 0;
  while (val > 0x7F) {
    out[n++] = (uint8_t)((val & 0x7F) | 0x80);
    val >>= 7;
  }
  out[n++] = (uint8_t)val;
  return n;
}

/* decodeVarint, encoding_utils.hpp:98-148 (same accept/reject behaviour, errors as negative codes). */
int orc_decode_varint(const uint8_t* buf, size_t max_size, int64_t* value) {
  if (max_size == 0) return ORC_ERR_TRUNCATED;
  uint64_t uval = 0;
  size_t count = 0;
  uint8_t shift = 0;
  for (;;) {
    if (count >= max_size) return ORC_ERR_TRUNCATED;
    const uint8_t byte = buf[count++];
    const uint8_t payload = byte & 0x7f;
    if (shift >= 64 || (shift == 63 && payload > 1)) return ORC_ERR_CORRUPT;
    uval |= ((uint64_t)payload << shift);
    if ((byte & 0x80) == 0) break;
    if (shift >= 63) return ORC_ERR_CORRUPT;
    shift = (uint8_t)(shift + 7);
  }
  if (uval == 0) return ORC_ERR_CORRUPT; /* unexpected NaN marker */
  uval--;
  *value = (int64_t)((uval >> 1) ^ (uint64_t)(-(int64_t)(uval & 1)));
  return (int)count;
}

/* appendUVarint / encodedUVarintSize, v5_codec.cpp:160-174, :244-251: plain LEB128. */
static size_t uvarint_put(uint64_t value, uint8_t* out) {
  size_t n = 0;
  while (value > 0x7Fu) {
    out[n++] = (uint8_t)((value & 0x7Fu) | 0x80u);
    value >>= 7u;
  }
  out[n++] = (uint8_t)value;
  return n;
}
static size_t uvarint_size(uint64_t value) {
  size_t bytes = 1;
  while (value > 0x7Fu) {
    value >>= 7u;
    ++bytes;
  }
  return bytes;
}
static size_t varint64_size(int64_t value) {
  uint8_t tmp[10];
  return orc_encode_varint64(value, tmp);
}

/* ------------------------------------------------------------------------------------------------
 * schema rules -- src/codec_common.cpp, src/v5_codec.cpp
 * ---------------------------------------------------------------------------------------------- */

static int size_of_type(uint8_t t) { /* basic_types.hpp:73-95 */
  switch (t) {
    case ORC_INT8: case ORC_UINT8: return 1;
    case ORC_INT16: case ORC_UINT16: return 2;
    case ORC_INT32: case ORC_UINT32: case ORC_FLOAT32: return 4;
    case ORC_FLOAT64: case ORC_INT64: case ORC_UINT64: return 8;
    default: return 0;
  }
}

static int is_adaptive_int_type(uint8_t t) { /* v5_codec.cpp:83-95 */
  return t == ORC_INT16 || t == ORC_UINT16 || t == ORC_INT32 || t == ORC_UINT32 || t == ORC_INT64 ||
         t == ORC_UINT64;
}

/* LeadingLossyFloatFieldCount, codec_common.cpp:69-82: 3 or 4 leading FLOAT32-with-resolution, else 0. */
int orc_leading_lossy_floats(const orc_schema_t* s) {
  if (s->encoding_opt != ORC_ENC_LOSSY) return 0;
  uint32_t c = 0;
  for (uint32_t i = 0; i < s->n_fields; ++i) {
    if (s->fields[i].type != ORC_FLOAT32 || !s->fields[i].has_resolution) break;
    ++c;
  }
  return (c == 3 || c == 4) ? (int)c : 0;
}

/* UsesV5Codec, v5_codec.cpp:883-892 */
int orc_uses_v5(const orc_schema_t* s) {
  if (s->version < 5 || s->encoding_opt != ORC_ENC_LOSSY) return 0;
  for (uint32_t i = (uint32_t)orc_leading_lossy_floats(s); i < s->n_fields; ++i) {
    if (is_adaptive_int_type(s->fields[i].type)) return 1;
  }
  return 0;
}

int orc_adaptive_field_count(const orc_schema_t* s) {
  if (!orc_uses_v5(s)) return 0;
  int c = 0;
  for (uint32_t i = (uint32_t)orc_leading_lossy_floats(s); i < s->n_fields; ++i) {
    if (is_adaptive_int_type(s->fields[i].type)) ++c;
  }
  return c;
}

/* MaxSerializedFieldSize / MaxSerializedPointSize, codec_common.cpp:29-67 */
uint64_t orc_max_point_bytes(const orc_schema_t* s) {
  uint64_t total = 0;
  for (uint32_t i = 0; i < s->n_fields; ++i) {
    const orc_field_t* f = &s->fields[i];
    switch (f->type) {
      case ORC_INT16: case ORC_UINT16: case ORC_INT32: case ORC_UINT32: case ORC_INT64: case ORC_UINT64:
        total += 10; break;
      case ORC_FLOAT32:
        total += (s->encoding_opt == ORC_ENC_LOSSY && f->has_resolution) ? 10 : 7; break;
      case ORC_FLOAT64:
        total += (s->encoding_opt == ORC_ENC_LOSSY && f->has_resolution) ? 10 : 11; break;
      case ORC_INT8: case ORC_UINT8:
        total += 1; break;
      default: break;
    }
  }
  return total;
}

/* MaxCompressedSize(info, n, false) with CompressionOption::NONE, cloudini.cpp:249-292 */
uint64_t orc_stage1_bound(const orc_schema_t* s, uint64_t n_points) {
  const uint64_t per_point = orc_max_point_bytes(s);
  const int v5 = orc_uses_v5(s);
  uint64_t total = 0, left = n_points;
  while (left > 0) {
    const uint64_t in_chunk = left < ORC_POINTS_PER_CHUNK ? left : ORC_POINTS_PER_CHUNK;
    left -= in_chunk;
    uint64_t chunk = in_chunk * per_point;
    if (v5) chunk += (uint64_t)s->n_fields * 32u + 1024u;
    total += 4 + chunk;
  }
  return total;
}

/* ------------------------------------------------------------------------------------------------
 * per-point ("regular") encoders -- field_encoder.hpp / field_encoder.cpp
 * ---------------------------------------------------------------------------------------------- */

enum { OP_FLOATN, OP_LOSSY_F32, OP_LOSSY_F64, OP_INT, OP_COPY, OP_XOR32, OP_XOR64, OP_GORILLA64 };

typedef struct {
  int kind;
  int lanes;           /* FLOATN: 3 or 4 */
  uint32_t offset[4];
  float mult_f[4];     /* FLOATN / LOSSY_F32 */
  double mult_d;       /* LOSSY_F64 */
  float res_f[4];      /* decode multipliers */
  double res_d;
  uint8_t type;        /* INT: field type; COPY: size via type */
  int size;            /* COPY */
  /* state */
  int32_t prev_i32[4];
  int64_t prev_i64;
  uint64_t prev_bits;
  uint8_t prev_leading, prev_trailing;
  int first;
} op_t;

static void op_reset(op_t* op) {
  memset(op->prev_i32, 0, sizeof(op->prev_i32));
  op->prev_i64 = 0;
  op->prev_bits = 0;
  op->prev_leading = 255; /* kLeadingSentinel, field_encoder.hpp:188 */
  op->prev_trailing = 0;
  op->first = 1;
}

/* cast_vector4f_to_vector4i, intrinsics.hpp:288-300, SSE4.1 branch:
 * _mm_cvtps_epi32(_mm_round_ps(x, NEAREST)): round half to even, then cvtps2dq whose out-of-range / NaN
 * result is the x86 "integer indefinite" 0x80000000. */
static int32_t cvt_rne_x86(float t) {
  const float r = nearbyintf(t); /* default FP environment: round-to-nearest-even */
  if (!(r >= -2147483648.0f && r < 2147483648.0f)) return INT32_MIN; /* also NaN */
  return (int32_t)r;
}

/* static_cast<int64_t>(float/double) as x86-64 cvttss2si/cvttsd2si behaves: indefinite = INT64_MIN. */
static int64_t cvt_i64_x86(double r) {
  if (!(r >= -9223372036854775808.0 && r < 9223372036854775808.0)) return INT64_MIN;
  return (int64_t)r;
}

/* ToInt64<T> + readIntAsI64, encoding_utils.hpp:69-73, v5_codec.cpp:97-114 */
static int64_t read_int_as_i64(const uint8_t* p, uint8_t type) {
  switch (type) {
    case ORC_INT16: { int16_t v; memcpy(&v, p, 2); return v; }
    case ORC_UINT16: { uint16_t v; memcpy(&v, p, 2); return v; }
    case ORC_INT32: { int32_t v; memcpy(&v, p, 4); return v; }
    case ORC_UINT32: { uint32_t v; memcpy(&v, p, 4); return v; }
    case ORC_INT64: { int64_t v; memcpy(&v, p, 8); return v; }
    case ORC_UINT64: { uint64_t v; memcpy(&v, p, 8); return (int64_t)v; }
    default: return 0;
  }
}
static uint64_t read_raw_bits(const uint8_t* p, int bytes) { /* v5_codec.cpp:116-120 */
  uint64_t out = 0;
  memcpy(&out, p, (size_t)bytes);
  return out;
}

/* Gorilla bit writer state for ONE encode() call: bits are emitted LSB-first and the call ends on a byte
 * boundary (field_encoder.hpp:199-306). 77 bits max -> two words. */
typedef struct { uint64_t lo, hi; int n; } bits_t;
static void bits_put(bits_t* b, uint64_t v, int nbits) {
  if (nbits < 64) v &= ((uint64_t)1 << nbits) - 1;
  if (b->n < 64) {
    b->lo |= v << b->n;
    if (b->n + nbits > 64 && b->n > 0) b->hi |= v >> (64 - b->n);
  } else {
    b->hi |= v << (b->n - 64);
  }
  b->n += nbits;
}

/* Encode one point with one op; returns bytes written. */
static size_t op_encode(op_t* op, const uint8_t* point, uint8_t* out) {
  switch (op->kind) {
    case OP_FLOATN: { /* field_encoder.cpp:42-91 */
      size_t n = 0;
      for (int i = 0; i < op->lanes; ++i) {
        float v;
        memcpy(&v, point + op->offset[i], 4);
        const float t = v * op->mult_f[i];
        const int32_t q = cvt_rne_x86(t);
        const int32_t d = (int32_t)((uint32_t)q - (uint32_t)op->prev_i32[i]); /* _mm_sub_epi32 wraps */
        op->prev_i32[i] = q;
        if (isnan(v)) {
          out[n++] = 0;
          op->prev_i32[i] = 0;
        } else {
          n += orc_encode_varint64((int64_t)d, out + n);
        }
      }
      return n;
    }
    case OP_LOSSY_F32: { /* field_encoder.hpp:342-357, FloatType = float */
      float v;
      memcpy(&v, point + op->offset[0], 4);
      if (isnan(v)) {
        out[0] = 0;
        op->prev_i64 = 0;
        return 1;
      }
      const float t = v * op->mult_f[0];
      const int64_t q = cvt_i64_x86((double)roundf(t));
      const int64_t d = (int64_t)((uint64_t)q - (uint64_t)op->prev_i64);
      op->prev_i64 = q;
      return orc_encode_varint64(d, out);
    }
    case OP_LOSSY_F64: { /* field_encoder.hpp:342-357, FloatType = double */
      double v;
      memcpy(&v, point + op->offset[0], 8);
      if (isnan(v)) {
        out[0] = 0;
        op->prev_i64 = 0;
        return 1;
      }
      const double t = v * op->mult_d;
      const int64_t q = cvt_i64_x86(round(t));
      const int64_t d = (int64_t)((uint64_t)q - (uint64_t)op->prev_i64);
      op->prev_i64 = q;
      return orc_encode_varint64(d, out);
    }
    case OP_INT: { /* field_encoder.hpp:78-85 */
      const int64_t v = read_int_as_i64(point + op->offset[0], op->type);
      const int64_t d = (int64_t)((uint64_t)v - (uint64_t)op->prev_i64);
      op->prev_i64 = v;
      return orc_encode_varint64(d, out);
    }
    case OP_COPY: /* field_encoder.hpp:56-60 */
      memcpy(out, point + op->offset[0], (size_t)op->size);
      return (size_t)op->size;
    case OP_XOR32: { /* field_encoder.hpp:359-370 */
      uint32_t cur;
      memcpy(&cur, point + op->offset[0], 4);
      const uint32_t r = cur ^ (uint32_t)op->prev_bits;
      op->prev_bits = cur;
      memcpy(out, &r, 4);
      return 4;
    }
    case OP_XOR64: {
      uint64_t cur;
      memcpy(&cur, point + op->offset[0], 8);
      const uint64_t r = cur ^ op->prev_bits;
      op->prev_bits = cur;
      memcpy(out, &r, 8);
      return 8;
    }
    case OP_GORILLA64: { /* field_encoder.hpp:246-306, IntType = uint64_t */
      uint64_t cur;
      memcpy(&cur, point + op->offset[0], 8);
      bits_t b = {0, 0, 0};
      if (op->first) {
        op->first = 0;
        op->prev_bits = cur;
        bits_put(&b, cur, 64);
      } else {
        const uint64_t x = cur ^ op->prev_bits;
        op->prev_bits = cur;
        if (x == 0) {
          bits_put(&b, 0, 1);
        } else {
          bits_put(&b, 1, 1);
          const uint8_t leading = (uint8_t)__builtin_clzll(x);
          const uint8_t trailing = (uint8_t)__builtin_ctzll(x);
          if (op->prev_leading != 255 && leading >= op->prev_leading && trailing >= op->prev_trailing) {
            bits_put(&b, 0, 1);
            const int meaningful = 64 - op->prev_leading - op->prev_trailing;
            bits_put(&b, x >> op->prev_trailing, meaningful);
          } else {
            bits_put(&b, 1, 1);
            uint8_t stored = leading > 31 ? 31 : leading;
            const int meaningful = 64 - stored - trailing;
            bits_put(&b, stored, 5);
            bits_put(&b, (uint64_t)(meaningful - 1), 6);
            bits_put(&b, x >> trailing, meaningful);
            op->prev_leading = stored;
            op->prev_trailing = trailing;
          }
        }
      }
      const size_t nbytes = (size_t)(b.n + 7) / 8;
      for (size_t k = 0; k < nbytes; ++k) {
        out[k] = (uint8_t)(k < 8 ? (b.lo >> (8 * k)) : (b.hi >> (8 * (k - 8))));
      }
      return nbytes;
    }
    default:
      return 0;
  }
}

/* Build the regular-op list. CreateCompatibleEncoder codec_common.cpp:116-153, BuildV4Encoders
 * v4_codec.cpp:26-40, buildV5Plan v5_codec.cpp:719-740. `skip_adaptive` removes the V5 adaptive ints. */
static int build_ops(const orc_schema_t* s, int skip_adaptive, op_t* ops, int max_ops) {
  int n = 0;
  if (s->encoding_opt == ORC_ENC_NONE) {
    for (uint32_t i = 0; i < s->n_fields; ++i) {
      if (n >= max_ops) return ORC_ERR_ARG;
      op_t* op = &ops[n++];
      memset(op, 0, sizeof(*op));
      op->kind = OP_COPY;
      op->offset[0] = s->fields[i].offset;
      op->size = size_of_type(s->fields[i].type);
      if (op->size == 0) return ORC_ERR_UNSUPPORTED;
    }
    return n;
  }
  const int lead = orc_leading_lossy_floats(s);
  if (lead) {
    op_t* op = &ops[n++];
    memset(op, 0, sizeof(*op));
    op->kind = OP_FLOATN;
    op->lanes = lead;
    for (int i = 0; i < lead; ++i) {
      op->offset[i] = s->fields[i].offset;
      op->mult_f[i] = 1.0F / s->fields[i].resolution; /* field_encoder.cpp:34 */
      op->res_f[i] = s->fields[i].resolution;         /* field_decoder.cpp:34 */
      if (!(op->mult_f[i] > 0.0f)) return ORC_ERR_ARG;
    }
  }
  for (uint32_t i = (uint32_t)lead; i < s->n_fields; ++i) {
    const orc_field_t* f = &s->fields[i];
    if (skip_adaptive && is_adaptive_int_type(f->type)) continue;
    if (n >= max_ops) return ORC_ERR_ARG;
    op_t* op = &ops[n++];
    memset(op, 0, sizeof(*op));
    op->offset[0] = f->offset;
    op->type = f->type;
    switch (f->type) {
      case ORC_FLOAT32:
        if (s->encoding_opt == ORC_ENC_LOSSY && f->has_resolution) {
          if (!(f->resolution > 0.0f)) return ORC_ERR_ARG;
          op->kind = OP_LOSSY_F32;
          op->mult_f[0] = (float)(1.0 / (double)f->resolution); /* field_encoder.hpp:101-102 */
          op->res_f[0] = f->resolution;
        } else if (s->encoding_opt == ORC_ENC_LOSSLESS) {
          op->kind = OP_XOR32;
        } else {
          op->kind = OP_COPY;
          op->size = 4;
        }
        break;
      case ORC_FLOAT64:
        if (s->encoding_opt == ORC_ENC_LOSSY && f->has_resolution) {
          if (!(f->resolution > 0.0f)) return ORC_ERR_ARG;
          op->kind = OP_LOSSY_F64;
          op->mult_d = 1.0 / (double)f->resolution;
          op->res_d = (double)f->resolution;
        } else if (!f->has_resolution && s->version >= 4) {
          op->kind = OP_GORILLA64;
        } else {
          op->kind = OP_XOR64;
        }
        break;
      case ORC_INT16: case ORC_UINT16: case ORC_INT32: case ORC_UINT32: case ORC_INT64: case ORC_UINT64:
        op->kind = OP_INT;
        break;
      case ORC_INT8: case ORC_UINT8:
        op->kind = OP_COPY;
        op->size = 1;
        break;
      default:
        return ORC_ERR_UNSUPPORTED;
    }
  }
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * V5 adaptive-int sections -- src/v5_codec.cpp
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
  uint32_t offset;
  uint8_t type;
  int bpv;
  int committed;
  uint8_t mode; /* 0 DeltaVarint, 1 Palette, 2 Rle, 3 DeltaRle (v5_codec.cpp:33-38) */
  int64_t* values;   /* [32768] */
  uint64_t* raw;     /* [32768] */
  uint64_t* palette; /* [32768] */
  uint32_t* indexes; /* [32768] */
} afield_t;

static int bits_for_palette_index(size_t unique_count) { /* v5_codec.cpp:196-207 */
  if (unique_count <= 1) return 0;
  int bits = 0;
  size_t m = unique_count - 1;
  while (m > 0) {
    ++bits;
    m >>= 1u;
  }
  return bits;
}

/* buildPaletteIndexes, v5_codec.cpp:369-379: palette in first-occurrence order. (The reference uses an
 * open-addressing table; the result -- order of first occurrence -- is what matters, so a sort-free
 * O(n*U) scan would do; we keep it O(n log n)-ish with a small hash of our own.) */
static size_t build_palette(afield_t* f, size_t n) {
  size_t cap = 16;
  while (cap < n * 2u) cap <<= 1u;
  uint32_t* slots = (uint32_t*)calloc(cap, sizeof(uint32_t)); /* index+1, 0 = empty */
  if (!slots) return (size_t)-1;
  size_t u = 0;
  for (size_t i = 0; i < n; ++i) {
    const uint64_t v = f->raw[i];
    uint64_t h = v;
    h ^= h >> 30u; h *= 0xbf58476d1ce4e5b9ULL; h ^= h >> 27u; h *= 0x94d049bb133111ebULL; h ^= h >> 31u;
    size_t slot = (size_t)h & (cap - 1);
    for (;;) {
      if (slots[slot] == 0) {
        f->palette[u] = v;
        slots[slot] = (uint32_t)u + 1u;
        f->indexes[i] = (uint32_t)u;
        ++u;
        break;
      }
      if (f->palette[slots[slot] - 1u] == v) {
        f->indexes[i] = slots[slot] - 1u;
        break;
      }
      slot = (slot + 1u) & (cap - 1);
    }
  }
  free(slots);
  return u;
}

/* analyzeAdaptiveIntField + selectBestAdaptiveIntMode, v5_codec.cpp:258-316, :381-412 over the first n
 * collected values. */
static int select_mode(afield_t* f, size_t n) {
  /* delta-varint */
  size_t delta = 1;
  int64_t prev = 0;
  for (size_t i = 0; i < n; ++i) {
    delta += varint64_size((int64_t)((uint64_t)f->values[i] - (uint64_t)prev));
    prev = f->values[i];
  }
  /* palette */
  const size_t u = build_palette(f, n);
  if (u == (size_t)-1) return ORC_ERR_NOMEM;
  const size_t palette = 1 + 2 + u * (size_t)f->bpv + ((size_t)bits_for_palette_index(u) * n + 7u) / 8u;
  /* rle */
  size_t rle = 1 + 4;
  for (size_t i = 0; i < n;) {
    size_t j = i + 1;
    while (j < n && f->raw[j] == f->raw[i]) ++j;
    rle += (size_t)f->bpv + uvarint_size(j - i);
    i = j;
  }
  /* delta-rle (forEachDeltaRun, v5_codec.cpp:269-288) */
  size_t drle = 1 + 4;
  prev = 0;
  for (size_t i = 0; i < n;) {
    const int64_t diff = (int64_t)((uint64_t)f->values[i] - (uint64_t)prev);
    prev = f->values[i];
    size_t j = i + 1;
    while (j < n && (int64_t)((uint64_t)f->values[j] - (uint64_t)prev) == diff) {
      prev = f->values[j];
      ++j;
    }
    drle += varint64_size(diff) + uvarint_size(j - i);
    i = j;
  }
  int mode = 0;
  size_t best = delta;
  if (palette < best) { best = palette; mode = 1; }
  if (rle < best) { best = rle; mode = 2; }
  if (drle < best) { mode = 3; }
  return mode;
}

/* Section writers, v5_codec.cpp:423-491. The streaming variants (:493-650) produce the same bytes as
 * these batch forms over the whole chunk (the probe values are replayed in order), so only the batch form
 * is restated. Returns bytes written. */
static int64_t write_section(afield_t* f, size_t n, uint8_t* out, uint64_t cap) {
  const uint64_t worst = 5u + (uint64_t)n * 21u + 8u;
  if (cap < worst) {
    /* be strict but simple: require worst case room (callers size buffers with slack) */
    uint8_t* tmp = (uint8_t*)malloc((size_t)worst);
    if (!tmp) return ORC_ERR_NOMEM;
    const int64_t r = write_section(f, n, tmp, worst);
    if (r >= 0) {
      if ((uint64_t)r > cap) { free(tmp); return ORC_ERR_CAPACITY; }
      memcpy(out, tmp, (size_t)r);
    }
    free(tmp);
    return r;
  }
  uint8_t* p = out;
  *p++ = f->mode;
  switch (f->mode) {
    case 0: { /* appendDeltaVarintSection :423-432 */
      int64_t prev = 0;
      for (size_t i = 0; i < n; ++i) {
        p += orc_encode_varint64((int64_t)((uint64_t)f->values[i] - (uint64_t)prev), p);
        prev = f->values[i];
      }
    } break;
    case 1: { /* appendPaletteSection :462-469 + appendBitpackedIndexes :209-227 */
      const size_t u = build_palette(f, n);
      if (u == (size_t)-1) return ORC_ERR_NOMEM;
      const uint16_t u16 = (uint16_t)u;
      memcpy(p, &u16, 2);
      p += 2;
      for (size_t k = 0; k < u; ++k) {
        memcpy(p, &f->palette[k], (size_t)f->bpv);
        p += f->bpv;
      }
      const int bits = bits_for_palette_index(u);
      if (bits > 0) {
        uint64_t scratch = 0;
        int held = 0;
        for (size_t i = 0; i < n; ++i) {
          scratch |= ((uint64_t)f->indexes[i] << held);
          held += bits;
          while (held >= 8) {
            *p++ = (uint8_t)(scratch & 0xFFu);
            scratch >>= 8u;
            held -= 8;
          }
        }
        if (held > 0) *p++ = (uint8_t)(scratch & 0xFFu);
      }
    } break;
    case 2: { /* appendRleSection :471-491 */
      uint8_t* count_ptr = p;
      p += 4;
      uint32_t runs = 0;
      for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && f->raw[j] == f->raw[i]) ++j;
        memcpy(p, &f->raw[i], (size_t)f->bpv);
        p += f->bpv;
        p += uvarint_put(j - i, p);
        ++runs;
        i = j;
      }
      memcpy(count_ptr, &runs, 4);
    } break;
    case 3: { /* appendDeltaRleSection :447-460 */
      uint8_t* count_ptr = p;
      p += 4;
      uint32_t runs = 0;
      int64_t prev = 0;
      for (size_t i = 0; i < n;) {
        const int64_t diff = (int64_t)((uint64_t)f->values[i] - (uint64_t)prev);
        prev = f->values[i];
        size_t j = i + 1;
        while (j < n && (int64_t)((uint64_t)f->values[j] - (uint64_t)prev) == diff) {
          prev = f->values[j];
          ++j;
        }
        p += orc_encode_varint64(diff, p);
        p += uvarint_put(j - i, p);
        ++runs;
        i = j;
      }
      memcpy(count_ptr, &runs, 4);
    } break;
    default:
      return ORC_ERR_CORRUPT;
  }
  return (int64_t)(p - out);
}

static void afields_free(afield_t* a, int n) {
  for (int i = 0; i < n; ++i) {
    free(a[i].values); free(a[i].raw); free(a[i].palette); free(a[i].indexes);
  }
}

static int afields_build(const orc_schema_t* s, afield_t* a, int max) {
  int n = 0;
  if (!orc_uses_v5(s)) return 0;
  for (uint32_t i = (uint32_t)orc_leading_lossy_floats(s); i < s->n_fields; ++i) {
    if (!is_adaptive_int_type(s->fields[i].type)) continue;
    if (n >= max) return ORC_ERR_ARG;
    afield_t* f = &a[n];
    memset(f, 0, sizeof(*f));
    f->offset = s->fields[i].offset;
    f->type = s->fields[i].type;
    f->bpv = size_of_type(f->type);
    f->values = (int64_t*)malloc(sizeof(int64_t) * ORC_POINTS_PER_CHUNK);
    f->raw = (uint64_t*)malloc(sizeof(uint64_t) * ORC_POINTS_PER_CHUNK);
    f->palette = (uint64_t*)malloc(sizeof(uint64_t) * ORC_POINTS_PER_CHUNK);
    f->indexes = (uint32_t*)malloc(sizeof(uint32_t) * ORC_POINTS_PER_CHUNK);
    ++n;
    if (!f->values || !f->raw || !f->palette || !f->indexes) {
      afields_free(a, n);
      return ORC_ERR_NOMEM;
    }
  }
  return n;
}

/* ------------------------------------------------------------------------------------------------
 * chunk loop -- EncodeV5Stage1 (v5_codec.cpp:900-963), EncodeV4Stage1Chunk (v4_codec.cpp:66-83),
 * WriteStage1Chunk (chunk_writer.cpp:27-48, NONE branch)
 * ---------------------------------------------------------------------------------------------- */

/* capacity of the op / adaptive-field tables of one call: every field yields at most one entry */
#define ORC_MAX_OPS(s) ((int)(s)->n_fields + 8)

static int64_t encode_stage1_impl(const orc_schema_t* s, const uint8_t* data, uint64_t n_points, uint8_t* out,
                                  uint64_t capacity, uint8_t* modes_out, uint32_t modes_capacity,
                                  const uint8_t* forced_modes, uint32_t n_forced);

int64_t orc_encode_stage1(const orc_schema_t* s, const uint8_t* data, uint64_t n_points, uint8_t* out,
                          uint64_t capacity, uint8_t* modes_out, uint32_t modes_capacity) {
  return encode_stage1_impl(s, data, n_points, out, capacity, modes_out, modes_capacity, NULL, 0);
}

/* A range of whole chunks of a larger cloud: the modes were committed on that cloud's first chunk
 * (v5_codec.cpp:934-949, once per encode call), everything else resets per chunk (:910-915). */
int64_t orc_encode_stage1_continued(const orc_schema_t* s, const uint8_t* data, uint64_t n_points, uint8_t* out,
                                    uint64_t capacity, const uint8_t* modes, uint32_t n_modes) {
  if (!modes && n_modes) return ORC_ERR_ARG;
  return encode_stage1_impl(s, data, n_points, out, capacity, NULL, 0, modes, n_modes);
}

static int64_t encode_stage1_impl(const orc_schema_t* s, const uint8_t* data, uint64_t n_points, uint8_t* out,
                                  uint64_t capacity, uint8_t* modes_out, uint32_t modes_capacity,
                                  const uint8_t* forced_modes, uint32_t n_forced) {
  if (!s || s->point_step == 0 || (!data && n_points) || (!out && capacity)) return ORC_ERR_ARG;
  const int v5 = orc_uses_v5(s);
  op_t* ops = (op_t*)malloc(sizeof(op_t) * (size_t)ORC_MAX_OPS(s));
  afield_t* af = (afield_t*)malloc(sizeof(afield_t) * (size_t)ORC_MAX_OPS(s));
  if (!ops || !af) { free(ops); free(af); return ORC_ERR_NOMEM; }
  int64_t result;
  const int n_ops = build_ops(s, v5, ops, ORC_MAX_OPS(s));
  if (n_ops < 0) { free(ops); free(af); return n_ops; }
  const int n_af = afields_build(s, af, ORC_MAX_OPS(s));
  if (n_af < 0) { free(ops); free(af); return n_af; }
  if (forced_modes) {
    if ((int)n_forced != n_af) { afields_free(af, n_af); free(ops); free(af); return ORC_ERR_ARG; }
    for (int a = 0; a < n_af; ++a) {
      if (forced_modes[a] > 3) { afields_free(af, n_af); free(ops); free(af); return ORC_ERR_ARG; }
      af[a].mode = forced_modes[a];
      af[a].committed = 1;
    }
  }

  uint64_t max_regular = 0; /* worst-case regular bytes per point, for the capacity check */
  for (int k = 0; k < n_ops; ++k) {
    switch (ops[k].kind) {
      case OP_FLOATN: max_regular += 5u * (uint64_t)ops[k].lanes; break;
      case OP_COPY: max_regular += (uint64_t)ops[k].size; break;
      case OP_XOR32: max_regular += 4; break;
      case OP_XOR64: max_regular += 8; break;
      case OP_GORILLA64: max_regular += 10; break;
      default: max_regular += 10; break;
    }
  }

  uint64_t written = 0, done = 0;
  while (done < n_points) {
    const uint64_t n = (n_points - done) < ORC_POINTS_PER_CHUNK ? (n_points - done) : ORC_POINTS_PER_CHUNK;
    if (capacity - written < 4 + n * max_regular) { result = ORC_ERR_CAPACITY; goto done_label; }
    uint8_t* size_ptr = out + written;
    uint8_t* p = size_ptr + 4;
    for (int k = 0; k < n_ops; ++k) op_reset(&ops[k]);
    for (uint64_t i = 0; i < n; ++i) {
      const uint8_t* point = data + (done + i) * s->point_step;
      for (int k = 0; k < n_ops; ++k) p += op_encode(&ops[k], point, p);
      for (int a = 0; a < n_af; ++a) { /* collectAdaptiveIntValue :680-688 */
        af[a].values[i] = read_int_as_i64(point + af[a].offset, af[a].type);
        af[a].raw[i] = read_raw_bits(point + af[a].offset, af[a].bpv);
      }
    }
    for (int a = 0; a < n_af; ++a) {
      if (!af[a].committed) {
        /* mode decision window, v5_codec.cpp:934-949: first 4096 values when the (first) chunk has more
         * than 4096 points, else the whole chunk; committed once per encode call. */
        const size_t window = (n > ORC_PROBE_POINTS) ? ORC_PROBE_POINTS : (size_t)n;
        const int mode = select_mode(&af[a], window);
        if (mode < 0) { result = mode; goto done_label; }
        af[a].mode = (uint8_t)mode;
        af[a].committed = 1;
      }
      const int64_t sz = write_section(&af[a], (size_t)n, p, capacity - (uint64_t)(p - out));
      if (sz < 0) { result = sz; goto done_label; }
      p += sz;
    }
    const uint32_t payload = (uint32_t)(p - size_ptr - 4);
    memcpy(size_ptr, &payload, 4);
    written += 4u + payload;
    done += n;
  }
  if (modes_out) {
    for (int a = 0; a < n_af && (uint32_t)a < modes_capacity; ++a) modes_out[a] = af[a].mode;
  }
  result = (int64_t)written;
done_label:
  afields_free(af, n_af);
  free(ops);
  free(af);
  return result;
}

/* ------------------------------------------------------------------------------------------------
 * decode -- field_decoder.{hpp,cpp}, v4_codec.cpp:85-117, v5_codec.cpp:764-879, :984-1012,
 * cloudini.cpp:635-684
 * ---------------------------------------------------------------------------------------------- */

typedef struct { const uint8_t* p; const uint8_t* end; } rd_t;

static int rd_bits(rd_t* r, uint64_t* acc_lo, uint64_t* acc_hi, int* have, int need) {
  while (*have < need) {
    if (r->p >= r->end) return ORC_ERR_TRUNCATED;
    const uint64_t byte = *r->p++;
    if (*have < 64) {
      *acc_lo |= byte << *have;
      if (*have > 56) *acc_hi |= byte >> (64 - *have);
    } else {
      *acc_hi |= byte << (*have - 64);
    }
    *have += 8;
  }
  return 0;
}
static uint64_t take_bits(uint64_t* lo, uint64_t* hi, int* have, int n) {
  uint64_t v;
  if (n == 64) {
    v = *lo;
    *lo = *hi;
    *hi = 0;
  } else {
    v = *lo & (((uint64_t)1 << n) - 1);
    if (n > 0) {
      *lo = (*lo >> n) | (*hi << (64 - n));
      *hi >>= n;
    }
  }
  *have -= n;
  return v;
}

static int op_decode(op_t* op, rd_t* r, uint8_t* point) {
  switch (op->kind) {
    case OP_FLOATN: { /* field_decoder.cpp:43-86 */
      int32_t nv[4] = {0, 0, 0, 0};
      float fv[4];
      if (r->p >= r->end) return ORC_ERR_TRUNCATED;
      for (int i = 0; i < op->lanes; ++i) {
        if (r->p >= r->end) return ORC_ERR_TRUNCATED;
        if (r->p[0] == 0) {
          nv[i] = 0;
          fv[i] = NAN;
          r->p++;
        } else {
          int64_t diff = 0;
          const int c = orc_decode_varint(r->p, (size_t)(r->end - r->p), &diff);
          if (c < 0) return c;
          nv[i] = (int32_t)((uint32_t)(int32_t)diff + (uint32_t)op->prev_i32[i]);
          fv[i] = (float)nv[i] * op->res_f[i];
          r->p += c;
        }
      }
      memcpy(op->prev_i32, nv, sizeof(nv));
      for (int i = 0; i < op->lanes; ++i) {
        if (op->offset[i] != UINT32_MAX) memcpy(point + op->offset[i], &fv[i], 4);
      }
      return 0;
    }
    case OP_LOSSY_F32: case OP_LOSSY_F64: { /* field_decoder.hpp:330-353 */
      if (r->p >= r->end) return ORC_ERR_TRUNCATED;
      if (r->p[0] == 0) {
        r->p++;
        op->prev_i64 = 0;
        if (op->offset[0] != UINT32_MAX) {
          if (op->kind == OP_LOSSY_F32) { const float nanv = NAN; memcpy(point + op->offset[0], &nanv, 4); }
          else { const double nanv = (double)NAN; memcpy(point + op->offset[0], &nanv, 8); }
        }
        return 0;
      }
      int64_t diff = 0;
      const int c = orc_decode_varint(r->p, (size_t)(r->end - r->p), &diff);
      if (c < 0) return c;
      r->p += c;
      const int64_t value = (int64_t)((uint64_t)op->prev_i64 + (uint64_t)diff);
      op->prev_i64 = value;
      if (op->offset[0] != UINT32_MAX) {
        if (op->kind == OP_LOSSY_F32) { const float f = (float)value * op->res_f[0]; memcpy(point + op->offset[0], &f, 4); }
        else { const double d = (double)value * op->res_d; memcpy(point + op->offset[0], &d, 8); }
      }
      return 0;
    }
    case OP_INT: { /* field_decoder.hpp:87-97 */
      int64_t diff = 0;
      const int c = orc_decode_varint(r->p, (size_t)(r->end - r->p), &diff);
      if (c < 0) return c;
      r->p += c;
      const int64_t value = (int64_t)((uint64_t)op->prev_i64 + (uint64_t)diff);
      op->prev_i64 = value;
      if (op->offset[0] != UINT32_MAX) memcpy(point + op->offset[0], &value, (size_t)size_of_type(op->type));
      return 0;
    }
    case OP_COPY:
      if ((size_t)(r->end - r->p) < (size_t)op->size) return ORC_ERR_TRUNCATED;
      if (op->offset[0] != UINT32_MAX) memcpy(point + op->offset[0], r->p, (size_t)op->size);
      r->p += op->size;
      return 0;
    case OP_XOR32: { /* field_decoder.hpp:355-371 */
      if ((size_t)(r->end - r->p) < 4) return ORC_ERR_TRUNCATED;
      uint32_t res;
      memcpy(&res, r->p, 4);
      r->p += 4;
      const uint32_t cur = res ^ (uint32_t)op->prev_bits;
      op->prev_bits = cur;
      if (op->offset[0] != UINT32_MAX) memcpy(point + op->offset[0], &cur, 4);
      return 0;
    }
    case OP_XOR64: {
      if ((size_t)(r->end - r->p) < 8) return ORC_ERR_TRUNCATED;
      uint64_t res;
      memcpy(&res, r->p, 8);
      r->p += 8;
      const uint64_t cur = res ^ op->prev_bits;
      op->prev_bits = cur;
      if (op->offset[0] != UINT32_MAX) memcpy(point + op->offset[0], &cur, 8);
      return 0;
    }
    case OP_GORILLA64: { /* field_decoder.hpp:262-305 */
      uint64_t lo = 0, hi = 0;
      int have = 0, e;
      uint64_t value_bits;
      if (op->first) {
        op->first = 0;
        if ((e = rd_bits(r, &lo, &hi, &have, 64)) < 0) return e;
        value_bits = take_bits(&lo, &hi, &have, 64);
        op->prev_bits = value_bits;
      } else {
        if ((e = rd_bits(r, &lo, &hi, &have, 1)) < 0) return e;
        if (take_bits(&lo, &hi, &have, 1) == 0) {
          value_bits = op->prev_bits;
        } else {
          if ((e = rd_bits(r, &lo, &hi, &have, 1)) < 0) return e;
          uint64_t x;
          if (take_bits(&lo, &hi, &have, 1) == 0) {
            const int meaningful = 64 - op->prev_leading - op->prev_trailing;
            if (meaningful <= 0 || meaningful > 64) return ORC_ERR_CORRUPT;
            if ((e = rd_bits(r, &lo, &hi, &have, meaningful)) < 0) return e;
            x = take_bits(&lo, &hi, &have, meaningful) << op->prev_trailing;
          } else {
            if ((e = rd_bits(r, &lo, &hi, &have, 11)) < 0) return e;
            const uint8_t stored = (uint8_t)take_bits(&lo, &hi, &have, 5);
            const int meaningful = (int)take_bits(&lo, &hi, &have, 6) + 1;
            if ((e = rd_bits(r, &lo, &hi, &have, meaningful)) < 0) return e;
            const uint64_t bits = take_bits(&lo, &hi, &have, meaningful);
            const int trailing = 64 - stored - meaningful;
            /* HARDENING beyond the reference: leading + meaningful > 64 makes field_decoder.hpp:283-285 shift a 64-bit word by
             * uint8_t(64 - leading - meaningful) = 192..255 bits -- undefined behaviour (x86 takes the count modulo 64 and
             * goes on with a window of "248 trailing bits"). No encoder writes such a token; every decoder of this
             * repository rejects it. (One of 998 011 damaged streams of tools/dev/oracle_vs_ref_campaign.py: the only
             * disagreement between this file and the compiled reference.) */
            if (trailing < 0) return ORC_ERR_CORRUPT;
            x = trailing >= 64 ? 0 : (bits << trailing);
            op->prev_leading = stored;
            op->prev_trailing = (uint8_t)trailing;
          }
          value_bits = x ^ op->prev_bits;
          op->prev_bits = value_bits;
        }
      }
      if (op->offset[0] != UINT32_MAX) memcpy(point + op->offset[0], &value_bits, 8);
      return 0; /* leftover padding bits of the last byte are discarded (byte aligned per call) */
    }
    default:
      return ORC_ERR_UNSUPPORTED;
  }
}

/* decodeV5AdaptiveIntSection, v5_codec.cpp:764-879 */
static int decode_section(const afield_t* f, rd_t* r, uint8_t* base, uint32_t step, size_t n) {
  if (r->p >= r->end) return ORC_ERR_TRUNCATED;
  const uint8_t mode = *r->p++;
  if (mode > 3) return ORC_ERR_CORRUPT;
  switch (mode) {
    case 0: {
      int64_t prev = 0;
      for (size_t i = 0; i < n; ++i) {
        int64_t diff = 0;
        const int c = orc_decode_varint(r->p, (size_t)(r->end - r->p), &diff);
        if (c < 0) return c;
        r->p += c;
        prev = (int64_t)((uint64_t)prev + (uint64_t)diff);
        memcpy(base + i * step + f->offset, &prev, (size_t)f->bpv);
      }
    } break;
    case 1: {
      if ((size_t)(r->end - r->p) < 2) return ORC_ERR_TRUNCATED;
      uint16_t count;
      memcpy(&count, r->p, 2);
      r->p += 2;
      if (count == 0) return ORC_ERR_CORRUPT;
      if ((size_t)(r->end - r->p) < (size_t)count * (size_t)f->bpv) return ORC_ERR_TRUNCATED;
      const uint8_t* pal = r->p;
      r->p += (size_t)count * (size_t)f->bpv;
      const int bits = bits_for_palette_index(count);
      const size_t index_bytes = ((size_t)bits * n + 7u) / 8u;
      if ((size_t)(r->end - r->p) < index_bytes) return ORC_ERR_TRUNCATED;
      const uint8_t* ip = r->p;
      uint64_t scratch = 0;
      int held = 0;
      for (size_t i = 0; i < n; ++i) {
        uint32_t idx = 0;
        if (bits) {
          while (held < bits) {
            scratch |= ((uint64_t)(*ip++) << held);
            held += 8;
          }
          idx = (uint32_t)(scratch & (((uint64_t)1 << bits) - 1u));
          scratch >>= bits;
          held -= bits;
        }
        if (idx >= count) return ORC_ERR_CORRUPT;
        memcpy(base + i * step + f->offset, pal + (size_t)idx * (size_t)f->bpv, (size_t)f->bpv);
      }
      r->p += index_bytes;
    } break;
    case 2: case 3: {
      if ((size_t)(r->end - r->p) < 4) return ORC_ERR_TRUNCATED;
      uint32_t runs;
      memcpy(&runs, r->p, 4);
      r->p += 4;
      size_t oi = 0;
      int64_t prev = 0;
      for (uint32_t k = 0; k < runs; ++k) {
        uint64_t raw = 0;
        int64_t diff = 0;
        if (mode == 2) {
          if ((size_t)(r->end - r->p) < (size_t)f->bpv) return ORC_ERR_TRUNCATED;
          memcpy(&raw, r->p, (size_t)f->bpv);
          r->p += f->bpv;
        } else {
          const int c = orc_decode_varint(r->p, (size_t)(r->end - r->p), &diff);
          if (c < 0) return c;
          r->p += c;
        }
        uint64_t run_len = 0;
        int shift = 0;
        for (;;) { /* readUVarint :176-194 */
          if (r->p >= r->end) return ORC_ERR_TRUNCATED;
          const uint8_t byte = *r->p++;
          run_len |= ((uint64_t)(byte & 0x7Fu) << shift);
          if ((byte & 0x80u) == 0) break;
          shift += 7;
          if (shift >= 64) return ORC_ERR_CORRUPT;
        }
        /* :836/:858 test `out_index + run_len > n`; a 10-byte run_len of 2^64-1 wraps that sum and the reference then
           writes out of bounds. The oracle (like the HIP decoder) compares without the addition and rejects. */
        if (run_len > n - oi) return ORC_ERR_CORRUPT;
        for (uint64_t q = 0; q < run_len; ++q) {
          if (mode == 2) {
            memcpy(base + oi * step + f->offset, &raw, (size_t)f->bpv);
          } else {
            prev = (int64_t)((uint64_t)prev + (uint64_t)diff);
            memcpy(base + oi * step + f->offset, &prev, (size_t)f->bpv);
          }
          ++oi;
        }
      }
      if (oi != n) return ORC_ERR_CORRUPT;
    } break;
  }
  return 0;
}

int64_t orc_decode_stage1(const orc_schema_t* s, const uint8_t* stream, uint64_t stream_size,
                          uint64_t n_points, uint8_t* out) {
  if (!s || s->point_step == 0) return ORC_ERR_ARG;
  const int v5 = orc_uses_v5(s);
  op_t* ops = (op_t*)malloc(sizeof(op_t) * (size_t)ORC_MAX_OPS(s));
  afield_t* af = (afield_t*)calloc((size_t)ORC_MAX_OPS(s), sizeof(afield_t));
  if (!ops || !af) { free(ops); free(af); return ORC_ERR_NOMEM; }
  int64_t result = 0;
  const int n_ops = build_ops(s, v5, ops, ORC_MAX_OPS(s));
  if (n_ops < 0) { free(ops); free(af); return n_ops; }
  int n_af = 0;
  if (v5) {
    for (uint32_t i = (uint32_t)orc_leading_lossy_floats(s); i < s->n_fields; ++i) {
      if (!is_adaptive_int_type(s->fields[i].type)) continue;
      af[n_af].offset = s->fields[i].offset;
      af[n_af].type = s->fields[i].type;
      af[n_af].bpv = size_of_type(s->fields[i].type);
      ++n_af;
    }
  }
  uint64_t pos = 0, done = 0;
  while (pos < stream_size) { /* cloudini.cpp:645-664 */
    if (done >= n_points) { result = ORC_ERR_CORRUPT; goto out_label; }
    if (stream_size - pos < 4) { result = ORC_ERR_TRUNCATED; goto out_label; }
    uint32_t chunk_size;
    memcpy(&chunk_size, stream + pos, 4);
    pos += 4;
    if (chunk_size > stream_size - pos) { result = ORC_ERR_CORRUPT; goto out_label; }
    const uint64_t n = (n_points - done) < ORC_POINTS_PER_CHUNK ? (n_points - done) : ORC_POINTS_PER_CHUNK;
    rd_t r = {stream + pos, stream + pos + chunk_size};
    uint8_t* base = out + done * s->point_step;
    for (int k = 0; k < n_ops; ++k) op_reset(&ops[k]);
    for (uint64_t i = 0; i < n; ++i) {
      for (int k = 0; k < n_ops; ++k) {
        const int e = op_decode(&ops[k], &r, base + i * s->point_step);
        if (e < 0) { result = e; goto out_label; }
      }
    }
    for (int a = 0; a < n_af; ++a) {
      const int e = decode_section(&af[a], &r, base, s->point_step, (size_t)n);
      if (e < 0) { result = e; goto out_label; }
    }
    if (v5 && r.p != r.end) { result = ORC_ERR_CORRUPT; goto out_label; } /* v5_codec.cpp:1008-1010 */
    pos += chunk_size;
    done += n;
  }
  if (done != n_points) { result = ORC_ERR_TRUNCATED; goto out_label; }
  result = (int64_t)pos;
out_label:
  free(ops);
  free(af);
  return result;
}

/* ------------------------------------------------------------------------------------------------
 * applyVizLossyPreprocessing -- src/ros_msg_utils.cpp:249-341 (voxel key: packVoxelKey21, :42-49)
 * ---------------------------------------------------------------------------------------------- */
static uint64_t viz_key(float fx, float fy, float fz, float inv_res) {
  /* static_cast<int32_t>(std::lround(f * inv_res)): float product, lround -> long, truncated to 32 bits */
  const int32_t q[3] = {(int32_t)lroundf(fx * inv_res), (int32_t)lroundf(fy * inv_res), (int32_t)lroundf(fz * inv_res)};
  uint64_t key = 0;
  for (int a = 0; a < 3; ++a) {
    const uint64_t u = (uint64_t)((int64_t)q[a] + ((int64_t)1 << 20)) & (((uint64_t)1 << 21) - 1u);
    key |= u << (21 * a);
  }
  return key;
}

int64_t orc_viz_preprocess(const uint8_t* data, uint64_t n_points, uint32_t point_step, uint32_t xyz_offset,
                           float resolution, uint8_t* out) {
  if (point_step == 0 || xyz_offset + 12u > point_step || (!data && n_points) || (!out && n_points)) return ORC_ERR_ARG;
  if (!(resolution > 0.0f) || !isfinite(resolution)) return ORC_ERR_ARG;
  const float inv_res = 1.0f / resolution;
  /* open addressing, keys + 1 so that 0 marks a free slot (keys have 63 bits) */
  uint64_t cap = 16;
  while (cap < 2 * n_points + 2) cap <<= 1;
  uint64_t* table = (uint64_t*)calloc(cap, sizeof(uint64_t));
  if (!table) return ORC_ERR_NOMEM;
  uint64_t kept = 0;
  for (uint64_t i = 0; i < n_points; ++i) {
    const uint8_t* p = data + i * point_step;
    float f[3];
    memcpy(f, p + xyz_offset, 12);
    if (!isfinite(f[0]) || !isfinite(f[1]) || !isfinite(f[2])) continue;
    const uint64_t key = viz_key(f[0], f[1], f[2], inv_res) + 1u;
    uint64_t h = (key * 0x9E3779B97F4A7C15ull) >> 17;
    int dup = 0;
    for (;;) {
      h &= cap - 1;
      if (table[h] == 0) { table[h] = key; break; }
      if (table[h] == key) { dup = 1; break; }
      ++h;
    }
    if (dup) continue;
    memcpy(out + kept * point_step, p, point_step);
    ++kept;
  }
  free(table);
  return (int64_t)kept;
}
