// stage1_decode.h -- stage-1 decode kernels (included by stage1_kernels.hip).
//
//   k_walk_chunks        one thread per cloud follows the [u32 size] chunk prefixes of its framed stream
//                        (src/cloudini.cpp:645-664) and fills the chunk table; checks the chunk count.
//   k_decode_general     one lane per chunk, a straight restatement of DecodeV4Stage1Chunk (src/v4_codec.cpp:85-117),
//                        DecodeV5Stage1Chunk (src/v5_codec.cpp:984-1012) and decodeV5AdaptiveIntSection (:764-879):
//                        complete (every codec the encoder side supports, every section mode), parallel across chunks
//                        only. It is the reference point for the parallel decoder below and the path for schemas whose
//                        token boundaries are not self-describing (raw Copy / XOR fields inside the point stream).
//   k_decode_floatn      the fast path for BASELINE configs 1/5 and the float part of 2-4: regular stream made of
//                        varint tokens only. Token ends are the bytes with a clear MSB (the NaN marker 0x00 included),
//                        so the byte offset of any point is a rank query on the end flags; every thread then parses
//                        a block of consecutive points twice (local delta sums -> block scan with NaN resets ->
//                        final values).
#pragma once

namespace cldn {

struct DecChunk {
  uint64_t src_off;   // offset of the payload inside the batch's stream buffer
  uint32_t src_size;  // payload bytes
  uint32_t n_points;
  uint64_t first_point;
  uint32_t cloud;
  uint32_t valid;
};

// grid = ceil(n_clouds / 64), one thread per cloud
__global__ void k_walk_chunks(const uint8_t* __restrict__ streams, const uint64_t* __restrict__ stream_offsets,
                              const uint64_t* __restrict__ cloud_first_point, const uint32_t* __restrict__ cloud_first_chunk,
                              uint32_t n_clouds, DecChunk* __restrict__ out, uint32_t* __restrict__ status) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_clouds) return;
  uint64_t pos = stream_offsets[k];
  const uint64_t end = stream_offsets[k + 1];
  const uint64_t n_points = cloud_first_point[k + 1] - cloud_first_point[k];
  uint64_t remaining = n_points;
  uint32_t c = cloud_first_chunk[k];
  const uint32_t c_end = cloud_first_chunk[k + 1];
  uint64_t first = cloud_first_point[k];
  bool bad = false;
  while (pos < end) {
    if (remaining == 0 || c >= c_end) { bad = true; break; }  // more chunks than declared points
    if (end - pos < 4) { bad = true; break; }
    uint32_t size = 0;
    for (int b = 0; b < 4; ++b) size |= ((uint32_t)streams[pos + b]) << (8 * b);
    pos += 4;
    if (size > end - pos) { bad = true; break; }              // "Invalid chunk size found while decoding"
    const uint32_t n = (uint32_t)(remaining < kPointsPerChunk ? remaining : kPointsPerChunk);
    DecChunk d;
    d.src_off = pos;
    d.src_size = size;
    d.n_points = n;
    d.first_point = first;
    d.cloud = k;
    d.valid = 1;
    out[c++] = d;
    pos += size;
    remaining -= n;
    first += n;
  }
  if (remaining != 0) bad = true;  // "Encoded data ended before all declared points were decoded"
  if (bad) {
    atomicOr(status, (uint32_t)ST_CORRUPT);
    for (; c < c_end; ++c) out[c].valid = 0;
    for (uint32_t q = cloud_first_chunk[k]; q < c_end; ++q) out[q].valid = 0;
  }
}

// ---- serial helpers (one lane) ------------------------------------------------------------------------------
struct Rd {
  const uint8_t* p;
  const uint8_t* end;
  bool bad;
};

// decodeVarint (include/cloudini_lib/encoding_utils.hpp:98-148); the NaN marker is rejected here
__device__ __forceinline__ int64_t rd_varint(Rd& r) {
  uint64_t uval = 0;
  uint32_t shift = 0;
  for (;;) {
    if (r.p >= r.end) { r.bad = true; return 0; }
    const uint8_t byte = *r.p++;
    const uint64_t payload = byte & 0x7fu;
    if (shift >= 64u || (shift == 63u && payload > 1u)) { r.bad = true; return 0; }
    uval |= payload << shift;
    if ((byte & 0x80u) == 0) break;
    if (shift >= 63u) { r.bad = true; return 0; }
    shift += 7u;
  }
  if (uval == 0) { r.bad = true; return 0; }
  uval--;
  return (int64_t)((uval >> 1) ^ (uint64_t)(-(int64_t)(uval & 1u)));
}

__device__ __forceinline__ uint64_t rd_uvarint(Rd& r) {  // readUVarint, src/v5_codec.cpp:176-194
  uint64_t value = 0;
  uint32_t shift = 0;
  for (;;) {
    if (r.p >= r.end) { r.bad = true; return 0; }
    const uint8_t byte = *r.p++;
    value |= ((uint64_t)(byte & 0x7fu)) << shift;
    if ((byte & 0x80u) == 0) return value;
    shift += 7u;
    if (shift >= 64u) { r.bad = true; return 0; }
  }
}

__device__ __forceinline__ uint64_t rd_raw(Rd& r, uint32_t nbytes) {
  if ((size_t)(r.end - r.p) < nbytes) { r.bad = true; return 0; }
  uint64_t v = 0;
  for (uint32_t b = 0; b < nbytes; ++b) v |= ((uint64_t)r.p[b]) << (8u * b);
  r.p += nbytes;
  return v;
}

__device__ __forceinline__ void st_raw(uint8_t* dst, uint64_t v, uint32_t nbytes) {
  for (uint32_t b = 0; b < nbytes; ++b) dst[b] = (uint8_t)(v >> (8u * b));
}

// decodeV5AdaptiveIntSection, src/v5_codec.cpp:764-879
__device__ void decode_section_serial(Rd& r, uint8_t* base, uint32_t step, uint32_t field_off, uint32_t bpv, uint32_t n) {
  if (r.p >= r.end) { r.bad = true; return; }
  const uint32_t mode = *r.p++;
  if (mode > 3u) { r.bad = true; return; }
  if (mode == 0u) {
    int64_t prev = 0;
    for (uint32_t i = 0; i < n && !r.bad; ++i) {
      prev = (int64_t)((uint64_t)prev + (uint64_t)rd_varint(r));
      if (!r.bad) st_raw(base + (size_t)i * step + field_off, (uint64_t)prev, bpv);
    }
  } else if (mode == 1u) {
    const uint32_t count = (uint32_t)rd_raw(r, 2);
    if (r.bad || count == 0u) { r.bad = true; return; }
    if ((size_t)(r.end - r.p) < (size_t)count * bpv) { r.bad = true; return; }
    const uint8_t* pal = r.p;
    r.p += (size_t)count * bpv;
    const uint32_t bits = palette_bits(count);
    const size_t index_bytes = ((size_t)bits * n + 7u) / 8u;
    if ((size_t)(r.end - r.p) < index_bytes) { r.bad = true; return; }
    const uint8_t* ip = r.p;
    uint64_t scratch = 0;
    uint32_t held = 0;
    for (uint32_t i = 0; i < n; ++i) {
      uint32_t idx = 0;
      if (bits) {
        while (held < bits) {
          scratch |= ((uint64_t)(*ip++)) << held;
          held += 8u;
        }
        idx = (uint32_t)(scratch & ((1ull << bits) - 1ull));
        scratch >>= bits;
        held -= bits;
      }
      if (idx >= count) { r.bad = true; return; }
      uint64_t v = 0;
      for (uint32_t b = 0; b < bpv; ++b) v |= ((uint64_t)pal[(size_t)idx * bpv + b]) << (8u * b);
      st_raw(base + (size_t)i * step + field_off, v, bpv);
    }
    r.p += index_bytes;
  } else {
    const uint32_t runs = (uint32_t)rd_raw(r, 4);
    uint64_t out_index = 0;
    int64_t prev = 0;
    for (uint32_t k = 0; k < runs && !r.bad; ++k) {
      uint64_t raw = 0;
      int64_t diff = 0;
      if (mode == 2u) raw = rd_raw(r, bpv);
      else diff = rd_varint(r);
      const uint64_t run_len = rd_uvarint(r);
      if (r.bad) return;
      if (out_index + run_len > n) { r.bad = true; return; }
      for (uint64_t q = 0; q < run_len; ++q) {
        if (mode == 3u) {
          prev = (int64_t)((uint64_t)prev + (uint64_t)diff);
          raw = (uint64_t)prev;
        }
        st_raw(base + (size_t)out_index * step + field_off, raw, bpv);
        ++out_index;
      }
    }
    if (!r.bad && out_index != n) r.bad = true;
  }
}

// grid = n_chunks, 64 threads, lane 0 works. `only_sections`: the regular stream was decoded by the fast kernel, which
// left the offset of the first section byte in reg_end[c].
__global__ __launch_bounds__(64) void k_decode_general(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                       const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                       uint32_t uses_v5, uint32_t only_sections,
                                                       const uint32_t* __restrict__ reg_end, uint32_t* __restrict__ status) {
  if (threadIdx.x != 0) return;
  const DecChunk dc = chunks[blockIdx.x];
  if (!dc.valid) return;
  const uint32_t step = plan.point_step;
  uint8_t* base = out + (size_t)dc.first_point * step;
  Rd r;
  r.p = streams + dc.src_off;
  r.end = r.p + dc.src_size;
  r.bad = false;
  const uint32_t n = dc.n_points;
  if (only_sections) {
    const uint32_t off = reg_end[blockIdx.x];
    if (off == 0xffffffffu || off > dc.src_size) r.bad = true;
    else r.p += off;
  } else {
    int64_t prev[kMaxOps];
    uint8_t gor_lead[kMaxOps], gor_trail[kMaxOps];
    for (uint32_t k = 0; k < plan.n_ops; ++k) {
      prev[k] = 0;
      gor_lead[k] = 255;  // kLeadingSentinel
      gor_trail[k] = 0;
    }
    for (uint32_t i = 0; i < n && !r.bad; ++i) {
      uint8_t* pt = base + (size_t)i * step;
      // "Truncated encoded data: not enough bytes for a complete point" (v4_codec.cpp:103-105)
      if (!uses_v5 && (size_t)(r.end - r.p) < plan.min_regular_bytes) { r.bad = true; break; }
      for (uint32_t k = 0; k < plan.n_ops && !r.bad; ++k) {
        const DevOp& op = plan.ops[k];
        const bool store = op.offset != 0xffffffffu;  // kDecodeButSkipStore
        switch (op.kind) {
          case OP_QF32: {  // FieldDecoderFloatN_Lossy, src/field_decoder.cpp:43-86
            if (r.p >= r.end) { r.bad = true; break; }
            float f;
            if (*r.p == 0) {
              ++r.p;
              prev[k] = 0;
              f = __uint_as_float(0x7fc00000u);
            } else {
              const int32_t q = (int32_t)((uint32_t)(int32_t)rd_varint(r) + (uint32_t)(int32_t)prev[k]);
              prev[k] = q;
              f = __fmul_rn((float)q, op.res_f);
            }
            if (store && !r.bad) st_raw(pt + op.offset, __float_as_uint(f), 4);
          } break;
          case OP_LOSSY_F32:
          case OP_LOSSY_F64: {  // FieldDecoderFloat_Lossy, include/cloudini_lib/field_decoder.hpp:330-353
            if (r.p >= r.end) { r.bad = true; break; }
            uint64_t bits;
            if (*r.p == 0) {
              ++r.p;
              prev[k] = 0;
              bits = op.kind == OP_LOSSY_F32 ? 0x7fc00000ull : 0x7ff8000000000000ull;
            } else {
              const int64_t q = (int64_t)((uint64_t)prev[k] + (uint64_t)rd_varint(r));
              prev[k] = q;
              if (op.kind == OP_LOSSY_F32) bits = __float_as_uint(__fmul_rn((float)q, op.res_f));
              else bits = (uint64_t)__double_as_longlong(__dmul_rn((double)q, op.res_d));
            }
            if (store && !r.bad) st_raw(pt + op.offset, bits, op.size);
          } break;
          case OP_INT: {  // FieldDecoderInt, field_decoder.hpp:87-97
            const int64_t q = (int64_t)((uint64_t)prev[k] + (uint64_t)rd_varint(r));
            prev[k] = q;
            if (store && !r.bad) st_raw(pt + op.offset, (uint64_t)q, op.size);
          } break;
          case OP_COPY: {
            const uint64_t v = rd_raw(r, op.size);
            if (store && !r.bad) st_raw(pt + op.offset, v, op.size);
          } break;
          case OP_XOR32:
          case OP_XOR64: {
            const uint64_t v = rd_raw(r, op.size) ^ (uint64_t)prev[k];
            prev[k] = (int64_t)v;
            if (store && !r.bad) st_raw(pt + op.offset, v, op.size);
          } break;
          case OP_GORILLA64: {  // FieldDecoderFloat_Gorilla<double>, include/cloudini_lib/field_decoder.hpp:262-305
            uint64_t value;
            if (i == 0u) {
              value = rd_raw(r, 8);  // first value of the chunk: raw bits
            } else {
              // bits are packed LSB-first and every point ends on a byte boundary: pull bytes on demand
              uint64_t lo = 0, hi = 0;
              uint32_t have = 0;
              auto need = [&](uint32_t nb) {
                while (have < nb && !r.bad) {
                  if (r.p >= r.end) { r.bad = true; break; }
                  const uint64_t byte = *r.p++;
                  if (have < 64u) {
                    lo |= byte << have;
                    if (have > 56u) hi |= byte >> (64u - have);
                  } else {
                    hi |= byte << (have - 64u);
                  }
                  have += 8u;
                }
              };
              auto take = [&](uint32_t nb) -> uint64_t {
                uint64_t v;
                if (nb >= 64u) {
                  v = lo;
                  lo = hi;
                  hi = 0;
                } else {
                  v = lo & ((1ull << nb) - 1ull);
                  if (nb) {
                    lo = (lo >> nb) | (hi << (64u - nb));
                    hi >>= nb;
                  }
                }
                have -= nb;
                return v;
              };
              need(1);
              if (r.bad) break;
              if (take(1) == 0) {
                value = (uint64_t)prev[k];
              } else {
                need(1);
                if (r.bad) break;
                uint64_t x;
                if (take(1) == 0) {
                  const uint32_t pl = gor_lead[k], pt = gor_trail[k];
                  if (pl > 64u || pl + pt >= 64u) { r.bad = true; break; }  // no window yet / corrupt
                  const uint32_t m = 64u - pl - pt;
                  need(m);
                  if (r.bad) break;
                  x = take(m) << pt;
                } else {
                  need(11);
                  if (r.bad) break;
                  const uint32_t sl = (uint32_t)take(5);
                  const uint32_t m = (uint32_t)take(6) + 1u;
                  if (sl + m > 64u) { r.bad = true; break; }
                  need(m);
                  if (r.bad) break;
                  const uint32_t tr = 64u - sl - m;
                  x = take(m) << tr;
                  gor_lead[k] = (uint8_t)sl;
                  gor_trail[k] = (uint8_t)tr;
                }
                value = x ^ (uint64_t)prev[k];
              }
            }
            prev[k] = (int64_t)value;
            if (store && !r.bad) st_raw(pt + op.offset, value, 8);
          } break;
          default:
            r.bad = true;
            break;
        }
      }
    }
  }
  if (uses_v5) {
    for (uint32_t a = 0; a < plan.n_adaptive && !r.bad; ++a)
      decode_section_serial(r, base, step, plan.adaptive[a].offset, plan.adaptive[a].bpv, n);
    if (!r.bad && r.p != r.end) r.bad = true;  // "V5 chunk has trailing bytes after decode" (v5_codec.cpp:1008-1010)
  }
  if (r.bad) atomicOr(status, (uint32_t)ST_CORRUPT);
}

}  // namespace cldn
