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

// decode statistics behind the status word (uint32 indexes into the codec's status buffer): chunks whose regular
// stream / sections went through the parallel kernels, and chunks / section sets the serial kernel had to do
constexpr uint32_t kStatFastRegular = 8, kStatFastSections = 9, kStatSerialChunks = 10, kStatSerialSections = 11;
// round 5: chunks whose sections the point kernel folded as small Palettes found from the end of the payload (not taken from
// columns): when that was every chunk of a call, the codec's next call skips the kernels that locate sections and decode them
// into columns (hip_abi.hip: dec_palette_hint)
constexpr uint32_t kStatFoldedByGuess = 12;
// round 6: chunks whose lone DeltaVarint section k_section_dv_w decoded. All of a call's chunks: the codec's next calls do not
// launch k_sections_cols_fast (it would find nothing to do); no chunk with a DeltaVarint section at all (kStatDvMode): they do
// not launch k_section_dv_w. Both are accelerators in
// front of k_decode_sections_cols, which decodes whatever is left whichever was launched (hip_abi.hip: dec_dv_hint)
constexpr uint32_t kStatDvChunks = 13;
constexpr uint32_t kStatDvMode = 14;  // chunks whose section k_locate_sections found to begin with mode byte 0 (whoever decodes it)

struct DecChunk {
  uint64_t src_off;   // offset of the payload inside the batch's stream buffer
  uint32_t src_size;  // payload bytes
  uint32_t n_points;
  uint64_t first_point;
  uint32_t cloud;
  uint32_t valid;     // 0 = do not touch; 1 = chunk of n_points points; 2 = unframed payload of a wire-version-2 stream:
                      // points until the payload is empty, n_points = the points the output buffer has room for
};

// The three per-cloud tables of a call (stream offsets, first point, first chunk; n_clouds + 1 entries each) for calls of a
// few clouds travel as a kernel argument: no upload in front of the call (round 6: one cloud per call is the ROS plugins' shape)
struct DecTablesArg {
  uint32_t n;  // 0: the tables are in device memory
  uint32_t fc[kDecInlineClouds + 1];
  uint64_t so[kDecInlineClouds + 1];
  uint64_t fp[kDecInlineClouds + 1];
};
#define DEC_TABLES(T, stream_offsets, cloud_first_point, cloud_first_chunk)                                             \
  auto so_at = [&](uint32_t i) __attribute__((always_inline)) -> uint64_t { return T.n ? T.so[i] : stream_offsets[i]; };     \
  auto fp_at = [&](uint32_t i) __attribute__((always_inline)) -> uint64_t { return T.n ? T.fp[i] : cloud_first_point[i]; };  \
  auto fc_at = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t { return T.n ? T.fc[i] : cloud_first_chunk[i]; };

// grid = ceil(n_clouds / 64), one thread per cloud
__global__ void k_walk_chunks(const uint8_t* __restrict__ streams, const uint64_t* __restrict__ stream_offsets,
                              const uint64_t* __restrict__ cloud_first_point, const uint32_t* __restrict__ cloud_first_chunk,
                              uint32_t n_clouds, DecChunk* __restrict__ out, uint32_t* __restrict__ status, const DecTablesArg T) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_clouds) return;
  DEC_TABLES(T, stream_offsets, cloud_first_point, cloud_first_chunk)
  uint64_t pos = so_at(k);
  const uint64_t end = so_at(k + 1);
  const uint64_t n_points = fp_at(k + 1) - fp_at(k);
  uint64_t remaining = n_points;
  uint32_t c = fc_at(k);
  const uint32_t c_end = fc_at(k + 1);
  uint64_t first = fp_at(k);
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
    for (uint32_t q = fc_at(k); q < c_end; ++q) out[q].valid = 0;
  }
}

// k_build_chunks: the chunk table from payload sizes the CALLER knows (the encoder's chunk_sizes output, or the prefixes
// a host caller has walked anyway): no dependent read per chunk -- one 10 M-point cloud has 306 of them, 0.45 us each
// when followed one after the other. Every size is still checked against the [u32] prefix in the stream, with the same
// rules as the walk. grid = n_clouds, 256 threads.
__global__ __launch_bounds__(256) void k_build_chunks(const uint8_t* __restrict__ streams, const uint64_t* __restrict__ stream_offsets,
                                                      const uint64_t* __restrict__ cloud_first_point,
                                                      const uint32_t* __restrict__ cloud_first_chunk,
                                                      const uint32_t* __restrict__ chunk_sizes, DecChunk* __restrict__ out,
                                                      uint32_t* __restrict__ status, const DecTablesArg T) {
  __shared__ unsigned long long wtot[4];
  __shared__ unsigned long long carry;
  __shared__ uint32_t bad_l;
  const uint32_t k = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  DEC_TABLES(T, stream_offsets, cloud_first_point, cloud_first_chunk)
  const uint64_t begin = so_at(k), end = so_at(k + 1);
  const uint64_t n_points = fp_at(k + 1) - fp_at(k);
  const uint32_t c0 = fc_at(k), c1 = fc_at(k + 1);
  if (tid == 0) {
    carry = 0ull;
    bad_l = 0u;
  }
  __syncthreads();
  for (uint32_t base = c0; base < c1; base += 256u) {
    const uint32_t c = base + tid;
    const unsigned long long mine = c < c1 ? 4ull + chunk_sizes[c] : 0ull;
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = (unsigned long long)__shfl_up((long long)incl, d);
      if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63u) wtot[wave] = incl;
    __syncthreads();
    unsigned long long before = carry;
    for (uint32_t w = 0; w < wave; ++w) before += wtot[w];
    const unsigned long long at = before + incl - mine;  // offset of my chunk's prefix inside the cloud's stream
    if (c < c1) {
      const uint64_t pos = begin + at;
      bool bad = pos + 4ull > end;
      uint32_t size = 0u;
      if (!bad) {
        for (int b = 0; b < 4; ++b) size |= ((uint32_t)streams[pos + b]) << (8 * b);
        bad = size != chunk_sizes[c] || (uint64_t)size > end - pos - 4ull;  // "Invalid chunk size found while decoding"
      }
      const uint64_t p0 = (uint64_t)(c - c0) * kPointsPerChunk;
      DecChunk d;
      d.src_off = pos + 4ull;
      d.src_size = size;
      d.n_points = (uint32_t)(n_points - p0 < kPointsPerChunk ? n_points - p0 : kPointsPerChunk);
      d.first_point = fp_at(k) + p0;
      d.cloud = k;
      d.valid = 1u;
      out[c] = d;
      if (bad) bad_l = 1u;
    }
    __syncthreads();
    if (tid == 255u) carry = before + incl;
    __syncthreads();
  }
  // the chunks must fill the cloud's stream exactly (more chunks than points / data that ends early are size errors here)
  if (tid == 0 && carry != end - begin) bad_l = 1u;
  __syncthreads();
  if (bad_l) {
    if (tid == 0) atomicOr(status, (uint32_t)ST_CORRUPT);
    for (uint32_t c = c0 + tid; c < c1; c += 256u) out[c].valid = 0u;
  }
}

// ---- serial helpers (one lane) ------------------------------------------------------------------------------
// The payload is read through a window in LDS: the one lane that parses would otherwise wait for a global load per BYTE
// (0.5-1 us each: 16 us per point of a layout with a Gorilla-coded stamp); a refill is 128 independent 16-byte loads.
constexpr uint32_t kRdWindow = 2048;
struct Rd {
  const uint8_t* p;
  const uint8_t* end;
  bool bad;
  uint8_t* win = nullptr;          // LDS, kRdWindow + 16 bytes, 16-byte aligned; nullptr = read global memory directly
  const uint8_t* wbase = nullptr;  // global address of win[0]
  uint32_t wlen = 0u;
};

// (plain arguments, no reference to the reader: a reader whose address is taken lives in scratch memory, and every use
// of its members becomes a memory round trip)
__device__ __noinline__ uint32_t rd_refill(uint8_t* win, const uint8_t* b, const uint8_t* end) {
  const size_t left = (size_t)(end - b);  // (unaligned 16-byte loads from b on: nothing in front of it is touched)
  const uint32_t len = (uint32_t)(left < (size_t)kRdWindow ? left : (size_t)kRdWindow);
  const uint32_t full = len >> 4;
  for (uint32_t i = 0; i < full; ++i) {
    uint4 v;
    __builtin_memcpy(&v, b + 16u * i, 16);
    reinterpret_cast<uint4*>(win)[i] = v;
  }
  for (uint32_t k = full * 16u; k < len; ++k) win[k] = b[k];
  return len;
}

// the byte at q (p <= q < end)
__device__ __forceinline__ uint8_t rd_at(Rd& r, const uint8_t* q) {
  if (r.win == nullptr) return *q;
  if (q < r.wbase || q >= r.wbase + r.wlen) {
    r.wlen = rd_refill(r.win, q, r.end);
    r.wbase = q;
  }
  return r.win[q - r.wbase];
}

// decodeVarint (include/cloudini_lib/encoding_utils.hpp:98-148); the NaN marker is rejected here
__device__ __forceinline__ int64_t rd_varint(Rd& r) {
  uint64_t uval = 0;
  uint32_t shift = 0;
  for (;;) {
    if (r.p >= r.end) { r.bad = true; return 0; }
    const uint8_t byte = rd_at(r, r.p);
    ++r.p;
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
    const uint8_t byte = rd_at(r, r.p);
    ++r.p;
    value |= ((uint64_t)(byte & 0x7fu)) << shift;
    if ((byte & 0x80u) == 0) return value;
    shift += 7u;
    if (shift >= 64u) { r.bad = true; return 0; }
  }
}

__device__ __forceinline__ uint64_t rd_raw(Rd& r, uint32_t nbytes) {
  if ((size_t)(r.end - r.p) < nbytes) { r.bad = true; return 0; }
  uint64_t v = 0;
  for (uint32_t b = 0; b < nbytes; ++b) v |= ((uint64_t)rd_at(r, r.p + b)) << (8u * b);
  r.p += nbytes;
  return v;
}

// the low nbytes of v at any address: the field sizes are single stores (gfx950 takes unaligned stores of 2, 4 and 8
// bytes: a memcpy compiles to one global_store_short / dword / dwordx2), anything else goes byte by byte
__device__ __forceinline__ void st_raw(uint8_t* dst, uint64_t v, uint32_t nbytes) {
  if (nbytes == 4u) {
    const uint32_t w = (uint32_t)v;
    __builtin_memcpy(dst, &w, 4);
  } else if (nbytes == 2u) {
    const uint16_t h = (uint16_t)v;
    __builtin_memcpy(dst, &h, 2);
  } else if (nbytes == 8u) {
    __builtin_memcpy(dst, &v, 8);
  } else {
    for (uint32_t b = 0; b < nbytes; ++b) dst[b] = (uint8_t)(v >> (8u * b));
  }
}

// decodeV5AdaptiveIntSection, src/v5_codec.cpp:764-879
__device__ void decode_section_serial(Rd& r, uint8_t* base, uint32_t step, uint32_t field_off, uint32_t bpv, uint32_t n) {
  if (r.p >= r.end) { r.bad = true; return; }
  const uint32_t mode = rd_at(r, r.p);
  ++r.p;
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
      if (run_len > (uint64_t)n - out_index) { r.bad = true; return; }  // no addition: run_len may be 2^64-1
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
// WIDE (round 5): PLAN = WidePlan -- a schema beyond the launch-argument plan (stage1_wide.h). Its ops are read from device
// memory and the per-op state lives in `wide_state` (n_ops * 16 bytes per chunk: [int64 prev][u8 lead][u8 trail]) instead of
// the fixed LDS arrays.
template <bool WIDE, class PLAN>
__device__ __forceinline__ void decode_general_body_t(const PLAN& plan, const uint8_t* __restrict__ streams,
                                                         const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                         uint32_t uses_v5, uint32_t only_sections,
                                                         const uint32_t* __restrict__ reg_end,
                                                         const uint8_t* __restrict__ sec_done, uint32_t* __restrict__ status,
                                                         uint8_t* wide_state) {
  if (threadIdx.x != 0) return;
  const DecChunk dc = chunks[blockIdx.x];
  if (!dc.valid) return;
  if (sec_done && sec_done[blockIdx.x]) return;  // regular stream and sections were decoded by the parallel kernels
  const uint32_t step = plan.point_step;
  uint8_t* base = out + (size_t)dc.first_point * step;
  __shared__ __attribute__((aligned(16))) uint8_t rd_window[kRdWindow + 16u];
  // the ops of the plan in LDS: an op read from the kernel-argument segment at a run-time index is a memory round trip
  // per member, several per token
  __shared__ DevOp ops_l[WIDE ? 1 : kMaxOps];
  if (!WIDE)
    for (uint32_t k = 0; k < plan.n_ops; ++k) ops_l[k] = plan.ops[k];
  Rd r;
  r.p = streams + dc.src_off;
  r.end = r.p + dc.src_size;
  r.bad = false;
  r.win = rd_window;
  const uint32_t n = dc.n_points;
  bool regular_done = false;
  if (only_sections) {
    const uint32_t off = reg_end[blockIdx.x];
    if (off != 0xfffffffeu) {  // kDecRedo: the fast kernel gave up on this chunk
      if (off > dc.src_size) r.bad = true;
      else r.p += off;
      regular_done = true;
      if (!uses_v5) return;    // V4 wire: nothing behind the regular stream
    }
  }
  atomicAdd(&status[regular_done ? kStatSerialSections : kStatSerialChunks], 1u);
  if (!regular_done) {
    // (per-op state in LDS: a 64-entry array indexed at run time would live in scratch, one memory round trip per access)
    __shared__ int64_t prev_l[WIDE ? 1 : kMaxOps];
    __shared__ uint8_t gor_lead_l[WIDE ? 1 : kMaxOps], gor_trail_l[WIDE ? 1 : kMaxOps];
    uint8_t* ws = WIDE ? wide_state + (size_t)blockIdx.x * plan.n_ops * 16u : nullptr;
    int64_t* prev = WIDE ? reinterpret_cast<int64_t*>(ws) : prev_l;
    uint8_t* gor_lead = WIDE ? ws + (size_t)plan.n_ops * 8u : gor_lead_l;
    uint8_t* gor_trail = WIDE ? ws + (size_t)plan.n_ops * 9u : gor_trail_l;
    for (uint32_t k = 0; k < plan.n_ops; ++k) {
      prev[k] = 0;
      gor_lead[k] = 255;  // kLeadingSentinel
      gor_trail[k] = 0;
    }
    // wire version 2 (src/cloudini.cpp:665-667, src/v4_codec.cpp:108-115): no chunks, no point count -- points until the
    // payload is empty; one more point than the output holds is "Output buffer is too small to hold the decoded data"
    const bool unframed = dc.valid == 2u;
    for (uint32_t i = 0; (unframed ? r.p < r.end : i < n) && !r.bad; ++i) {
      if (unframed && i >= n) { r.bad = true; break; }
      uint8_t* pt = base + (size_t)i * step;
      // "Truncated encoded data: not enough bytes for a complete point" (v4_codec.cpp:103-105)
      if (!uses_v5 && (size_t)(r.end - r.p) < plan.min_regular_bytes) { r.bad = true; break; }
      for (uint32_t k = 0; k < plan.n_ops && !r.bad; ++k) {
        const DevOp& op = WIDE ? plan.ops[k] : ops_l[k];
        const bool store = op.offset != 0xffffffffu;  // kDecodeButSkipStore
        switch (op.kind) {
          case OP_QF32: {  // FieldDecoderFloatN_Lossy, src/field_decoder.cpp:43-86
            if (r.p >= r.end) { r.bad = true; break; }
            float f;
            if (rd_at(r, r.p) == 0) {
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
            if (rd_at(r, r.p) == 0) {
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
                  const uint64_t byte = rd_at(r, r.p);
                  ++r.p;
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

__device__ __forceinline__ void decode_general_body(const DevPlan& plan, const uint8_t* __restrict__ streams,
                                                       const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                       uint32_t uses_v5, uint32_t only_sections,
                                                       const uint32_t* __restrict__ reg_end,
                                                       const uint8_t* __restrict__ sec_done, uint32_t* __restrict__ status) {
  decode_general_body_t<false>(plan, streams, chunks, out, uses_v5, only_sections, reg_end, sec_done, status, nullptr);
}

// the WIDE route's decoder: grid = n_chunks, 64 threads, lane 0 works (whole chunks, regular stream and sections)
__global__ __launch_bounds__(64) void k_decode_wide(const WidePlan plan, const uint8_t* __restrict__ streams,
                                                    const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                    uint32_t uses_v5, uint32_t* __restrict__ status, uint8_t* wide_state) {
  decode_general_body_t<true>(plan, streams, chunks, out, uses_v5, 0u, nullptr, nullptr, status, wide_state);
}

__global__ __launch_bounds__(64) void k_decode_general(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                       const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                       uint32_t uses_v5, uint32_t only_sections,
                                                       const uint32_t* __restrict__ reg_end,
                                                       const uint8_t* __restrict__ sec_done, uint32_t* __restrict__ status) {
  decode_general_body(plan, streams, chunks, out, uses_v5, only_sections, reg_end, sec_done, status);
}


// ---------------------------------------------------------------------------------------------------------------
// Parallel decode of varint token streams.
//
//   dv_tokens_tile    one tile = 8 bytes per thread. End-of-token flags (MSB clear; the NaN marker 0x00 included) ->
//                     one block scan gives every thread the index of its first token; the tokens ending inside a
//                     thread's 8 bytes are consecutive, and the thread rebuilds each from the 8-byte window that ends
//                     at the token's last byte (tokens of more than 7 bytes flag the chunk for the serial decoder; the
//                     64-bit instantiation takes up to 10 from the history bytes: a chunk's first epoch stamp).
//   dv_stream         regular stream / DeltaVarint section: token values staged in LDS in token order; every thread
//                     then takes a run of whole points: local sums per op with NaN resets -> segmented block scan ->
//                     second walk that adds the incoming value, converts and stores (FieldDecoderFloatN_Lossy::decode,
//                     src/field_decoder.cpp:43-86; FieldDecoderFloat_Lossy / FieldDecoderInt,
//                     include/cloudini_lib/field_decoder.hpp:330-353, :79-106). Tokens of a point cut by the tile
//                     edge wait at the front of the LDS buffer for the next tile.
//   k_decode_varint   dv_stream over the regular stream of plans whose regular ops are all varint-coded (<= 8 per point)
//   k_decode_sections the V5 adaptive-int sections of a chunk, field after field (decodeV5AdaptiveIntSection,
//                     src/v5_codec.cpp:764-879): DeltaVarint = dv_stream with one integer op; Palette = 32 indexes per
//                     thread; DeltaRle = (diff, run) token pairs -> run table in LDS (two block scans) -> every output
//                     finds its run by binary search; Rle = one lane parses the runs from an LDS copy of the bytes,
//                     then the same fill.
//
// reg_end[c] = offset of the first byte behind the regular stream, or kDecRedo when anything looked irregular;
// sec_done[c] = 1 when all sections were decoded here. k_decode_general redoes whatever is left (whole chunks or
// their sections), so all error reporting stays with the restatement of the reference's checks.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kDecRedo = 0xfffffffeu;
constexpr int kDvThreads = 1024;
constexpr uint32_t kDvTileBytes = kDvThreads * 8u;

// tokens ending in this thread's 8 bytes of the tile at payload offset `pos`. tileb: [16 history bytes + tile].
// store(kl, x, end_off): kl = index behind the `left` waiting tokens, x = the token's 7-bit groups joined (< 2^49),
// end_off = payload offset behind the token. Tokens with a global index >= `limit` are left alone. Returns the
// number of tokens ending in the tile (block-uniform). Contains barriers.
template <bool LONG_TOKENS = false, typename Store>
__device__ __forceinline__ uint32_t dv_tokens_tile(const uint8_t* __restrict__ src, uint32_t src_size, uint32_t pos,
                                                   uint32_t* tileb, uint32_t* misc, uint32_t left, uint32_t seen,
                                                   uint32_t limit, Store store) {
  constexpr int T = kDvThreads;
  const uint32_t tid = threadIdx.x;
  const uint32_t my = pos + tid * 8u;
  uint32_t b0 = 0xffffffffu, b1 = 0xffffffffu;  // past the end: continuation bytes, no token ends
  if (my + 8u <= src_size) {
    const uint8_t* q = src + my;
    if ((((uintptr_t)q) & 3u) == 0u) {
      b0 = reinterpret_cast<const uint32_t*>(q)[0];
      b1 = reinterpret_cast<const uint32_t*>(q)[1];
    } else {
      b0 = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
      b1 = (uint32_t)q[4] | ((uint32_t)q[5] << 8) | ((uint32_t)q[6] << 16) | ((uint32_t)q[7] << 24);
    }
  } else if (my < src_size) {
    uint64_t w = ~0ull;
    for (uint32_t k = 0; k < src_size - my; ++k) w = (w & ~(0xffull << (8u * k))) | ((uint64_t)src[my + k] << (8u * k));
    b0 = (uint32_t)w;
    b1 = (uint32_t)(w >> 32);
  }
  tileb[4u + tid * 2u] = b0;
  tileb[5u + tid * 2u] = b1;
  const uint32_t e0 = ~b0 & 0x80808080u, e1 = ~b1 & 0x80808080u;
  // bit j of `ends` = byte j ends a token (the multiply gathers bits 0, 8, 16, 24 into 21..24)
  const uint32_t ends = (((e0 >> 7) * 0x00204081u) >> 21 & 0xfu) | ((((e1 >> 7) * 0x00204081u) >> 21 & 0xfu) << 4);
  uint32_t n_tile;
  const uint32_t tb = block_exclusive_scan<T>((uint32_t)__builtin_popcount(ends), misc + 2, &n_tile);  // has a barrier
  for (uint32_t m = ends, r = 0u; m; m &= m - 1u, ++r) {
    const uint32_t j = (uint32_t)__builtin_ctz(m);
    const uint32_t kl = left + tb + r;
    if (seen + kl >= limit) break;                   // bytes of whatever follows the token stream
    const uint32_t e = 16u + tid * 8u + j;           // byte index of the token's last byte in tileb
    const uint32_t w0 = e - 7u;                      // window [w0, e]
    const uint32_t d0 = tileb[w0 >> 2], d1 = tileb[(w0 >> 2) + 1u], d2 = tileb[(w0 >> 2) + 2u];
    const uint32_t sh = (w0 & 3u) * 8u;
    const uint32_t lo = sh ? ((d0 >> sh) | (d1 << (32u - sh))) : d0;
    const uint32_t hi = sh ? ((d1 >> sh) | (d2 << (32u - sh))) : d1;
    // continuation bytes right before the end byte (window bytes 6, 5, ...)
    const uint32_t c_lo = lo & 0x80808080u, c_hi = hi & 0x00808080u;
    const uint32_t cm = (((c_lo >> 7) * 0x00204081u) >> 21 & 0xfu) | ((((c_hi >> 7) * 0x00204081u) >> 21 & 0x7u) << 4);
    const uint32_t lencont = (uint32_t)__builtin_clz(~(cm << 25));  // leading ones of the 7-bit mask
    if (lencont >= 7u && !LONG_TOKENS) {
      misc[0] = 1u;  // token of 8 or more bytes
      continue;
    }
    uint64_t x = ((((uint64_t)hi) << 32) | lo) >> (8u * (7u - min(lencont, 7u)));
    x &= 0x7f7f7f7f7f7f7f7full;
    x = (x & 0x007f007f007f007full) | ((x & 0x7f007f007f007f00ull) >> 1);
    x = (x & 0x00003fff00003fffull) | ((x & 0x3fff00003fff0000ull) >> 2);
    x = (x & 0x000000000fffffffull) | ((x & 0x0fffffff00000000ull) >> 4);
    if (LONG_TOKENS && lencont >= 7u) {
      // The window is the token's LAST 8 bytes; an int64 varint has up to 10 (the first value of a chunk's epoch time
      // stamp at 1 us has 8): up to two more bytes in front of the window, inside the 16 bytes of history. Arithmetic
      // modulo 2^64 like decodeVarint's shifts; the tenth byte may carry one bit (rd_varint's check).
      const uint32_t p8 = e - 8u, p9 = e - 9u, p10 = e - 10u;
      const uint32_t B8 = (tileb[p8 >> 2] >> ((p8 & 3u) * 8u)) & 0xffu;
      const uint32_t B9 = (tileb[p9 >> 2] >> ((p9 & 3u) * 8u)) & 0xffu;
      const uint32_t B10 = (tileb[p10 >> 2] >> ((p10 & 3u) * 8u)) & 0xffu;
      if ((B8 & 0x80u) == 0u) {
        // exactly 8 bytes: x is complete
      } else if ((B9 & 0x80u) == 0u) {
        x = (uint64_t)(B8 & 0x7fu) | (x << 7);
      } else if ((B10 & 0x80u) == 0u && ((hi >> 24) & 0x7fu) <= 1u) {
        x = (uint64_t)(B9 & 0x7fu) | ((uint64_t)(B8 & 0x7fu) << 7) | (x << 14);
      } else {
        misc[0] = 1u;  // more than 10 bytes, or bits beyond 64: the serial decoder raises the error
        continue;
      }
    }
    if (x == 0ull && lencont != 0u) {
      // an overlong zero (0x80 0x00, ...): decodeVarint (encoding_utils.hpp:139-141) rejects every zero it is handed -- only the
      // single byte 0x00 in front of it is the NaN marker (src/field_decoder.cpp:58-62). The serial decoder raises the error.
      misc[0] = 1u;
      continue;
    }
    store(kl, x, pos + tid * 8u + j + 1u);
  }
  __syncthreads();
  return n_tile;
}

// after a tile: the last 16 bytes become the history of the next one (barrier included)
__device__ __forceinline__ void dv_roll_history(uint32_t* tileb) {
  uint32_t hist = 0u;
  if (threadIdx.x < 4u) hist = tileb[4u + (uint32_t)kDvThreads * 2u - 4u + threadIdx.x];
  __syncthreads();
  if (threadIdx.x < 4u) tileb[threadIdx.x] = hist;
}

template <int NOPS, bool WIDE>
struct DvLds {
  using Acc = typename std::conditional<WIDE, long long, int>::type;
  static constexpr uint32_t kBytesOff = 0;                                  // [16 history + tile] bytes
  static constexpr uint32_t kValOff = 16u + kDvTileBytes + 16u;             // Acc [8 + tile tokens]
  static constexpr uint32_t kMarkOff = kValOff + (8u + kDvTileBytes) * (uint32_t)sizeof(Acc);  // u8 [8 + tile tokens]
  static constexpr uint32_t kScanOff = (kMarkOff + 8u + kDvTileBytes + 15u) & ~15u;  // per wave: Acc[NOPS] + flags
  static constexpr uint32_t kWaveRec = NOPS * (uint32_t)sizeof(Acc) + 8u;
  static constexpr uint32_t kMiscOff = kScanOff + 17u * kWaveRec;           // 16 waves + carry record
  static constexpr uint32_t kTotal = kMiscOff + 256u;  // misc: [0] bad, [1] end offset, [2..34) block-scan scratch
};

// One token stream of n_points x n_ops varint tokens that starts at src + start. Results: misc[0] != 0 -> irregular,
// misc[1] = payload offset behind the last token. op_at(o) -> const DevOp&. Ends with a barrier.
template <int NOPS, bool WIDE, typename OpAt>
__device__ __forceinline__ void dv_stream(OpAt op_at, uint32_t n_ops, const uint8_t* __restrict__ src,
                                          uint32_t src_size, uint32_t start, uint32_t n_points, uint8_t* base,
                                          uint32_t step, uint8_t* smem) {
  using L = DvLds<NOPS, WIDE>;
  using Acc = typename L::Acc;
  using UAcc = typename std::make_unsigned<Acc>::type;
  constexpr int T = kDvThreads;
  uint32_t* tileb = reinterpret_cast<uint32_t*>(smem + L::kBytesOff);
  Acc* val = reinterpret_cast<Acc*>(smem + L::kValOff);
  uint8_t* mark = smem + L::kMarkOff;
  uint8_t* scanrec = smem + L::kScanOff;
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + L::kMiscOff);
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t target = n_points * n_ops;

  __syncthreads();  // the caller may have used the LDS
  if (tid < 4u) tileb[tid] = 0u;  // history before the stream: token ends
  if (tid == 0) {
    misc[0] = 0u;
    misc[1] = target == 0u ? start : 0xffffffffu;
  }
  if (tid < (uint32_t)(NOPS * sizeof(Acc) + 8u) / 4u)
    reinterpret_cast<uint32_t*>(scanrec + 16u * L::kWaveRec)[tid] = 0u;  // carry record: sums 0
  __syncthreads();

  uint32_t pos = start;     // payload offset of the current tile
  uint32_t pts_done = 0u;   // points already written
  uint32_t left = 0u;       // tokens of a cut point waiting at val[0..left)
  while (pts_done * n_ops + left < target) {
    if (pos >= src_size) {  // stream ends before all tokens
      if (tid == 0) misc[0] = 1u;
      break;
    }
    const uint32_t seen = pts_done * n_ops;  // tokens handed to points so far
    const uint32_t n_tile = dv_tokens_tile<WIDE>(src, src_size, pos, tileb, misc, left, seen, target,
                                           [&](uint32_t kl, uint64_t x, uint32_t end_off) {
                                             const bool marker = (x == 0ull);
                                             const uint64_t u1 = x - 1ull;
                                             const uint64_t d = (u1 >> 1) ^ (0ull - (u1 & 1ull));
                                             val[kl] = marker ? (Acc)0 : (Acc)(UAcc)d;
                                             mark[kl] = marker ? 1u : 0u;
                                             if (seen + kl + 1u == target) misc[1] = end_off;
                                           });

    // ---- whole points of this tile
    const uint32_t avail = min(left + n_tile, target - seen);
    const uint32_t npts = avail / n_ops;
    const uint32_t per = (npts + T - 1u) / T;
    const uint32_t p0 = min(npts, tid * per), p1 = min(npts, p0 + per);
    Acc acc[NOPS];
    uint32_t fl = 0u;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) acc[o] = 0;
    for (uint32_t i = p0; i < p1; ++i) {
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        if ((uint32_t)o < n_ops) {
          const uint32_t idx = i * n_ops + (uint32_t)o;
          if (mark[idx]) {
            acc[o] = 0;
            fl |= 1u << o;
            if (op_at(o).kind == OP_INT) misc[0] = 1u;  // the marker is not a valid integer token
          } else {
            acc[o] = (Acc)((UAcc)acc[o] + (UAcc)val[idx]);
          }
        }
      }
    }
    // segmented inclusive scan over the threads: (f1, v1) o (f2, v2) = (f1 | f2, f2 ? v2 : v1 + v2)
    Acc inc[NOPS];
    uint32_t fin = fl;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) inc[o] = acc[o];
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const uint32_t of = (uint32_t)__shfl_up((int)fin, dlt);
      Acc ov[NOPS];
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        if (WIDE) ov[o] = (Acc)__shfl_up((long long)inc[o], dlt);
        else ov[o] = (Acc)__shfl_up((int)inc[o], dlt);
      }
      if (lane >= (uint32_t)dlt) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o)
          if (!(fin & (1u << o))) inc[o] = (Acc)((UAcc)inc[o] + (UAcc)ov[o]);
        fin |= of;
      }
    }
    if (lane == 63u) {
      Acc* rec = reinterpret_cast<Acc*>(scanrec + wave * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) rec[o] = inc[o];
      *reinterpret_cast<uint32_t*>(scanrec + wave * L::kWaveRec + NOPS * sizeof(Acc)) = fin;
    }
    __syncthreads();
    // incoming state of this thread = carry o waves before o lanes before
    Acc in[NOPS];
    {
      const Acc* crec = reinterpret_cast<const Acc*>(scanrec + 16u * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) in[o] = crec[o];
      for (uint32_t w = 0; w < wave; ++w) {
        const Acc* rec = reinterpret_cast<const Acc*>(scanrec + w * L::kWaveRec);
        const uint32_t rf = *reinterpret_cast<const uint32_t*>(scanrec + w * L::kWaveRec + NOPS * sizeof(Acc));
#pragma unroll
        for (int o = 0; o < NOPS; ++o) in[o] = (rf & (1u << o)) ? rec[o] : (Acc)((UAcc)in[o] + (UAcc)rec[o]);
      }
      // exclusive within the wave: the inclusive state of the previous lane
      const uint32_t pf = (uint32_t)__shfl_up((int)fin, 1);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        Acc pv;
        if (WIDE) pv = (Acc)__shfl_up((long long)inc[o], 1);
        else pv = (Acc)__shfl_up((int)inc[o], 1);
        if (lane > 0u) in[o] = (pf & (1u << o)) ? pv : (Acc)((UAcc)in[o] + (UAcc)pv);
      }
    }
    // second walk: final values
    for (uint32_t i = p0; i < p1; ++i) {
      uint8_t* pt = base + (size_t)(pts_done + i) * step;
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        if ((uint32_t)o < n_ops) {
          const DevOp& op = op_at(o);
          const uint32_t idx = i * n_ops + (uint32_t)o;
          const bool isnan = mark[idx] != 0u;
          in[o] = isnan ? (Acc)0 : (Acc)((UAcc)in[o] + (UAcc)val[idx]);
          if (op.offset != 0xffffffffu) {
            if (op.kind == OP_QF32) {
              const uint32_t bits = isnan ? 0x7fc00000u : __float_as_uint(__fmul_rn((float)(int32_t)in[o], op.res_f));
              if ((op.offset & 3u) == 0u && (step & 3u) == 0u) *reinterpret_cast<uint32_t*>(pt + op.offset) = bits;
              else st_raw(pt + op.offset, bits, 4);
            } else if (op.kind == OP_LOSSY_F32) {
              const uint32_t bits = isnan ? 0x7fc00000u : __float_as_uint(__fmul_rn((float)(long long)in[o], op.res_f));
              st_raw(pt + op.offset, bits, 4);
            } else if (op.kind == OP_LOSSY_F64) {
              const uint64_t bits = isnan ? 0x7ff8000000000000ull
                                          : (uint64_t)__double_as_longlong(__dmul_rn((double)(long long)in[o], op.res_d));
              st_raw(pt + op.offset, bits, 8);
            } else {
              st_raw(pt + op.offset, (uint64_t)(long long)in[o], op.size);
            }
          }
        }
      }
    }
    // block state after this tile -> carry; the cut point's tokens and the last 16 bytes move to the front
    const uint32_t used = npts * n_ops;
    const uint32_t rest = avail - used;  // < n_ops <= 8
    Acc mv = 0;
    uint8_t mm = 0;
    if (tid < rest) {
      mv = val[used + tid];
      mm = mark[used + tid];
    }
    dv_roll_history(tileb);  // barrier inside: every thread is done with val / mark / scanrec
    if (tid == (uint32_t)T - 1u) {  // the last thread's inclusive state is the block total
      Acc* crec = reinterpret_cast<Acc*>(scanrec + 16u * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) crec[o] = in[o];
    }
    if (tid < rest) {
      val[tid] = mv;
      mark[tid] = mm;
    }
    __syncthreads();
    pos += kDvTileBytes;
    pts_done += npts;
    left = rest;
    if (misc[0]) break;  // uniform after the barrier
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// dv_stream2: the same job as dv_stream without staging token values in LDS. The tokens that end inside a thread's
// BPT bytes are consecutive tokens of the stream, so the thread that decodes them also owns them: it keeps their
// values in registers (indexed by the byte position of the token's last byte), folds them into one partial sum per
// op (token g belongs to op g % n_ops), the segmented block scan hands it the running values in front of its first
// token, and a second walk over the same registers produces and stores the final values -- consecutive tokens are
// consecutive fields of consecutive points, so the stores of neighbouring threads are neighbours too. Nothing
// waits at a tile edge: a point cut by it simply continues with the carried per-op state.
// ---------------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------------
// k_mark_token_ends: regular streams with raw fields between the varints (FieldEncoderCopy, field_encoder.hpp:342-357:
// XYZ + a packed rgb float without resolution is the PCL PointXYZRGB layout). A raw byte may look like anything, so the
// ends of the varints cannot be read off the MSBs; what is known is the FORM of a point -- op after op, a varint or
// `size` raw bytes. Tiles of 1 KiB that start at a point boundary:
//   J[0][p] = where the point that would start at byte p ends (every p of the tile, four per thread),
//   J[k][p] = J[k-1][J[k-1][p]]                                   (pointer doubling, 10 levels),
//   point t of the tile starts where the set bits of t lead from byte 0 through the J[k] -- every thread finds its
//   point in 10 reads, walks it once more and sets the end bits of its tokens in an LDS bitmap,
//   the tile's bits leave as whole 32-bit words (the partial last word is carried into the next tile).
// The bitmap of chunk c begins at word token_ends_word(src_off, c): one bit per payload byte. A malformed or truncated
// stream sets reg_end[c] = kDecRedo: the serial decoder takes the chunk and raises the error.
// grid = n_chunks, 256 threads.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kMtThreads = 256;
constexpr uint32_t kMtTile = 1024;
constexpr uint32_t kMtLook = 256;      // a point has at most this many bytes
constexpr uint32_t kMtLevels = 10;     // 2^10 points per tile at most (a point has at least one byte per op)
constexpr uint32_t kMtMaxOps = 8;

__host__ __device__ __forceinline__ uint64_t token_ends_word(uint64_t src_off, uint32_t c) { return (src_off >> 5) + c; }

// end (exclusive) of the point that starts at byte p of the staged bytes, 0xffff when it is malformed or incomplete
__device__ __forceinline__ uint32_t mt_point_end(const uint8_t* bytes, uint32_t avail, uint32_t p, uint32_t n_ops, const uint8_t* raw_size) {
  uint32_t q = p;
  for (uint32_t o = 0; o < n_ops; ++o) {
    const uint32_t rs = raw_size[o];
    if (rs != 0u) {
      q += rs;
    } else {
      uint32_t len = 0u;
      for (uint32_t k = 0; k < 10u; ++k) {
        if (q + k >= avail) return 0xffffu;
        if ((bytes[q + k] & 0x80u) == 0u) {
          len = k + 1u;
          break;
        }
      }
      if (len == 0u) return 0xffffu;  // more than 10 bytes
      q += len;
    }
    if (q > avail) return 0xffffu;
  }
  return q;
}

__global__ __launch_bounds__(kMtThreads) void k_mark_token_ends(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                const DecChunk* __restrict__ chunks,
                                                                uint32_t* __restrict__ token_ends, uint32_t* __restrict__ reg_end) {
  __shared__ __attribute__((aligned(16))) uint8_t bytes[kMtTile + kMtLook + 16u];
  __shared__ uint16_t J[kMtLevels][kMtTile];
  __shared__ uint32_t bm[(kMtTile + kMtLook) / 32u + 2u];
  __shared__ uint8_t raw_size[kMtMaxOps];
  __shared__ uint32_t sh[4];  // [0] bad, [1] points of the tile, [2] bytes consumed, [3] carried partial word
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const DecChunk dc = chunks[c];
  if (!dc.valid) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  const uint32_t n_ops = plan.n_ops;
  uint32_t* out_words = token_ends + token_ends_word(dc.src_off, c);
  if (tid < kMtMaxOps) {
    const uint32_t kd = tid < n_ops ? plan.ops[tid].kind : 0xffu;
    raw_size[tid] = (kd == OP_COPY || kd == OP_XOR32 || kd == OP_XOR64) ? plan.ops[tid].size : (uint8_t)0;
  }
  if (tid == 0) {
    sh[0] = 0u;
    sh[3] = 0u;
    reg_end[c] = 0u;
  }
  __syncthreads();
  {
    // every op raw (EncodingOptions::NONE, lossless floats): a point has a fixed size and the ends follow from the byte
    // index alone -- the threads write the bitmap's words directly
    uint32_t fixed = 0u;
    bool all_raw = true;
    for (uint32_t o = 0; o < n_ops; ++o) {
      all_raw = all_raw && raw_size[o] != 0u;
      fixed += raw_size[o];
    }
    if (all_raw) {  // uniform
      if ((uint64_t)fixed * n > src_size) {
        if (tid == 0) reg_end[c] = kDecRedo;  // truncated: the serial decoder raises the error
        return;
      }
      const uint32_t used = fixed * n;
      for (uint32_t w = tid; w < (used + 31u) / 32u; w += kMtThreads) {
        uint32_t r = (w * 32u) % fixed, bits = 0u;
        for (uint32_t b = 0; b < 32u && w * 32u + b < used; ++b) {
          uint32_t acc = 0u;  // r + 1 == an op's end offset inside the point?
          bool is_end = false;
          for (uint32_t o = 0; o < n_ops; ++o) {
            acc += raw_size[o];
            is_end = is_end || r + 1u == acc;
          }
          if (is_end) bits |= 1u << b;
          r = r + 1u == fixed ? 0u : r + 1u;
        }
        out_words[w] = bits;
      }
      return;
    }
  }
  // levels of the doubling: 2^levels > the most points a tile can hold (the plan knows a point's fewest bytes)
  uint32_t levels = 1u;
  while (levels < kMtLevels && (1u << levels) <= kMtTile / max(1u, plan.min_regular_bytes)) ++levels;
  uint32_t ps = 0u, pts_done = 0u;  // uniform
  while (pts_done < n) {
    if (ps >= src_size) {
      if (tid == 0) sh[0] = 1u;
      break;
    }
    const uint32_t avail = min(kMtTile + kMtLook, src_size - ps);
    for (uint32_t i = tid; i < (kMtTile + kMtLook) / 4u; i += kMtThreads) {
      uint32_t w = 0u;
      const uint32_t o = i * 4u;
      if (o + 4u <= avail) __builtin_memcpy(&w, src + ps + o, 4);
      else
        for (uint32_t bb = 0; bb < 4u && o + bb < avail; ++bb) w |= (uint32_t)src[ps + o + bb] << (8u * bb);
      reinterpret_cast<uint32_t*>(bytes)[i] = w;
    }
    for (uint32_t i = tid; i < (kMtTile + kMtLook) / 32u + 2u; i += kMtThreads) bm[i] = 0u;
    __syncthreads();
    for (uint32_t p = tid; p < kMtTile; p += kMtThreads) J[0][p] = (uint16_t)(p < avail ? mt_point_end(bytes, avail, p, n_ops, raw_size) : 0xffffu);
    __syncthreads();
    for (uint32_t k = 1; k < levels; ++k) {
      for (uint32_t p = tid; p < kMtTile; p += kMtThreads) {
        const uint32_t j = J[k - 1u][p];
        J[k][p] = j < kMtTile ? J[k - 1u][j] : (uint16_t)j;
      }
      __syncthreads();
    }
    if (tid == 0) {  // how many points start inside the tile: the longest walk from byte 0 that stays inside
      uint32_t p = 0u, t = 0u;
      for (int k = (int)levels - 1; k >= 0; --k) {
        const uint32_t q = J[k][p];
        if (q < kMtTile) {
          p = q;
          t += 1u << k;
        }
      }
      sh[1] = min(t + 1u, n - pts_done);
    }
    __syncthreads();
    const uint32_t want = sh[1];
    for (uint32_t t = tid; t < want; t += kMtThreads) {
      uint32_t p = 0u;
      for (uint32_t k = 0; k < levels; ++k)
        if ((t >> k) & 1u) p = J[k][p];
      const uint32_t end = J[0][p];
      if (end == 0xffffu) {
        sh[0] = 1u;  // malformed, or the payload ends inside the point
      } else {
        uint32_t q = p;
        for (uint32_t o = 0; o < n_ops; ++o) {
          const uint32_t rs = raw_size[o];
          if (rs != 0u) {
            q += rs;
          } else {
            while (bytes[q] & 0x80u) ++q;  // (mt_point_end has checked the token)
            ++q;
          }
          atomicOr(&bm[(q - 1u) >> 5], 1u << ((q - 1u) & 31u));
        }
        if (t + 1u == want) sh[2] = end;
      }
    }
    __syncthreads();
    if (sh[0]) break;  // uniform
    const uint32_t consumed = sh[2];
    {  // bits [0, consumed) of the tile = bits [ps, ps + consumed) of the chunk
      const uint32_t s = ps & 31u, w0 = ps >> 5;
      const uint32_t total_bits = s + consumed;
      const uint32_t n_full = total_bits >> 5;  // whole words; the rest is carried
      const uint32_t carry = sh[3];
      __syncthreads();
      for (uint32_t i = tid; i <= n_full; i += kMtThreads) {
        uint32_t v = s ? ((bm[i] << s) | (i ? (bm[i - 1u] >> (32u - s)) : 0u)) : bm[i];
        if (i == 0u) v |= carry;
        if (i < n_full) out_words[w0 + i] = v;
        else sh[3] = (total_bits & 31u) ? v : 0u;  // the partial last word waits for the next tile
      }
    }
    ps += consumed;
    pts_done += want;
    __syncthreads();
  }
  __syncthreads();
  if (tid == 0) {
    if (sh[0]) reg_end[c] = kDecRedo;
    else if (ps & 31u) out_words[ps >> 5] = sh[3];
  }
}

template <int NOPS, bool WIDE, int BPT>
struct Dv2Lds {
  using Acc = typename std::conditional<WIDE, long long, int>::type;
  static constexpr uint32_t kTileBytes = kDvThreads * BPT;
  static constexpr uint32_t kBytesOff = 0;                                 // [16 history + tile + 16] bytes
  static constexpr uint32_t kScanOff = 16u + kTileBytes + 16u;            // per wave: Acc[NOPS] + flags; [16] = carry
  static constexpr uint32_t kWaveRec = NOPS * (uint32_t)sizeof(Acc) + 8u;
  static constexpr uint32_t kMiscOff = (kScanOff + 17u * kWaveRec + 15u) & ~15u;
  static constexpr uint32_t kTotal = kMiscOff + 256u;  // misc: [0] bad, [1] end offset, [2..34) block-scan scratch
};

// end_bits != NULL (WIDE, BPT = 8): the token ends are not read off the bytes' MSBs but from a bitmap (bit p = payload
// byte p ends a token) that k_mark_token_ends has laid out -- streams with raw (FieldEncoderCopy) fields between the
// varints. A raw field is a token like any other there: its value is its bytes, it resets its op's running sum.
template <int NOPS, bool WIDE, int BPT, typename OpAt>
__device__ __forceinline__ void dv_stream2(OpAt op_at, uint32_t n_ops, const uint8_t* __restrict__ src,
                                           uint32_t src_size, uint32_t start, uint32_t n_points, uint8_t* base,
                                           uint32_t step, uint8_t* smem, const uint8_t* __restrict__ end_bits = nullptr) {
  using L = Dv2Lds<NOPS, WIDE, BPT>;
  using Acc = typename L::Acc;
  using UAcc = typename std::make_unsigned<Acc>::type;
  constexpr int T = kDvThreads;
  constexpr int DW = BPT / 4;
  static_assert(BPT == 8 || BPT == 16, "8 or 16 bytes per thread");
  uint32_t* tileb = reinterpret_cast<uint32_t*>(smem + L::kBytesOff);
  uint8_t* scanrec = smem + L::kScanOff;
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + L::kMiscOff);
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t target = n_points * n_ops;

  __syncthreads();  // the caller may have used the LDS
  if (tid < 4u) tileb[tid] = 0u;  // history before the stream: token ends
  if (tid == 0) {
    misc[0] = 0u;
    misc[1] = target == 0u ? start : 0xffffffffu;
  }
  if (tid < (uint32_t)(NOPS * sizeof(Acc) + 8u) / 4u)
    reinterpret_cast<uint32_t*>(scanrec + 16u * L::kWaveRec)[tid] = 0u;  // carry record: sums 0
  __syncthreads();

  uint32_t pos = start;  // payload offset of the current tile
  uint32_t seen = 0u;    // tokens decoded so far
  while (seen < target) {
    if (pos >= src_size) {  // stream ends before all tokens
      if (tid == 0) misc[0] = 1u;
      break;
    }
    // ---- bytes, end flags, token indexes
    const uint32_t my = pos + tid * BPT;
    uint32_t b[DW];
#pragma unroll
    for (int k = 0; k < DW; ++k) b[k] = 0xffffffffu;  // past the end: continuation bytes, no token ends
    if (my + BPT <= src_size) {
      const uint8_t* q = src + my;
      if ((((uintptr_t)q) & 3u) == 0u) {
#pragma unroll
        for (int k = 0; k < DW; ++k) b[k] = reinterpret_cast<const uint32_t*>(q)[k];
      } else {
#pragma unroll
        for (int k = 0; k < DW; ++k)
          b[k] = (uint32_t)q[4 * k] | ((uint32_t)q[4 * k + 1] << 8) | ((uint32_t)q[4 * k + 2] << 16) |
                 ((uint32_t)q[4 * k + 3] << 24);
      }
    } else if (my < src_size) {
#pragma unroll
      for (int k = 0; k < DW; ++k) {
        uint32_t w = 0xffffffffu;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb)
          if (my + 4u * k + bb < src_size) w = (w & ~(0xffu << (8 * bb))) | ((uint32_t)src[my + 4u * k + bb] << (8 * bb));
        b[k] = w;
      }
    }
    uint32_t ends = 0u;  // bit j = byte j ends a token
#pragma unroll
    for (int k = 0; k < DW; ++k) {
      tileb[4u + tid * DW + k] = b[k];
      ends |= ((((~b[k] & 0x80808080u) >> 7) * 0x00204081u) >> 21 & 0xfu) << (4 * k);
    }
    uint32_t ehist = 0u;  // bitmap mode: bit k = byte my - 16 + k ends a token (k < 16: in front of my bytes; the byte in front of the payload counts as an end)
    if (WIDE && BPT == 8 && end_bits != nullptr) {
      ends = my < src_size ? (uint32_t)end_bits[my >> 3] : 0u;  // (my is a multiple of 8)
      const uint32_t g8 = my >> 3;
      const uint32_t h1 = g8 >= 1u ? (uint32_t)end_bits[g8 - 1u] : 0x80u;                  // bytes my-8 .. my-1
      const uint32_t h0 = g8 >= 2u ? (uint32_t)end_bits[g8 - 2u] : (g8 == 1u ? 0x80u : 0u);  // bytes my-16 .. my-9
      ehist = h0 | (h1 << 8) | (ends << 16);
    }
    uint32_t n_tile;
    const uint32_t tb = block_exclusive_scan<T>((uint32_t)__builtin_popcount(ends), misc + 2, &n_tile);  // barrier inside
    const uint32_t g0 = seen + tb;  // stream index of my first token

    // ---- my tokens -> registers (indexed by the byte position of the token's last byte)
    Acc d[BPT];
    uint32_t valid = 0u, marks = 0u, raws = 0u, xors = 0u;
    const uint32_t p0 = g0 / n_ops, o0 = g0 - p0 * n_ops;  // point and op of my first token
    {
      uint32_t g = g0;
      uint32_t og = o0;
#pragma unroll
      for (int j = 0; j < BPT; ++j) {
        d[j] = 0;
        if ((ends >> j) & 1u) {
          if (g < target) {
            const uint32_t e = 16u + tid * BPT + (uint32_t)j;  // byte index of the token's last byte in tileb
            const uint32_t w0 = e - 7u;                        // window [w0, e]
            const uint32_t d0 = tileb[w0 >> 2], d1 = tileb[(w0 >> 2) + 1u], d2 = tileb[(w0 >> 2) + 2u];
            const uint32_t sh = (w0 & 3u) * 8u;
            const uint32_t lo = sh ? ((d0 >> sh) | (d1 << (32u - sh))) : d0;
            const uint32_t hi = sh ? ((d1 >> sh) | (d2 << (32u - sh))) : d1;
            const uint32_t c_lo = lo & 0x80808080u, c_hi = hi & 0x00808080u;
            const uint32_t cm = (((c_lo >> 7) * 0x00204081u) >> 21 & 0xfu) | ((((c_hi >> 7) * 0x00204081u) >> 21 & 0x7u) << 4);
            uint32_t lencont = (uint32_t)__builtin_clz(~(cm << 25));  // continuation bytes before the end byte
            bool from_bits = false;
            if (WIDE && BPT == 8 && end_bits != nullptr) {
              // with raw fields in the stream the bytes in front of a token say nothing about it: its length is the
              // distance to the end in front of it (bit 16 + j of `ehist` is this token's end)
              const uint32_t P = 16u + (uint32_t)j;
              const uint32_t below = ehist & ((1u << P) - 1u);
              const uint32_t tl = below ? P - (31u - (uint32_t)__builtin_clz(below)) : 99u;
              if (tl > 10u) misc[0] = 1u;
              lencont = min(tl, 10u) - 1u;
              from_bits = true;
            }
            if (lencont >= 7u && !WIDE) misc[0] = 1u;                        // token of 8 or more bytes
            uint64_t x = ((((uint64_t)hi) << 32) | lo) >> (8u * (7u - min(lencont, WIDE ? 7u : 6u)));
            x &= 0x7f7f7f7f7f7f7f7full;
            x = (x & 0x007f007f007f007full) | ((x & 0x7f007f007f007f00ull) >> 1);
            x = (x & 0x00003fff00003fffull) | ((x & 0x3fff00003fff0000ull) >> 2);
            x = (x & 0x000000000fffffffull) | ((x & 0x0fffffff00000000ull) >> 4);
            if (WIDE && lencont >= 7u) {
              // The window is the token's LAST 8 bytes; an int64 varint has up to 10 (the first value of a chunk's epoch
              // time stamp at 1 us has 8): up to two more bytes in front of the window, inside the 16 bytes of history.
              // Arithmetic modulo 2^64 like decodeVarint's shifts; the tenth byte may carry one bit (rd_varint's check).
              const uint32_t p8 = e - 8u, p9 = e - 9u, p10 = e - 10u;
              const uint32_t B8 = (tileb[p8 >> 2] >> ((p8 & 3u) * 8u)) & 0xffu;
              const uint32_t B9 = (tileb[p9 >> 2] >> ((p9 & 3u) * 8u)) & 0xffu;
              const uint32_t B10 = (tileb[p10 >> 2] >> ((p10 & 3u) * 8u)) & 0xffu;
              const bool is8 = from_bits ? lencont == 7u : (B8 & 0x80u) == 0u;
              const bool is9 = from_bits ? lencont == 8u : (B9 & 0x80u) == 0u;
              const bool is10 = from_bits ? lencont == 9u : (B10 & 0x80u) == 0u;
              if (is8) {
                // exactly 8 bytes: x is complete
              } else if (is9) {
                x = (uint64_t)(B8 & 0x7fu) | (x << 7);
              } else if (is10 && ((hi >> 24) & 0x7fu) <= 1u) {
                x = (uint64_t)(B9 & 0x7fu) | ((uint64_t)(B8 & 0x7fu) << 7) | (x << 14);
              } else {
                misc[0] = 1u;  // more than 10 bytes, or bits beyond 64: the serial decoder raises the error
              }
            }
            const uint64_t u1 = x - 1ull;
            d[j] = (Acc)(UAcc)((u1 >> 1) ^ (0ull - (u1 & 1ull)));
            valid |= 1u << j;
            if (x == 0ull) marks |= 1u << j;
            if (WIDE && end_bits != nullptr) {  // a raw field: the last `size` bytes of the window are its value
              uint32_t kind = 0u, size = 0u;
#pragma unroll
              for (int oo = 0; oo < NOPS; ++oo)
                if (og == (uint32_t)oo) {
                  kind = op_at(oo).kind;
                  size = op_at(oo).size;
                }
              if (kind == OP_COPY || kind == OP_XOR32 || kind == OP_XOR64) {
                const uint64_t win = (((uint64_t)hi) << 32) | lo;
                d[j] = (Acc)(UAcc)(size >= 8u ? win : (win >> (64u - 8u * size)));
                if (kind == OP_COPY) raws |= 1u << j;
                else xors |= 1u << j;  // FieldDecoderFloat_XOR (field_decoder.hpp): the value is the bytes XOR the value before
                marks &= ~(1u << j);
              }
            }
            // an overlong zero (a varint token of two or more bytes whose value bits are all 0) is no NaN marker: decodeVarint
            // rejects it (encoding_utils.hpp:139-141; the marker is the single byte 0x00, src/field_decoder.cpp:58-62)
            if (((marks >> j) & 1u) && lencont != 0u) misc[0] = 1u;
            if (g + 1u == target) misc[1] = pos + tid * BPT + (uint32_t)j + 1u;
          }
          ++g;
          og = (og + 1u == n_ops) ? 0u : og + 1u;
        }
      }
    }

    // ---- partial sums per op with NaN resets
    Acc acc[NOPS];
    uint32_t fl = 0u;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) acc[o] = 0;
    {
      uint32_t o = o0;
#pragma unroll
      for (int j = 0; j < BPT; ++j) {
        if ((valid >> j) & 1u) {
          const bool mk = (marks >> j) & 1u;
          const bool rw = (raws >> j) & 1u;
          const bool xr = (xors >> j) & 1u;
#pragma unroll
          for (int oo = 0; oo < NOPS; ++oo) {
            if (o == (uint32_t)oo) {
              acc[oo] = xr ? (Acc)((UAcc)acc[oo] ^ (UAcc)d[j]) : (rw ? d[j] : (mk ? (Acc)0 : (Acc)((UAcc)acc[oo] + (UAcc)d[j])));
              if (rw) fl |= 1u << oo;
              if (mk) {
                fl |= 1u << oo;
                if (op_at(oo).kind == OP_INT) misc[0] = 1u;  // the marker is not a valid integer token
              }
            }
          }
          o = (o + 1u == n_ops) ? 0u : o + 1u;
        }
      }
    }
    // segmented inclusive scan over the threads: (f1, v1) o (f2, v2) = (f1 | f2, f2 ? v2 : v1 + v2)
    // (XOR-coded ops combine with ^ instead of +; they never reset)
    uint32_t xmask = 0u;
    if (WIDE && end_bits != nullptr) {
#pragma unroll
      for (int o = 0; o < NOPS; ++o)
        if ((uint32_t)o < n_ops && (op_at(o).kind == OP_XOR32 || op_at(o).kind == OP_XOR64)) xmask |= 1u << o;
    }
    auto comb = [&](int o, Acc a, Acc b) -> Acc { return ((xmask >> o) & 1u) ? (Acc)((UAcc)a ^ (UAcc)b) : (Acc)((UAcc)a + (UAcc)b); };
    Acc inc[NOPS];
    uint32_t fin = fl;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) inc[o] = acc[o];
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) {
      const uint32_t of = (uint32_t)__shfl_up((int)fin, dlt);
      Acc ov[NOPS];
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        if (WIDE) ov[o] = (Acc)__shfl_up((long long)inc[o], dlt);
        else ov[o] = (Acc)__shfl_up((int)inc[o], dlt);
      }
      if (lane >= (uint32_t)dlt) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o)
          if (!(fin & (1u << o))) inc[o] = comb(o, inc[o], ov[o]);
        fin |= of;
      }
    }
    if (lane == 63u) {
      Acc* rec = reinterpret_cast<Acc*>(scanrec + wave * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) rec[o] = inc[o];
      *reinterpret_cast<uint32_t*>(scanrec + wave * L::kWaveRec + NOPS * sizeof(Acc)) = fin;
    }
    __syncthreads();
    // incoming state of this thread = carry o waves before o lanes before
    Acc in[NOPS];
    {
      const Acc* crec = reinterpret_cast<const Acc*>(scanrec + 16u * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) in[o] = crec[o];
      for (uint32_t w = 0; w < wave; ++w) {
        const Acc* rec = reinterpret_cast<const Acc*>(scanrec + w * L::kWaveRec);
        const uint32_t rf = *reinterpret_cast<const uint32_t*>(scanrec + w * L::kWaveRec + NOPS * sizeof(Acc));
#pragma unroll
        for (int o = 0; o < NOPS; ++o) in[o] = (rf & (1u << o)) ? rec[o] : comb(o, in[o], rec[o]);
      }
      const uint32_t pf = (uint32_t)__shfl_up((int)fin, 1);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        Acc pv;
        if (WIDE) pv = (Acc)__shfl_up((long long)inc[o], 1);
        else pv = (Acc)__shfl_up((int)inc[o], 1);
        if (lane > 0u) in[o] = (pf & (1u << o)) ? pv : comb(o, in[o], pv);
      }
    }
    // ---- second walk: final values, converted and stored
    {
      uint32_t o = o0;
      uint8_t* pt = base + (size_t)p0 * step;
#pragma unroll
      for (int j = 0; j < BPT; ++j) {
        if ((valid >> j) & 1u) {
          const bool mk = (marks >> j) & 1u;
          const bool rw = (raws >> j) & 1u;
          const bool xr = (xors >> j) & 1u;
          Acc cur = 0;
          uint32_t kind = 0u, size = 0u, off = 0xffffffffu;
          float resf = 0.0f;
          double resd = 0.0;
#pragma unroll
          for (int oo = 0; oo < NOPS; ++oo) {
            if (o == (uint32_t)oo) {
              in[oo] = xr ? (Acc)((UAcc)in[oo] ^ (UAcc)d[j]) : (rw ? d[j] : (mk ? (Acc)0 : (Acc)((UAcc)in[oo] + (UAcc)d[j])));
              cur = in[oo];
              const DevOp& op = op_at(oo);
              kind = op.kind;
              size = op.size;
              off = op.offset;
              resf = op.res_f;
              if (WIDE) resd = op.res_d;
            }
          }
          if (off != 0xffffffffu) {
            if (!WIDE || kind == OP_QF32) {
              const uint32_t bits = mk ? 0x7fc00000u : __float_as_uint(__fmul_rn((float)(int32_t)cur, resf));
              if (((off | step) & 3u) == 0u) *reinterpret_cast<uint32_t*>(pt + off) = bits;
              else st_raw(pt + off, bits, 4);
            } else if (kind == OP_LOSSY_F32) {
              st_raw(pt + off, mk ? 0x7fc00000u : __float_as_uint(__fmul_rn((float)(long long)cur, resf)), 4);
            } else if (kind == OP_LOSSY_F64) {
              st_raw(pt + off, mk ? 0x7ff8000000000000ull : (uint64_t)__double_as_longlong(__dmul_rn((double)(long long)cur, resd)), 8);
            } else {
              st_raw(pt + off, (uint64_t)(long long)cur, size);
            }
          }
          if (o + 1u == n_ops) {
            o = 0u;
            pt += step;
          } else {
            ++o;
          }
        }
      }
    }
    // block state after this tile -> carry; the last 16 bytes become the next tile's history
    uint32_t hist = 0u;
    if (tid < 4u) hist = tileb[4u + (uint32_t)T * DW - 4u + tid];
    __syncthreads();  // every thread has read the carry record and its window bytes
    if (tid == (uint32_t)T - 1u) {  // the last thread's running values are the block total
      Acc* crec = reinterpret_cast<Acc*>(scanrec + 16u * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) crec[o] = in[o];
    }
    if (tid < 4u) tileb[tid] = hist;
    __syncthreads();
    seen += min(n_tile, target - seen);
    pos += L::kTileBytes;
    if (misc[0]) break;  // uniform after the barrier
  }
  __syncthreads();
}

template <int NOPS, bool WIDE>
__device__ __forceinline__ void decode_varint_body(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                              const DecChunk* __restrict__ chunks,
                                                              uint8_t* __restrict__ out, uint32_t* __restrict__ reg_end,
                                                              uint32_t* __restrict__ status, uint32_t redo_only,
                                                              const uint32_t* __restrict__ token_ends = nullptr) {
  constexpr int BPT = WIDE ? 8 : 16;
  using L = Dv2Lds<NOPS, WIDE, BPT>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t c = blockIdx.x;
  const DecChunk dc = chunks[c];
  if (!dc.valid) return;
  if (redo_only && reg_end[c] != kDecRedo) return;  // behind k_decode_points: only the chunks it handed back
  uint8_t* base = out + (size_t)dc.first_point * plan.point_step;
  const uint8_t* end_bits = nullptr;
  if (token_ends != nullptr) {
    if (reg_end[c] == kDecRedo) return;  // k_mark_token_ends found the stream irregular: the serial decoder takes the chunk
    end_bits = reinterpret_cast<const uint8_t*>(token_ends + token_ends_word(dc.src_off, c));
  }
  dv_stream2<NOPS, WIDE, BPT>([&](int o) -> const DevOp& { return plan.ops[o]; }, plan.n_ops, streams + dc.src_off,
                              dc.src_size, 0u, dc.n_points, base, plan.point_step, smem, end_bits);
  const uint32_t* misc = reinterpret_cast<const uint32_t*>(smem + L::kMiscOff);
  if (threadIdx.x == 0) {
    const bool redo = misc[0] || misc[1] == 0xffffffffu;
    reg_end[c] = redo ? kDecRedo : misc[1];
    if (!redo) atomicAdd(&status[kStatFastRegular], 1u);
  }
}

template <int NOPS, bool WIDE>
__global__ __launch_bounds__(kDvThreads) __attribute__((amdgpu_waves_per_eu(WIDE ? 4 : 8, 8))) void k_decode_varint(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                              const DecChunk* __restrict__ chunks,
                                                              uint8_t* __restrict__ out, uint32_t* __restrict__ reg_end,
                                                              uint32_t* __restrict__ status, uint32_t redo_only,
                                                              const uint32_t* __restrict__ token_ends) {
  decode_varint_body<NOPS, WIDE>(plan, streams, chunks, out, reg_end, status, redo_only, token_ends);
}


// ---- sections -------------------------------------------------------------------------------------------------
constexpr uint32_t kRunTile = 4096;     // DeltaRle: runs per table tile (8192 tokens)
constexpr uint32_t kRleRunTile = 2048;  // Rle: runs parsed per round
constexpr uint32_t kRleStage = 16384;   // Rle: section bytes staged per round
struct DecSecLds {  // the DvLds<1, true> layout (DeltaVarint uses it as is) + a run-start table behind it
  using DL = DvLds<1, true>;
  static constexpr uint32_t kRawOff = DL::kValOff;                      // u64 [8 + 8192]: token values, then run table
  static constexpr uint32_t kStageOff = kRawOff + 2u * kRleRunTile * 8u;  // Rle: staged bytes behind its run table
  static constexpr uint32_t kStartOff = DL::kTotal;                     // u32 [kRunTile + 4]
  static constexpr uint32_t kTotal = kStartOff + (kRunTile + 4u) * 4u;
  static_assert(kStageOff + kRleStage + 16u <= DL::kMarkOff + 8u + kDvTileBytes, "Rle staging fits the token area");
};

template <int T>
__device__ __forceinline__ uint64_t block_exclusive_scan_u64(uint64_t x, uint64_t* wtot, uint64_t* total) {
  constexpr int NW = T / 64;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint64_t incl = x;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const uint64_t o = (uint64_t)__shfl_up((long long)incl, dlt);
    if (lane >= (uint32_t)dlt) incl += o;
  }
  if (lane == 63u) wtot[wave] = incl;
  __syncthreads();
  uint64_t basev = 0u, tot = 0u;
  for (uint32_t w = 0; w < (uint32_t)NW; ++w) {
    const uint64_t t = wtot[w];
    if (w < wave) basev += t;
    tot += t;
  }
  *total = tot;
  return basev + incl - x;
}

// every output index of [s0, s1) finds its run r in start[0..n_runs] (start[n_runs] = s1) and stores
// tab[2r] + tab[2r+1] * (i - start[r] + 1)   (DeltaRle, v5_codec.cpp:851-872; Rle has tab[2r+1] = 0)
__device__ __forceinline__ void fill_runs(const uint32_t* start, const uint64_t* tab, uint32_t n_runs, uint32_t s0,
                                          uint32_t s1, uint8_t* base, uint32_t step, uint32_t field_off, uint32_t bpv) {
  for (uint32_t i = s0 + threadIdx.x; i < s1; i += kDvThreads) {
    uint32_t lo = 0u, hi = n_runs;  // last r in [0, n_runs) with start[r] <= i (skips empty runs)
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (start[mid] <= i) lo = mid;
      else hi = mid;
    }
    const uint64_t v = tab[2u * lo] + tab[2u * lo + 1u] * (uint64_t)(i - start[lo] + 1u);
    st_raw(base + (size_t)i * step + field_off, v, bpv);
  }
}

// Palette section behind its mode byte (decodeV5AdaptiveIntSection, src/v5_codec.cpp:788-822): u16 count, the
// palette values, bit-packed indexes; thread t unpacks values [32t, 32t+32). The palette is staged in `pal_l` (LDS)
// when it has at most `pal_cap` entries. Returns true when something is wrong (block-uniform); `off` moves behind
// the section. Contains barriers.
__device__ __forceinline__ bool dec_palette(const uint8_t* __restrict__ src, uint32_t src_size, uint32_t& off, uint32_t n,
                                            uint8_t* base, uint32_t step, uint32_t field_off, uint32_t bpv,
                                            uint64_t* pal_l, uint32_t pal_cap, uint32_t* stage, uint32_t* misc) {
  constexpr int T = kDvThreads;
  const uint32_t tid = threadIdx.x;
  if (src_size - off < 2u) return true;
  const uint32_t count = (uint32_t)src[off] | ((uint32_t)src[off + 1u] << 8);
  off += 2u;
  if (count == 0u || (uint64_t)(src_size - off) < (uint64_t)count * bpv) return true;
  const uint8_t* pal = src + off;
  off += count * bpv;
  const uint32_t bits = palette_bits(count);
  const uint32_t index_bytes = (uint32_t)(((uint64_t)bits * n + 7u) / 8u);
  if (src_size - off < index_bytes) return true;
  const uint8_t* ip = src + off;
  __syncthreads();
  if (tid == 0) misc[0] = 0u;
  const bool pal_in_lds = count <= pal_cap;
  if (pal_in_lds) {
    for (uint32_t k = tid; k < count; k += T) {
      uint64_t v = 0;
      for (uint32_t b = 0; b < bpv; ++b) v |= ((uint64_t)pal[(size_t)k * bpv + b]) << (8u * b);
      pal_l[k] = v;
    }
  }
  __syncthreads();
  const uint8_t* ip_end = ip + index_bytes;
  if (bpv <= 4u && stage != nullptr) {
    // Rounds of 8192 values: thread t unpacks 8 consecutive indexes (their bits start on a byte boundary: 8 * bits
    // bits = `bits` bytes), looks the values up and parks them in LDS; after a barrier the same round is written out
    // with consecutive lanes on consecutive points, so a store instruction touches 8 cache lines instead of 64.
    const bool st_fast = (bpv == 2u && ((field_off | step) & 1u) == 0u) || (bpv == 4u && ((field_off | step) & 3u) == 0u);
    for (uint32_t r0 = 0; r0 < n; r0 += 8192u) {
      const uint32_t i0 = r0 + tid * 8u;
      if (i0 < n) {
        const uint32_t cnt = min(8u, n - i0);
        const uint8_t* ib = ip + (size_t)(i0 >> 3) * bits;
        const uint32_t mis = (uint32_t)((uintptr_t)ib & 3u);
        const uint32_t* iq = reinterpret_cast<const uint32_t*>(ib - mis);
        uint32_t nxt = (reinterpret_cast<const uint8_t*>(iq) < ip_end && bits) ? iq[0] : 0u;
        uint64_t scratch = 0u;
        uint32_t held = 0u, k = 0u;
        for (uint32_t j = 0; j < cnt; ++j) {
          uint32_t idx = 0u;
          if (bits) {
            if (held < bits) {
              const uint32_t cur_dw = nxt;
              ++k;
              nxt = (reinterpret_cast<const uint8_t*>(iq + k) < ip_end) ? iq[k] : 0u;
              scratch |= (uint64_t)__builtin_amdgcn_alignbyte(nxt, cur_dw, mis) << held;
              held += 32u;
            }
            idx = (uint32_t)(scratch & ((1ull << bits) - 1ull));
            scratch >>= bits;
            held -= bits;
          }
          if (idx >= count) {
            misc[0] = 1u;
            break;
          }
          uint32_t v = 0;
          if (pal_in_lds) {
            v = (uint32_t)pal_l[idx];
          } else {
            for (uint32_t b = 0; b < bpv; ++b) v |= ((uint32_t)pal[(size_t)idx * bpv + b]) << (8u * b);
          }
          stage[tid * 8u + j] = v;
        }
      }
      __syncthreads();
#pragma unroll
      for (uint32_t k = 0; k < 8u; ++k) {
        const uint32_t q = k * (uint32_t)T + tid;  // value q of the round
        if (r0 + q < n) {
          const uint32_t v = stage[q];
          uint8_t* o = base + (size_t)(r0 + q) * step + field_off;
          if (st_fast && bpv == 2u) *reinterpret_cast<uint16_t*>(o) = (uint16_t)v;
          else if (st_fast && bpv == 4u) *reinterpret_cast<uint32_t*>(o) = v;
          else st_raw(o, v, bpv);
        }
      }
      __syncthreads();
    }
    off += index_bytes;
    return misc[0] != 0u;
  }
  const uint32_t i0 = tid * 32u;
  if (i0 < n) {
    const uint32_t cnt = min(32u, n - i0);
    const uint32_t byte0 = tid * 4u * bits;  // 32 indexes = `bits` dwords
    // index dwords through aligned loads: the stream position is arbitrary, so fetch the aligned dwords around mine
    // and realign
    const uint8_t* ib = ip + byte0;
    const uint32_t mis = (uint32_t)((uintptr_t)ib & 3u);
    const uint32_t* iq = reinterpret_cast<const uint32_t*>(ib - mis);
    const uint8_t* ip_end = ip + index_bytes;
    uint32_t nxt = (reinterpret_cast<const uint8_t*>(iq) < ip_end && bits) ? iq[0] : 0u;
    const bool st_fast = (bpv == 2u && ((field_off | step) & 1u) == 0u) || (bpv == 4u && ((field_off | step) & 3u) == 0u);
    uint64_t scratch = 0u;
    uint32_t held = 0u, k = 0u;
    for (uint32_t produced = 0u; produced < cnt; ++produced) {
      uint32_t idx = 0u;
      if (bits) {
        if (held < bits) {
          const uint32_t cur_dw = nxt;
          ++k;
          nxt = (reinterpret_cast<const uint8_t*>(iq + k) < ip_end) ? iq[k] : 0u;
          const uint32_t dw = __builtin_amdgcn_alignbyte(nxt, cur_dw, mis);
          scratch |= (uint64_t)dw << held;
          held += 32u;
        }
        idx = (uint32_t)(scratch & ((1ull << bits) - 1ull));
        scratch >>= bits;
        held -= bits;
      }
      if (idx >= count) {
        misc[0] = 1u;
        break;
      }
      uint64_t v = 0;
      if (pal_in_lds) {
        v = pal_l[idx];
      } else {
        for (uint32_t b = 0; b < bpv; ++b) v |= ((uint64_t)pal[(size_t)idx * bpv + b]) << (8u * b);
      }
      uint8_t* o = base + (size_t)(i0 + produced) * step + field_off;
      if (st_fast && bpv == 2u) *reinterpret_cast<uint16_t*>(o) = (uint16_t)v;
      else if (st_fast && bpv == 4u) *reinterpret_cast<uint32_t*>(o) = (uint32_t)v;
      else st_raw(o, v, bpv);
    }
  }
  __syncthreads();
  off += index_bytes;
  return misc[0] != 0u;
}

// Chunks whose sections are all Palette with small palettes (the common case: intensity, ring, reflectivity ...) need
// almost no LDS; this kernel takes them at two workgroups per CU and leaves every other chunk to k_decode_sections,
// whose 100 KiB of LDS (token tiles, run tables) allow only one.
constexpr uint32_t kSmallPalEntries = 1024;
constexpr uint32_t kSmallSecLds = kSmallPalEntries * 8u + 8192u * 4u + 256u;  // palette, transposition buffer, misc

__device__ __forceinline__ void decode_sections_small_body(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                      const DecChunk* __restrict__ chunks,
                                                                      uint8_t* __restrict__ out,
                                                                      const uint32_t* __restrict__ reg_end,
                                                                      uint8_t* __restrict__ sec_done,
                                                                      uint32_t* __restrict__ status, uint32_t honor_folded) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (honor_folded && sec_done[blockIdx.x] == 2u) return;  // k_decode_points wrote this chunk's sections with its points
  uint64_t* pal_l = reinterpret_cast<uint64_t*>(smem);
  uint32_t* stage = reinterpret_cast<uint32_t*>(smem + kSmallPalEntries * 8u);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + kSmallPalEntries * 8u + 8192u * 4u);
  const uint32_t c = blockIdx.x;
  const DecChunk dc = chunks[c];
  if (threadIdx.x == 0) sec_done[c] = 0u;
  if (!dc.valid) return;
  uint32_t off = reg_end[c];
  if (off == kDecRedo || off > dc.src_size) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t step = plan.point_step;
  uint8_t* base = out + (size_t)dc.first_point * step;
  for (uint32_t a = 0; a < plan.n_adaptive; ++a) {
    // Palette with at most kSmallPalEntries values, else this chunk is not ours
    if (src_size - min(src_size, off) < 3u || src[off] != 1u) return;
    const uint32_t count = (uint32_t)src[off + 1u] | ((uint32_t)src[off + 2u] << 8);
    if (count > kSmallPalEntries) return;
    ++off;
    if (dec_palette(src, src_size, off, dc.n_points, base, step, plan.adaptive[a].offset, plan.adaptive[a].bpv, pal_l,
                    kSmallPalEntries, stage, misc))
      return;
  }
  if (off != src_size) return;  // trailing bytes: the serial decoder raises the error
  if (threadIdx.x == 0) {
    sec_done[c] = 1u;
    atomicAdd(&status[kStatFastSections], 1u);
  }
}

__global__ __launch_bounds__(kDvThreads) void k_decode_sections_small(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                      const DecChunk* __restrict__ chunks,
                                                                      uint8_t* __restrict__ out,
                                                                      const uint32_t* __restrict__ reg_end,
                                                                      uint8_t* __restrict__ sec_done,
                                                                      uint32_t* __restrict__ status, uint32_t honor_folded) {
  decode_sections_small_body(plan, streams, chunks, out, reg_end, sec_done, status, honor_folded);
}


// Where the decoded values of an adaptive field go: value i at base + i * step + off (the points of an AoS cloud, or
// a dense column: step = bytes per value, off = 0).
struct SecFieldOut {
  uint8_t* base;
  uint32_t step, off;
};

// The V5 sections of one chunk, field after field, from payload offset `off` on (see the overview above k_decode_varint).
// Returns true when every section was decoded and the payload ends with the last one; false = leave it to the serial
// decoder (which raises the errors). Block-uniform.
template <typename FieldOut>
__device__ __forceinline__ bool decode_sections_core(const DevPlan& plan, const uint8_t* __restrict__ src, uint32_t src_size,
                                                     uint32_t off, uint32_t n, FieldOut field_out, uint8_t* smem) {
  using DL = DvLds<1, true>;
  constexpr int T = kDvThreads;
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + DL::kMiscOff);  // [0] bad, [1] end, [2..34) scan, [40..48) handlers
  uint32_t* tileb = reinterpret_cast<uint32_t*>(smem + DL::kBytesOff);
  uint64_t* raw = reinterpret_cast<uint64_t*>(smem + DecSecLds::kRawOff);
  uint32_t* start = reinterpret_cast<uint32_t*>(smem + DecSecLds::kStartOff);
  uint64_t* scan64 = reinterpret_cast<uint64_t*>(smem + DL::kScanOff);  // 16 wave totals
  const uint32_t tid = threadIdx.x;
  bool bad = false;  // block-uniform throughout

  for (uint32_t a = 0; a < plan.n_adaptive && !bad; ++a) {
    const uint32_t bpv = plan.adaptive[a].bpv;
    const SecFieldOut fo = field_out(a);  // where the values of this field go
    uint8_t* const base = fo.base;
    const uint32_t step = fo.step, field_off = fo.off;
    if (off >= src_size) { bad = true; break; }
    const uint32_t mode = src[off];
    ++off;
    if (mode == 0u) {  // ---- DeltaVarint: n integer tokens
      DevOp op;
      op.kind = OP_INT;
      op.size = (uint8_t)bpv;
      op.offset = field_off;
      dv_stream<1, true>([&](int) -> const DevOp& { return op; }, 1u, src, src_size, off, n, base, step, smem);
      if (misc[0] || misc[1] == 0xffffffffu) bad = true;
      else off = misc[1];
      __syncthreads();
    } else if (mode == 1u) {  // ---- Palette
      // palette in the first 8192 token slots, the 8192-value transposition buffer behind it (both free here)
      if (dec_palette(src, src_size, off, n, base, step, field_off, bpv, raw, 4096u, reinterpret_cast<uint32_t*>(raw + 4096), misc))
        bad = true;
    } else if (mode == 2u || mode == 3u) {  // ---- Rle / DeltaRle
      if (src_size - off < 4u) { bad = true; break; }
      const uint32_t runs = (uint32_t)src[off] | ((uint32_t)src[off + 1u] << 8) | ((uint32_t)src[off + 2u] << 16) |
                            ((uint32_t)src[off + 3u] << 24);
      off += 4u;
      if (runs > n) { bad = true; break; }  // more runs than elements: empty runs, leave it to the serial decoder
      __syncthreads();
      if (tid == 0) {
        misc[0] = 0u;
        misc[1] = runs == 0u ? off : 0xffffffffu;
      }
      if (tid < 4u) tileb[tid] = 0u;
      __syncthreads();
      uint32_t out_index = 0u;  // elements written so far
      uint32_t runs_done = 0u;
      if (mode == 3u) {
        // (diff, run_len) token pairs, all varints: tokens -> LDS; per tile two block scans turn them into the run
        // table {start, value before the run, diff}, written over the tokens of the run
        uint64_t prev = 0u;  // value before the next run
        const uint32_t target = 2u * runs;
        uint32_t pos = off, left = 0u;
        while (2u * runs_done + left < target) {
          if (pos >= src_size) { bad = true; break; }
          const uint32_t seen = 2u * runs_done;
          const uint32_t n_tile = dv_tokens_tile(src, src_size, pos, tileb, misc, left, seen, target,
                                                 [&](uint32_t kl, uint64_t x, uint32_t end_off) {
                                                   raw[kl] = x;
                                                   if (seen + kl + 1u == target) misc[1] = end_off;
                                                 });
          const uint32_t avail = min(left + n_tile, target - seen);
          const uint32_t nr = avail / 2u;  // <= 4096
          const uint32_t per = (nr + T - 1u) / T;
          const uint32_t r0 = min(nr, tid * per), r1 = min(nr, r0 + per);
          uint32_t lsum = 0u;
          uint64_t psum = 0u;
          for (uint32_t r = r0; r < r1; ++r) {
            const uint64_t dx = raw[2u * r], len = raw[2u * r + 1u];
            if (dx == 0ull || len > (uint64_t)n) misc[0] = 1u;  // decodeVarint rejects the marker; run too long
            const uint64_t u1 = dx - 1ull;
            const uint64_t d = (u1 >> 1) ^ (0ull - (u1 & 1ull));
            lsum += (uint32_t)min(len, (uint64_t)n + 1ull);
            psum += d * len;
          }
          uint32_t ltot;
          const uint32_t lex = block_exclusive_scan<T>(lsum, misc + 2, &ltot);  // barrier inside
          __syncthreads();
          uint64_t ptot;
          const uint64_t pex = block_exclusive_scan_u64<T>(psum, scan64, &ptot);
          __syncthreads();
          if (misc[0] || (uint64_t)out_index + ltot > (uint64_t)n) { bad = true; break; }
          {
            uint32_t sidx = out_index + lex;
            uint64_t pv = prev + pex;
            for (uint32_t r = r0; r < r1; ++r) {
              const uint64_t dx = raw[2u * r], len = raw[2u * r + 1u];
              const uint64_t u1 = dx - 1ull;
              const uint64_t d = (u1 >> 1) ^ (0ull - (u1 & 1ull));
              start[r] = sidx;
              raw[2u * r] = pv;
              raw[2u * r + 1u] = d;
              sidx += (uint32_t)len;
              pv += d * len;
            }
          }
          if (tid == 0) start[nr] = out_index + ltot;
          __syncthreads();
          if (nr) fill_runs(start, raw, nr, out_index, out_index + ltot, base, step, field_off, bpv);
          const uint64_t carry_tok = (avail & 1u) ? raw[avail - 1u] : 0ull;  // a run cut by the tile edge
          dv_roll_history(tileb);  // barrier: everyone is done with the table
          if ((avail & 1u) && tid == 0) raw[0] = carry_tok;
          __syncthreads();
          out_index += ltot;
          prev += ptot;
          runs_done += nr;
          left = avail & 1u;
          pos += kDvTileBytes;
        }
        if (!bad) {
          if (misc[0] || misc[1] == 0xffffffffu) bad = true;
          else off = misc[1];
        }
      } else {
        // (raw value, run_len): raw bytes hide the token ends, so one lane walks the runs -- from an LDS copy of the
        // bytes, kRleRunTile runs per round -- and the workgroup fills
        uint8_t* stage = smem + DecSecLds::kStageOff;
        uint32_t pos = off;
        while (runs_done < runs) {
          const uint32_t nbytes = min(kRleStage, src_size - pos);
          for (uint32_t u = tid; u < nbytes; u += T) stage[u] = src[pos + u];
          __syncthreads();
          if (tid == 0) {
            uint32_t p = 0u, r = 0u, sidx = out_index;
            bool fail = false;
            const bool tail = (pos + nbytes == src_size);
            while (runs_done + r < runs && r < kRleRunTile) {
              if (!tail && nbytes - p < bpv + 10u) break;  // the record may continue behind the staged bytes
              if (nbytes - p < bpv) { fail = true; break; }
              uint64_t v = 0;
              for (uint32_t b = 0; b < bpv; ++b) v |= ((uint64_t)stage[p + b]) << (8u * b);
              p += bpv;
              uint64_t len = 0;
              uint32_t shift = 0;
              for (;;) {  // readUVarint, src/v5_codec.cpp:176-194
                if (p >= nbytes) { fail = true; break; }
                const uint8_t byte = stage[p++];
                len |= ((uint64_t)(byte & 0x7fu)) << shift;
                if ((byte & 0x80u) == 0) break;
                shift += 7u;
                if (shift >= 64u) { fail = true; break; }
              }
              if (fail) break;
              if (len > (uint64_t)n - (uint64_t)sidx) { fail = true; break; }  // no addition: len may be 2^64-1
              start[r] = sidx;
              raw[2u * r] = v;
              raw[2u * r + 1u] = 0ull;
              sidx += (uint32_t)len;
              ++r;
            }
            if (r == 0u && !fail) fail = true;  // no progress: a record larger than the stage cannot exist
            start[r] = sidx;
            misc[40] = r;
            misc[41] = p;
            misc[42] = sidx;
            if (fail) misc[0] = 1u;
          }
          __syncthreads();
          if (misc[0]) { bad = true; break; }
          const uint32_t nr = misc[40], used = misc[41], sidx = misc[42];
          fill_runs(start, raw, nr, out_index, sidx, base, step, field_off, bpv);
          __syncthreads();
          out_index = sidx;
          runs_done += nr;
          pos += used;
        }
        if (!bad) off = pos;
      }
      if (!bad && out_index != n) bad = true;
    } else {
      bad = true;
    }
  }
  __syncthreads();
  if (!bad && off != src_size) bad = true;  // trailing bytes: the serial decoder raises the error
  return !bad;
}

__device__ __forceinline__ void decode_sections_body(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                const DecChunk* __restrict__ chunks,
                                                                uint8_t* __restrict__ out,
                                                                const uint32_t* __restrict__ reg_end,
                                                                uint8_t* __restrict__ sec_done,
                                                                uint32_t* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t c = blockIdx.x;
  const DecChunk dc = chunks[c];
  if (sec_done[c]) return;  // k_decode_sections_small took this chunk (it also cleared the flag of all others)
  if (!dc.valid) return;
  const uint32_t off = reg_end[c];
  if (off == kDecRedo || off > dc.src_size) return;  // the serial decoder owns this chunk
  const uint32_t step = plan.point_step;
  uint8_t* base = out + (size_t)dc.first_point * step;
  const bool ok = decode_sections_core(plan, streams + dc.src_off, dc.src_size, off, dc.n_points,
                                       [&](uint32_t a) {
                                         SecFieldOut fo;
                                         fo.base = base;
                                         fo.step = step;
                                         fo.off = plan.adaptive[a].offset;
                                         return fo;
                                       },
                                       smem);
  if (ok && threadIdx.x == 0) {
    sec_done[c] = 1u;
    atomicAdd(&status[kStatFastSections], 1u);
  }
}

__global__ __launch_bounds__(kDvThreads) void k_decode_sections(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                const DecChunk* __restrict__ chunks,
                                                                uint8_t* __restrict__ out,
                                                                const uint32_t* __restrict__ reg_end,
                                                                uint8_t* __restrict__ sec_done,
                                                                uint32_t* __restrict__ status) {
  decode_sections_body(plan, streams, chunks, out, reg_end, sec_done, status);
}

// k_decode_tail: everything that may be left behind k_decode_points, in ONE launch (four mostly idle launches cost
// ~5 us each): chunks it handed back -> the varint decoder; sections it did not fold -> the two section decoders;
// whatever those stepped away from -> the serial decoder on thread 0. The bodies are the kernels above; a barrier
// between them orders what one leaves in reg_end / sec_done for the next. All early exits of the bodies are uniform
// over the workgroup (they depend on the chunk table, reg_end, sec_done and LDS words read behind barriers).
__global__ __launch_bounds__(kDvThreads) void k_decode_tail(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                            const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                            uint32_t* __restrict__ reg_end, uint8_t* __restrict__ sec_done,
                                                            uint32_t* __restrict__ status, uint32_t uses_v5,
                                                            uint32_t with_sections) {
  decode_varint_body<4, false>(plan, streams, chunks, out, reg_end, status, 1u);
  __syncthreads();
  if (with_sections) {
    decode_sections_small_body(plan, streams, chunks, out, reg_end, sec_done, status, 1u);
    __syncthreads();
    decode_sections_body(plan, streams, chunks, out, reg_end, sec_done, status);
    __syncthreads();
  }
  decode_general_body(plan, streams, chunks, out, uses_v5, 1u, reg_end, with_sections ? sec_done : nullptr, status);
}


}  // namespace cldn
