// stage1_wide.h -- the WIDE route of the encoder (round 5; included by stage1_kernels.hip behind the section functions).
//
// The reference accepts any schema (CreateCompatibleEncoder, src/codec_common.cpp:116-153; buildV5Plan,
// src/v5_codec.cpp:719-740): hundreds of fields, points of several KiB. The kernels of the ordinary route take their plan
// as a launch argument (DevPlan: at most 64 regular tokens, 64 adaptive integer fields, 1024-byte points). Everything
// beyond goes here, with the plan in device memory (WidePlan) and no structure that grows with the schema inside a
// kernel's arguments or LDS:
//
//   k_wide_probe    one workgroup per (cloud, adaptive field): the mode decision on the first <= 4096 values
//                   (analyzeAdaptiveIntField + selectBestAdaptiveIntMode, src/v5_codec.cpp:387-412, :934-949), values
//                   read from the AoS input -- probe_mode_t, the function the ordinary route uses.
//   k_wide_encode   one workgroup per chunk writes the chunk's payload as ONE run at the start of its slot:
//                   (1) regular stream, tiles of 1024 points, one thread per point: token lengths -> block scan ->
//                   tokens, straight from / to global memory (EncodeV5Stage1 / EncodeV4Stage1Chunk's per-point loop,
//                   src/v5_codec.cpp:900-963, src/v4_codec.cpp:66-83); (2) the adaptive sections field after field
//                   (src/v5_codec.cpp:690-717): the field's values -> a dense column in the chunk's scratch area, the
//                   section functions of k_encode_sections (section_delta_varint / section_runs / section_palette) ->
//                   scratch, copied behind what the chunk has so far.
//   framing         k_finish<256, 0> with one segment per chunk (stage1_launch_frame's arguments), or k_chunk_sizes for
//                   chunk-table calls.
// Gorilla-coded fields keep their pre-pass (k_gorilla_tokens), launched once per group of <= 64 such fields with a plan
// that holds only them.
//
// Speed: one workgroup per chunk and byte stores -- tens of Mpoints/s for a 100-field schema. It exists so that no schema
// the reference encodes is refused; the layouts sensors really produce never come here.
#pragma once

namespace cldn {

// scratch of one chunk: [column: 32768 x 8 B][first-occurrence indexes: 32768 x 2 B][one section, k_encode_sections' layout]
constexpr uint32_t kWideColOff = 0u;
constexpr uint32_t kWideFirstOff = kPointsPerChunk * 8u;
constexpr uint32_t kWideSecOff = kWideFirstOff + kPointsPerChunk * 2u;
constexpr uint32_t kWideScratchBytes = kWideSecOff + kSectionStride;
constexpr int kWideThreads = kSecThreads;  // the section functions' workgroup size

struct WideEncodeArgs {
  WidePlan plan;
  const uint8_t* points;
  const uint8_t* points_end;
  const ChunkDesc* chunks;
  const uint32_t* cloud_first_chunk;
  uint8_t* modes;                  // [n_clouds * n_adaptive]
  uint8_t* slots;
  unsigned long long slot_stride;
  Seg* segs;                       // [n_chunks]: one segment per chunk
  uint8_t* scratch;                // [n_chunks * kWideScratchBytes]
  const uint4* const* pre;         // device [n_gorilla]: k_gorilla_tokens' token buffers
};

// the `size` bytes at p, any alignment, nothing else touched
__device__ __forceinline__ uint64_t wide_load(const uint8_t* p, uint32_t size) {
  if (size == 1u) return *p;
  return aos_field(p, size);
}

// Regular op `op` of the point at `cur` (previous point of the chunk at `prev`, nullptr for the chunk's first point).
// EMIT = false: only the token's length. Same arithmetic as eval_op (k_encode_regular), fields read from global memory.
template <bool EMIT>
__device__ __forceinline__ Tok wide_eval_op(const DevOp& op, const uint8_t* cur, const uint8_t* prev, const uint4* gor, size_t gi) {
  Tok t;
  t.w0 = t.w1 = t.w2 = 0;
  t.len = 0;
  const uint8_t* fp = cur + op.offset;
  switch (op.kind) {
    case OP_QF32: {  // src/field_encoder.cpp:42-91
      const float v = __uint_as_float((uint32_t)wide_load(fp, 4u));
      if (is_nan_f32(v)) {
        t.len = 1;  // marker byte 0x00
        break;
      }
      int32_t prevq = 0;
      if (prev) {
        const float pv = __uint_as_float((uint32_t)wide_load(prev + op.offset, 4u));
        prevq = is_nan_f32(pv) ? 0 : quant_rne_i32(pv, op.mult_f);  // a NaN resets that lane's reference to 0
      }
      const int32_t d = (int32_t)((uint32_t)quant_rne_i32(v, op.mult_f) - (uint32_t)prevq);
      if (EMIT) t = varint32_tok(d);
      else t.len = varint32_len(d);
    } break;
    case OP_LOSSY_F32: {  // include/cloudini_lib/field_encoder.hpp:342-357
      const float v = __uint_as_float((uint32_t)wide_load(fp, 4u));
      if (is_nan_f32(v)) {
        t.len = 1;
        break;
      }
      int64_t prevq = 0;
      if (prev) {
        const float pv = __uint_as_float((uint32_t)wide_load(prev + op.offset, 4u));
        prevq = is_nan_f32(pv) ? 0 : quant_away_i64_f32(pv, op.mult_f);
      }
      const int64_t d = (int64_t)((uint64_t)quant_away_i64_f32(v, op.mult_f) - (uint64_t)prevq);
      if (EMIT) t = varint64_tok(d);
      else t.len = varint64_len(d);
    } break;
    case OP_LOSSY_F64: {
      const double v = __longlong_as_double((long long)wide_load(fp, 8u));
      if (is_nan_f64(v)) {
        t.len = 1;
        break;
      }
      int64_t prevq = 0;
      if (prev) {
        const double pv = __longlong_as_double((long long)wide_load(prev + op.offset, 8u));
        prevq = is_nan_f64(pv) ? 0 : quant_away_i64_f64(pv, op.mult_d);
      }
      const int64_t d = (int64_t)((uint64_t)quant_away_i64_f64(v, op.mult_d) - (uint64_t)prevq);
      if (EMIT) t = varint64_tok(d);
      else t.len = varint64_len(d);
    } break;
    case OP_INT: {  // include/cloudini_lib/field_encoder.hpp:78-85
      const int64_t v = int_field_as_i64(wide_load(fp, op.size), op.type);
      const int64_t pv = prev ? int_field_as_i64(wide_load(prev + op.offset, op.size), op.type) : 0;
      const int64_t d = (int64_t)((uint64_t)v - (uint64_t)pv);
      if (EMIT) t = varint64_tok(d);
      else t.len = varint64_len(d);
    } break;
    case OP_COPY: {  // include/cloudini_lib/field_encoder.hpp:56-60
      if (EMIT) t = raw_tok(wide_load(fp, op.size), op.size);
      else t.len = op.size;
    } break;
    case OP_XOR32:
    case OP_XOR64: {  // include/cloudini_lib/field_encoder.hpp:359-370
      if (EMIT) {
        const uint64_t v = wide_load(fp, op.size);
        const uint64_t pv = prev ? wide_load(prev + op.offset, op.size) : 0;
        t = raw_tok(v ^ pv, op.size);
      } else {
        t.len = op.size;
      }
    } break;
    case OP_GORILLA64: {  // tokens built by k_gorilla_tokens
      const uint4 g = gor[gi];
      t.w0 = g.x;
      t.w1 = g.y;
      t.w2 = g.z;
      t.len = g.w;
    } break;
    default:
      break;
  }
  return t;
}

// grid = n_clouds * n_adaptive, kS2Threads threads, kProbeLds bytes of LDS
__global__ __launch_bounds__(kS2Threads) void k_wide_probe(const WideEncodeArgs A, uint32_t n_clouds) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t na = A.plan.n_adaptive;
  const uint32_t cloud = blockIdx.x / na, a = blockIdx.x - cloud * na;
  uint8_t* mode_out = A.modes + (size_t)cloud * na + a;
  const uint32_t fc = A.cloud_first_chunk[cloud];
  if (fc == A.cloud_first_chunk[cloud + 1u]) {  // empty cloud
    if (threadIdx.x == 0) *mode_out = 0u;
    return;
  }
  const ChunkDesc cd = A.chunks[fc];
  const uint32_t n = cd.n_points > kProbePoints ? kProbePoints : cd.n_points;
  const DevAdaptive f = A.plan.adaptive[a];
  const uint32_t step = A.plan.point_step;
  const uint8_t* fp = A.points + (size_t)cd.first_point * step + f.offset;
  uint32_t* wtot = reinterpret_cast<uint32_t*>(smem + kProbeSlots * 8u + 16u);
  uint8_t mode;
  if (f.bpv == 2u)
    mode = probe_mode_t<uint16_t, kS2Threads>([&](uint32_t i) { return (uint16_t)aos_field(fp + (size_t)i * step, 2u); }, n, f.type, smem,
                                              kProbeSlots, wtot);
  else if (f.bpv == 4u)
    mode = probe_mode_t<uint32_t, kS2Threads>([&](uint32_t i) { return (uint32_t)aos_field(fp + (size_t)i * step, 4u); }, n, f.type, smem,
                                              kProbeSlots, wtot);
  else
    mode = probe_mode_t<uint64_t, kS2Threads>([&](uint32_t i) { return (uint64_t)aos_field(fp + (size_t)i * step, 8u); }, n, f.type, smem,
                                              kProbeSlots, wtot);
  if (threadIdx.x == 0) *mode_out = mode;
}

// grid = n_chunks, kWideThreads threads, kSecLdsTotal bytes of LDS (the section functions' carve-up)
__global__ __launch_bounds__(kWideThreads) void k_wide_encode(const WideEncodeArgs A) {
  constexpr int T = kWideThreads;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SecLds l = sec_lds_carve(smem);
  const uint32_t tid = threadIdx.x;
  const uint32_t c = blockIdx.x;
  const ChunkDesc cd = A.chunks[c];
  const uint32_t n = cd.n_points;
  const uint32_t step = A.plan.point_step;
  const uint32_t n_ops = A.plan.n_ops;
  const uint8_t* gchunk = A.points + (size_t)cd.first_point * step;
  uint8_t* slot = A.slots + (size_t)c * A.slot_stride;

  // ---- (1) the regular stream: for each point in order, for each regular encoder in order, its bytes
  uint32_t R = 0u;
  for (uint32_t p0 = 0; p0 < n; p0 += T) {
    const uint32_t i = p0 + tid;
    const bool active = i < n;
    const uint8_t* cur = gchunk + (size_t)(active ? i : 0u) * step;
    const uint8_t* prev = i > 0u ? cur - step : nullptr;
    const size_t gi = (size_t)cd.first_point + i;
    uint32_t my_len = 0u;
    if (active)
      for (uint32_t k = 0; k < n_ops; ++k) {
        const DevOp op = A.plan.ops[k];
        const uint4* gor = op.kind == OP_GORILLA64 ? A.pre[A.plan.op_aux[k]] : nullptr;
        my_len += wide_eval_op<false>(op, cur, prev, gor, gi).len;
      }
    uint32_t tile_total;
    const uint32_t excl = block_exclusive_scan<T>(my_len, l.wtot, &tile_total);
    if (active) {
      uint8_t* dst = slot + R + excl;
      for (uint32_t k = 0; k < n_ops; ++k) {
        const DevOp op = A.plan.ops[k];
        const uint4* gor = op.kind == OP_GORILLA64 ? A.pre[A.plan.op_aux[k]] : nullptr;
        const Tok t = wide_eval_op<true>(op, cur, prev, gor, gi);
        const uint64_t lo = (((uint64_t)t.w1) << 32) | t.w0;
        for (uint32_t b = 0; b < t.len; ++b) dst[b] = (uint8_t)(b < 8u ? (lo >> (8u * b)) : ((uint64_t)t.w2 >> (8u * (b - 8u))));
        dst += t.len;
      }
    }
    R += tile_total;
    __syncthreads();  // (wtot is reused by the next scan)
  }

  // ---- (2) the adaptive sections, field after field in field order
  const uint32_t na = A.plan.n_adaptive;
  uint8_t* scratch = A.scratch + (size_t)c * kWideScratchBytes;
  uint8_t* col = scratch + kWideColOff;
  uint16_t* first_idx = reinterpret_cast<uint16_t*>(scratch + kWideFirstOff);
  uint8_t* sec = scratch + kWideSecOff;
  for (uint32_t a = 0; a < na; ++a) {
    const DevAdaptive f = A.plan.adaptive[a];
    const uint32_t bpv = f.bpv;
    const uint32_t mode = A.modes[(size_t)cd.cloud * na + a];
    for (uint32_t i = tid; i < n; i += T) {
      const uint64_t raw = aos_field(gchunk + (size_t)i * step + f.offset, bpv);
      if (bpv == 2u) reinterpret_cast<uint16_t*>(col)[i] = (uint16_t)raw;
      else if (bpv == 4u) reinterpret_cast<uint32_t*>(col)[i] = (uint32_t)raw;
      else reinterpret_cast<uint64_t*>(col)[i] = raw;
    }
    for (uint32_t i = tid; i < kRingU4; i += T) reinterpret_cast<uint4*>(l.ring)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    uint32_t size_a = 0u, size_b = 0u;
    if (mode == 0u) {
      size_a = section_delta_varint<T, true>(col, bpv, f.type, n, l.ring, l.wtot, sec);
    } else if (mode == 2u) {
      size_a = section_runs<T, true, false>(col, bpv, f.type, n, l.ring, l.wtot, sec, l.list_pos, l.list_key);
    } else if (mode == 3u) {
      size_a = section_runs<T, true, true>(col, bpv, f.type, n, l.ring, l.wtot, sec, l.list_pos, l.list_key);
    } else {
      section_palette<T>(col, bpv, n, sec, first_idx, l.main, l.wtot, &size_a, &size_b);
    }
    __syncthreads();  // the section's bytes are in `sec` (and its indexes at kPaletteIndexOffset)
    for (uint32_t i = tid; i < size_a; i += T) slot[R + i] = sec[i];
    for (uint32_t i = tid; i < size_b; i += T) slot[R + size_a + i] = sec[kPaletteIndexOffset + i];
    R += size_a + size_b;
    __syncthreads();  // before the next field overwrites the column and the section area
  }
  if (tid == 0) {
    Seg s;
    s.off = 0u;
    s.size = R;
    A.segs[c] = s;
  }
}

}  // namespace cldn
