// stage1_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the Cloudini stage-1 encoder.
//
// Work decomposition: one workgroup per 32768-point chunk. All encoder state of the reference
// resets at a chunk boundary (src/v4_codec.cpp:69, src/v5_codec.cpp:910-915), so chunks are independent; inside
// a chunk the only sequential quantity is the output byte position, which becomes a loop-carried scalar of the
// workgroup's tile loop plus one block-wide prefix sum per tile.
//
//   k_encode_regular   AoS tile -> LDS (coalesced 16 B/lane), per point: quantise / delta against the previous
//                      point / varint tokens; block scan of token bytes; tokens OR-ed into an LDS byte ring;
//                      ring flushed with 16 B/lane stores. Also splits the V5 adaptive-int fields out into SoA
//                      columns (the "AoS->SoA channel split") for the section kernel.
//   k_chunk_offsets    exclusive scan of (4 + payload) over the batch's chunks.
//   k_compact          concatenates each chunk's segments behind its [u32 size] prefix into the final framed
//                      stream (byte-exact, arbitrary destination alignment).
//
// This is integer / bit-pack work bounded by HBM bandwidth: no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "stage1_device.h"
#include "stage1_math.h"
#include "stage1_prims.h"

namespace cldn {

// ------------------------------------------------------------------------------------------------------------
// byte-stream writer: tokens are OR-ed into a zero-initialised LDS ring at their stream byte offset, the ring
// is flushed to global memory in whole 16-byte units (coalesced dwordx4 stores) and re-zeroed as it drains.
// ------------------------------------------------------------------------------------------------------------

constexpr uint32_t kRingBytes = 16384;
constexpr uint32_t kRingDw = kRingBytes / 4;
constexpr uint32_t kRingU4 = kRingBytes / 16;

template <bool WINDOWED>
__device__ __forceinline__ void ring_or(uint32_t* ring, uint32_t g, uint32_t d, uint32_t win_lo_dw) {
  if (d == 0u) return;
  if (WINDOWED) {
    if (g < win_lo_dw || g >= win_lo_dw + kRingDw) return;
  }
  atomicOr(&ring[g & (kRingDw - 1u)], d);
}

// OR token `t` (<= 12 bytes) into the ring at stream byte offset `off`
template <bool WINDOWED>
__device__ __forceinline__ void ring_put(uint32_t* ring, uint32_t off, const Tok t, uint32_t win_lo_dw) {
  const uint32_t sh = (off & 3u) * 8u;
  const uint32_t g = off >> 2;
  const uint64_t lo = ((((uint64_t)t.w1) << 32) | t.w0) << sh;
  ring_or<WINDOWED>(ring, g, (uint32_t)lo, win_lo_dw);
  if (((off & 3u) + t.len) > 4u) {
    ring_or<WINDOWED>(ring, g + 1u, (uint32_t)(lo >> 32), win_lo_dw);
    if (((off & 3u) + t.len) > 8u) {
      const uint64_t hi = ((((uint64_t)t.w2) << 32) | t.w1) << sh;
      ring_or<WINDOWED>(ring, g + 2u, (uint32_t)(hi >> 32), win_lo_dw);
      ring_or<WINDOWED>(ring, g + 3u, (uint32_t)((((uint64_t)t.w2) << sh) >> 32), win_lo_dw);
    }
  }
}

// flush ring bytes [from, to) (both multiples of 16) to dst + offset and zero them in the ring
template <int T>
__device__ __forceinline__ void ring_flush(uint32_t* ring, uint8_t* dst, uint32_t from, uint32_t to) {
  uint4* ring4 = reinterpret_cast<uint4*>(ring);
  for (uint32_t u = (from >> 4) + threadIdx.x; u < (to >> 4); u += T) {
    const uint32_t r = u & (kRingU4 - 1u);
    const uint4 v = ring4[r];
    *reinterpret_cast<uint4*>(dst + (size_t)u * 16u) = v;
    ring4[r] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// State of one output stream of a workgroup (uniform across the block).
struct StreamState {
  uint32_t R;  // bytes produced so far
  uint32_t F;  // bytes flushed so far (multiple of 16, F <= R)
};

// ------------------------------------------------------------------------------------------------------------
// k_encode_regular
// ------------------------------------------------------------------------------------------------------------

struct PointRef {
  uint32_t cur;       // LDS byte offset of this thread's point
  uint32_t prev;      // LDS byte offset of the previous point of the chunk (valid if has_prev)
  bool has_prev;
};

// Evaluate regular op `op` for one point. EMIT=false: only the token length.
template <bool EMIT>
__device__ __forceinline__ Tok eval_op(const DevOp& op, const uint32_t* tile, const PointRef p, const PreTokenPtrs& pre,
                                       size_t gi) {
  Tok t;
  t.w0 = t.w1 = t.w2 = 0;
  t.len = 0;
  switch (op.kind) {
    case OP_QF32: {  // src/field_encoder.cpp:42-91
      const float v = __uint_as_float(lds_u32(tile, p.cur + op.offset));
      if (is_nan_f32(v)) {
        t.len = 1;  // marker byte 0x00
        break;
      }
      int32_t prevq = 0;
      if (p.has_prev) {
        const float pv = __uint_as_float(lds_u32(tile, p.prev + op.offset));
        prevq = is_nan_f32(pv) ? 0 : quant_rne_i32(pv, op.mult_f);  // NaN resets that lane's prev to 0
      }
      const int32_t d = (int32_t)((uint32_t)quant_rne_i32(v, op.mult_f) - (uint32_t)prevq);
      if (EMIT) t = varint32_tok(d);
      else t.len = varint32_len(d);
    } break;
    case OP_LOSSY_F32: {  // include/cloudini_lib/field_encoder.hpp:342-357
      const float v = __uint_as_float(lds_u32(tile, p.cur + op.offset));
      if (is_nan_f32(v)) {
        t.len = 1;
        break;
      }
      int64_t prevq = 0;
      if (p.has_prev) {
        const float pv = __uint_as_float(lds_u32(tile, p.prev + op.offset));
        prevq = is_nan_f32(pv) ? 0 : quant_away_i64_f32(pv, op.mult_f);
      }
      const int64_t d = (int64_t)((uint64_t)quant_away_i64_f32(v, op.mult_f) - (uint64_t)prevq);
      if (EMIT) t = varint64_tok(d);
      else t.len = varint64_len(d);
    } break;
    case OP_LOSSY_F64: {
      const double v = __longlong_as_double((long long)lds_u64(tile, p.cur + op.offset));
      if (is_nan_f64(v)) {
        t.len = 1;
        break;
      }
      int64_t prevq = 0;
      if (p.has_prev) {
        const double pv = __longlong_as_double((long long)lds_u64(tile, p.prev + op.offset));
        prevq = is_nan_f64(pv) ? 0 : quant_away_i64_f64(pv, op.mult_d);
      }
      const int64_t d = (int64_t)((uint64_t)quant_away_i64_f64(v, op.mult_d) - (uint64_t)prevq);
      if (EMIT) t = varint64_tok(d);
      else t.len = varint64_len(d);
    } break;
    case OP_INT: {  // include/cloudini_lib/field_encoder.hpp:78-85
      const int64_t v = int_field_as_i64(lds_raw(tile, p.cur + op.offset, op.size), op.type);
      const int64_t pv = p.has_prev ? int_field_as_i64(lds_raw(tile, p.prev + op.offset, op.size), op.type) : 0;
      const int64_t d = (int64_t)((uint64_t)v - (uint64_t)pv);
      if (EMIT) t = varint64_tok(d);
      else t.len = varint64_len(d);
    } break;
    case OP_COPY: {  // include/cloudini_lib/field_encoder.hpp:56-60
      if (EMIT) t = raw_tok(lds_raw(tile, p.cur + op.offset, op.size), op.size);
      else t.len = op.size;
    } break;
    case OP_XOR32:
    case OP_XOR64: {  // include/cloudini_lib/field_encoder.hpp:359-370
      if (EMIT) {
        const uint64_t v = lds_raw(tile, p.cur + op.offset, op.size);
        const uint64_t pv = p.has_prev ? lds_raw(tile, p.prev + op.offset, op.size) : 0;
        t = raw_tok(v ^ pv, op.size);
      } else {
        t.len = op.size;
      }
    } break;
    case OP_GORILLA64: {  // bit-packed XOR window codec, tokens built by k_gorilla_tokens
      const uint4 g = pre.p[op.type][gi];
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

struct TileGeom {
  const uint8_t* a0;   // 16-byte aligned global address of the first staged unit
  uint32_t units;      // 16-byte units to stage
  uint32_t first_off;  // LDS byte offset of the tile's first point
  uint32_t npts;       // points in the tile
  uint32_t p0;         // index of the tile's first point inside the chunk
};

__device__ __forceinline__ TileGeom tile_geom(const uint8_t* gchunk, uint32_t step, uint32_t P, uint32_t n,
                                              uint32_t it, bool first_has_prev) {
  TileGeom g;
  g.p0 = it * P;
  g.npts = min(P, n - g.p0);
  const uint32_t lead = (it > 0u || first_has_prev) ? step : 0u;  // stage the previous point too (delta reference)
  const uint8_t* ga0 = gchunk + (size_t)g.p0 * step - lead;
  const uint32_t mis = (uint32_t)((uintptr_t)ga0 & 15u);
  g.a0 = ga0 - mis;
  g.units = (mis + lead + g.npts * step + 15u) >> 4;
  g.first_off = mis + lead;
  return g;
}

// 16 bytes from global memory; bytes outside [lo, hi) are never touched (first/last unit of a buffer whose
// base or end is not 16-byte aligned)
__device__ __forceinline__ uint4 load_unit_guarded(const uint8_t* addr, const uint8_t* lo, const uint8_t* hi) {
  if (addr >= lo && addr + 16 <= hi) {
    return *reinterpret_cast<const uint4*>(addr);
  }
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  for (int k = 0; k < 16; ++k) {
    const uint8_t* q = addr + k;
    if (q >= lo && q < hi) w[k >> 2] |= ((uint32_t)(*q)) << ((k & 3) * 8);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// LDS bytes of one staged tile: 16 KiB (one unit per thread) for points up to 256 bytes, 64 points for wider ones
__host__ __device__ inline uint32_t regular_tile_lds(uint32_t T, uint32_t step) {
  const uint32_t body = step <= kWidePointStep ? T * 16u : 64u * step;
  return ((body + step + 48u) + 15u) & ~15u;
}

// WIDE: point_step > 256 (tiles of 64 points, up to 5 staged units per thread)
template <int T, bool WIDE = false>
__global__ __launch_bounds__(T) void k_encode_regular(const DevPlan plan, const uint8_t* __restrict__ points,
                                                      const uint8_t* points_end,
                                                      const ChunkDesc* __restrict__ chunks,
                                                      uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                      Seg* __restrict__ segs, uint32_t segs_per_chunk,
                                                      const ColumnPtrs cols, uint32_t subs, uint32_t sub_points,
                                                      uint32_t sub_stride, const PreTokenPtrs pre) {
  constexpr int UPT = WIDE ? 5 : 2;  // 16-byte units a thread stages per tile
  const uint32_t kTileLds = regular_tile_lds(T, plan.point_step);
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* ring = reinterpret_cast<uint32_t*>(smem + 2u * kTileLds);
  uint32_t* wtot = reinterpret_cast<uint32_t*>(smem + 2u * kTileLds + kRingBytes);

  const uint32_t tid = threadIdx.x;
  // one workgroup per sub-chunk: points [sub_first, sub_first + n) of chunk `chunk_id`. Sub-chunks produce
  // independent byte streams (own segment); only the delta reference of the first point crosses the boundary.
  const uint32_t chunk_id = blockIdx.x / subs;
  const uint32_t sub_id = blockIdx.x - chunk_id * subs;
  const ChunkDesc cd = chunks[chunk_id];
  const uint32_t step = plan.point_step;
  const uint32_t sub_first = sub_id * sub_points;
  // points of this sub-chunk; written with a saturating subtraction (hipcc 7.2 folded the guarded form
  // "n_points > sub_first ? min(sub_points, n_points - sub_first) : 0" into an unguarded unsigned min)
  const uint32_t n = min(sub_points, cd.n_points - min(cd.n_points, sub_first));
  const uint32_t P = WIDE ? 64u : min((uint32_t)T, ((T * 16u) / step) & ~63u);  // points per tile (multiple of 64)
  const uint32_t n_tiles = (n + P - 1u) / P;
  const uint8_t* gchunk = points + ((size_t)cd.first_point + sub_first) * step;
  uint8_t* slot = slots + (size_t)chunk_id * slot_stride + (size_t)sub_id * sub_stride;
  if (n == 0u) {
    if (tid == 0) {
      Seg s;
      s.off = sub_id * sub_stride;
      s.size = 0u;
      segs[(size_t)chunk_id * segs_per_chunk + sub_id] = s;
    }
    return;
  }

  // zero the ring
  for (uint32_t i = tid; i < kRingU4; i += T) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0u, 0u, 0u, 0u);

  // stage tile 0
  const bool sub_has_prev = sub_first > 0u;
  TileGeom g = tile_geom(gchunk, step, P, n, 0u, sub_has_prev);
  for (uint32_t u = tid; u < g.units; u += T) {
    reinterpret_cast<uint4*>(smem)[u] = load_unit_guarded(g.a0 + (size_t)u * 16u, points, points_end);
  }
  __syncthreads();

  StreamState ss;
  ss.R = 0u;
  ss.F = 0u;

  for (uint32_t it = 0; it < n_tiles; ++it) {
    const uint32_t* tile = reinterpret_cast<const uint32_t*>(smem + (it & 1u) * kTileLds);
    uint32_t* tile_next = reinterpret_cast<uint32_t*>(smem + ((it + 1u) & 1u) * kTileLds);

    // issue the global loads of the next tile now; they land in LDS at the end of this iteration
    TileGeom gn;
    gn.units = 0u;
    uint4 staged[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) staged[k] = make_uint4(0u, 0u, 0u, 0u);
    if (it + 1u < n_tiles) {
      gn = tile_geom(gchunk, step, P, n, it + 1u, sub_has_prev);
#pragma unroll
      for (int k = 0; k < UPT; ++k)
        if (tid + (uint32_t)k * T < gn.units) staged[k] = load_unit_guarded(gn.a0 + (size_t)(tid + (uint32_t)k * T) * 16u, points, points_end);
    }

    const bool active = tid < g.npts;
    PointRef pr;
    pr.cur = g.first_off + tid * step;
    pr.prev = pr.cur - step;
    pr.has_prev = (sub_first + g.p0 + tid) > 0u;
    const size_t gi_point = (size_t)cd.first_point + sub_first + g.p0 + tid;  // index into per-point side buffers

    // Tokens of this point. Schemas with at most kKeepOps regular ops build every token once and keep it in
    // registers across the scan; longer ones make a length pass first and rebuild the tokens when they emit.
    constexpr uint32_t kKeepOps = 8;
    const bool keep = plan.n_ops <= kKeepOps;  // uniform
    Tok kept[kKeepOps];
    uint32_t my_len = 0u;
    if (keep) {
#pragma unroll
      for (uint32_t k = 0; k < kKeepOps; ++k) {
        kept[k].w0 = kept[k].w1 = kept[k].w2 = 0u;
        kept[k].len = 0u;
        if (active && k < plan.n_ops) kept[k] = eval_op<true>(plan.ops[k], tile, pr, pre, gi_point);
        my_len += kept[k].len;
      }
    } else if (active) {
      for (uint32_t k = 0; k < plan.n_ops; ++k) my_len += eval_op<false>(plan.ops[k], tile, pr, pre, gi_point).len;
    }
    uint32_t tile_total;
    const uint32_t excl = block_exclusive_scan<T>(my_len, wtot, &tile_total);

    const uint32_t r_end = ss.R + tile_total;
    const bool last = (it + 1u == n_tiles);
    const uint32_t target = last ? ((r_end + 15u) & ~15u) : (r_end & ~15u);

    if (r_end - ss.F <= kRingBytes) {
      // common case: the whole tile fits the ring
      if (active) {
        uint32_t off = ss.R + excl;
        if (keep) {
#pragma unroll
          for (uint32_t k = 0; k < kKeepOps; ++k) {
            if (kept[k].len) ring_put<false>(ring, off, kept[k], 0u);
            off += kept[k].len;
          }
        } else {
          for (uint32_t k = 0; k < plan.n_ops; ++k) {
            const Tok t = eval_op<true>(plan.ops[k], tile, pr, pre, gi_point);
            ring_put<false>(ring, off, t, 0u);
            off += t.len;
          }
        }
      }
      __syncthreads();
      ring_flush<T>(ring, slot, ss.F, target);
      ss.F = target;
    } else {
      // rare: huge tokens (many wide fields); emit in ring-sized windows
      for (;;) {
        if (active) {
          uint32_t off = ss.R + excl;
          for (uint32_t k = 0; k < plan.n_ops; ++k) {
            const Tok t = eval_op<true>(plan.ops[k], tile, pr, pre, gi_point);  // rebuilt: no dynamic index into `kept`
            ring_put<true>(ring, off, t, ss.F >> 2);
            off += t.len;
          }
        }
        __syncthreads();
        const uint32_t nf = min(ss.F + kRingBytes, target);
        ring_flush<T>(ring, slot, ss.F, nf);
        const bool done = (ss.F + kRingBytes >= r_end);
        ss.F = nf;
        if (done) break;
        __syncthreads();
      }
    }
    ss.R = r_end;

    // AoS -> SoA split of the adaptive-int fields
    if (active) {
      const size_t gi = (size_t)cd.first_point + sub_first + g.p0 + tid;
      for (uint32_t a = 0; a < plan.n_adaptive; ++a) {
        const uint32_t bpv = plan.adaptive[a].bpv;
        const uint64_t raw = lds_raw(tile, pr.cur + plan.adaptive[a].offset, bpv);
        uint8_t* col = cols.p[a];
        if (bpv == 2u) reinterpret_cast<uint16_t*>(col)[gi] = (uint16_t)raw;
        else if (bpv == 4u) reinterpret_cast<uint32_t*>(col)[gi] = (uint32_t)raw;
        else reinterpret_cast<uint64_t*>(col)[gi] = raw;
      }
    }

    // land the prefetched tile
    if (it + 1u < n_tiles) {
#pragma unroll
      for (int k = 0; k < UPT; ++k)
        if (tid + (uint32_t)k * T < gn.units) reinterpret_cast<uint4*>(tile_next)[tid + (uint32_t)k * T] = staged[k];
      g = gn;
    }
    __syncthreads();
  }

  if (tid == 0) {
    Seg s;
    s.off = sub_id * sub_stride;
    s.size = ss.R;
    segs[(size_t)chunk_id * segs_per_chunk + sub_id] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_encode_floatn: the hot kernel. Regular stream == one FieldEncoderFloatN_Lossy (3 or 4 fused float32 lanes at
// consecutive offsets; src/field_encoder.cpp:42-91), 4-byte aligned layout. Differences to the generic kernel:
//   * no LDS staging of the input: every lane loads its own point straight from global memory (the wave reads
//     64 consecutive points = one contiguous, coalesced span);
//   * each wave covers 63 new points plus, in lane 0, the point before them: the delta reference of lane l is
//     lane l-1's quantised value, fetched with one DPP wave shift -- no second load, no second quantisation, no
//     cross-wave exchange. Lane 0 never emits;
//   * PPT rows of 63*NW points per barrier pair; the cross-wave part of the scan is a single wave scan over the
//     PPT*NW row/wave totals;
//   * tokens are built once (<= 5 bytes each) and OR-ed into the byte ring, 1-2 LDS atomics per token.
// ------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t dpp_wave_shr1(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false);  // wave_shr:1
}

// token of one FloatN lane; exact for every input incl. NaN (marker byte) and the 33-bit case d == INT32_MIN
__device__ __forceinline__ void floatn_token(bool is_nan, int32_t d, uint32_t& w0, uint32_t& w1, uint32_t& len) {
  const uint32_t zz = ((uint32_t)d << 1) ^ (uint32_t)(d >> 31);
  const bool ov = zz == 0xffffffffu;
  const uint32_t u = zz + 1u;
  uint32_t l = groups7(32u - (uint32_t)__clz((int)u));
  l = ov ? 5u : l;
  l = is_nan ? 1u : l;
  // continuation flags in the low (l-1) bytes: 0x0080808080 >> 8*(5-l), taken from a 64-bit constant
  const uint32_t cont = (uint32_t)(0x0080808080ull >> (8u * (5u - l)));
  uint32_t a = spread28(u & 0x0fffffffu) | cont;
  uint32_t b = ov ? 0x10u : (u >> 28);
  w0 = is_nan ? 0u : a;
  w1 = is_nan ? 0u : b;
  len = l;
}

template <uint32_t RING_BYTES, bool WINDOWED>
__device__ __forceinline__ void ring_put5(uint32_t* ring, uint32_t off, uint32_t w0, uint32_t w1, uint32_t len,
                                          uint32_t win_lo_dw) {
  constexpr uint32_t kMask = RING_BYTES / 4u - 1u;
  const uint32_t sh = (off & 3u) * 8u;
  const uint32_t g = off >> 2;
  const uint64_t v = ((((uint64_t)w1) << 32) | w0) << sh;
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  if (!WINDOWED || (g >= win_lo_dw && g < win_lo_dw + RING_BYTES / 4u)) {
    if (lo) atomicOr(&ring[g & kMask], lo);
  }
  if (((off & 3u) + len) > 4u) {
    if (!WINDOWED || (g + 1u >= win_lo_dw && g + 1u < win_lo_dw + RING_BYTES / 4u)) {
      if (hi) atomicOr(&ring[(g + 1u) & kMask], hi);
    }
  }
}

// token of <= 4 bytes at byte `off` (whole tile fits the ring): two dword ORs, the second one predicated
template <uint32_t RING_BYTES>
__device__ __forceinline__ void ring_put4(uint32_t* ring, uint32_t off, uint32_t t) {
  const uint64_t v = ((uint64_t)t) << ((off << 3) & 31u);
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  uint8_t* rb = reinterpret_cast<uint8_t*>(ring);
  atomicOr(reinterpret_cast<uint32_t*>(rb + (off & (RING_BYTES - 4u))), lo);
  if (hi) atomicOr(reinterpret_cast<uint32_t*>(rb + ((off + 4u) & (RING_BYTES - 4u))), hi);
}

// u < 2^28 -> its four 7-bit groups in the low 7 bits of the four bytes, continuation bits of a token of `l`
// bytes set (two bit-field inserts per halving step; the stray bits they leave sit on the continuation positions)
__device__ __forceinline__ uint32_t token4(uint32_t u, uint32_t l) {
  const uint32_t x = (u & 0x00003fffu) | ((u << 2) & ~0x00003fffu);
  const uint32_t y = (x & 0x007f007fu) | ((x << 1) & ~0x007f007fu);
  return (y & 0x7f7f7f7fu) | (0x00808080u >> ((32u - 8u * l) & 31u));
}

template <int T, uint32_t RING_BYTES>
__device__ __forceinline__ void ring_flush_n(uint32_t* ring, uint8_t* dst, uint32_t from, uint32_t to) {
  uint4* ring4 = reinterpret_cast<uint4*>(ring);
  for (uint32_t u = (from >> 4) + threadIdx.x; u < (to >> 4); u += T) {
    const uint32_t r = u & (RING_BYTES / 16u - 1u);
    const uint4 v = ring4[r];
    __builtin_memcpy(dst + (size_t)u * 16u, &v, 16);  // (dst may be unaligned: a section appended behind the regular stream)
    ring4[r] = make_uint4(0u, 0u, 0u, 0u);
  }
}

constexpr uint32_t kStagedCols = 2;  // adaptive fields (2 or 4 bytes wide) whose SoA copy is staged through LDS

// adaptive field `a` lies completely inside the `loadw` dwords loaded behind the first float lane. The host picks
// loadw so that this holds for every field (floatn_loadw) -- except for the padded-fourth-lane layout, whose window
// is fixed at 8 dwords: CHECK turns the test on for that instantiation only (it costs registers in the others).
template <bool CHECK>
__device__ __forceinline__ bool col_in_regs(const DevPlan& plan, uint32_t a, int loadw) {
  if (!CHECK) return true;
  const uint32_t off = plan.adaptive[a].offset, off0 = plan.ops[0].offset;
  return off >= off0 && off - off0 + plan.adaptive[a].bpv <= (uint32_t)loadw * 4u;
}
// ... and its SoA copy goes through the LDS staging area
template <bool CHECK>
__device__ __forceinline__ bool col_staged(const DevPlan& plan, uint32_t a, int loadw, int lanes) {
  return loadw > lanes && a < kStagedCols && plan.adaptive[a].bpv <= 4u && col_in_regs<CHECK>(plan, a, loadw);
}

// bytes [rel, rel + 8) of the dwords loaded for one point (rel + field size <= 4 * LOADW, guaranteed by the host)
// (three dwords: an 8-byte field at an offset that is no multiple of 4 spans them. Rounds 2-5 took two and shifted -- the top
// 1..3 bytes of such a field were lost; found by round 6's fuzz range 900000+, seed 923696: a UINT64 field at offset 25)
template <int LOADW>
__device__ __forceinline__ uint64_t field_from_regs(const FloatVec<LOADW>& pt, uint32_t rel) {
  const uint32_t di = rel >> 2;
  uint32_t d0 = 0u, d1 = 0u, d2 = 0u;
#pragma unroll
  for (int k = 0; k < LOADW; ++k) {
    if ((uint32_t)k == di) d0 = __float_as_uint(pt.v[k]);
    if ((uint32_t)k == di + 1u) d1 = __float_as_uint(pt.v[k]);
    if ((uint32_t)k == di + 2u) d2 = __float_as_uint(pt.v[k]);
  }
  const uint32_t mis = rel & 3u;
  return (((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, mis)) << 32) | __builtin_amdgcn_alignbyte(d1, d0, mis);
}

// UNAL: points are not 4-byte aligned (odd point_step / offset / base, e.g. packed 18-byte points): every lane loads
// LOADW + 1 dwords from the aligned address below its point and realigns them with v_alignbyte; the dwords may reach
// into the next point, so only the last points of the whole batch need the guarded path (points_end).
// L3: dword (behind the first lane) of the fourth lane -- 3 for x y z w back to back, 4 for the PCL / Ouster layout
// "x y z <pad> intensity" (the fused encoder takes any four offsets, src/field_encoder.cpp:24-40).
template <int T, int LANES, int PPT, uint32_t RING_BYTES, int LOADW, bool PREFETCH = true, bool UNAL = false, int L3 = 3>
__global__ __launch_bounds__(T) void k_encode_floatn(const DevPlan plan, const uint8_t* __restrict__ points,
                                                     const uint8_t* __restrict__ points_end,
                                                     const ChunkDesc* __restrict__ chunks,
                                                     uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                     Seg* __restrict__ segs, uint32_t segs_per_chunk,
                                                     const ColumnPtrs cols, uint32_t subs, uint32_t sub_points,
                                                     uint32_t sub_stride, uint32_t ablate) {
  constexpr int NW = T / 64;
  constexpr uint32_t ROW = NW * 63u;
  constexpr uint32_t TILE = ROW * PPT;
  static_assert(NW * PPT <= 64, "row/wave totals must fit one wave scan");
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* ring = reinterpret_cast<uint32_t*>(smem);
  uint32_t* wtot = reinterpret_cast<uint32_t*>(smem + RING_BYTES);
  uint8_t* colstage = smem + RING_BYTES + 256u;  // [kStagedCols][TILE * 4] SoA staging of the adaptive fields

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t chunk_id = blockIdx.x / subs;
  const uint32_t sub_id = blockIdx.x - chunk_id * subs;
  const ChunkDesc cd = chunks[chunk_id];
  const uint32_t step = plan.point_step;
  const uint32_t sub_first = sub_id * sub_points;  // first point of this workgroup's sub-chunk inside the chunk
  // points of this sub-chunk (saturating subtraction, see k_encode_regular)
  const int32_t n = (int32_t)min(sub_points, cd.n_points - min(cd.n_points, sub_first));
  const int32_t idx_lo = sub_first > 0u ? -1 : 0;  // the point before the sub-chunk is a valid delta reference
  const size_t first_point = (size_t)cd.first_point + sub_first;
  const uint8_t* gbase = points + first_point * step + plan.ops[0].offset;
  uint8_t* slot = slots + (size_t)chunk_id * slot_stride + (size_t)sub_id * sub_stride;
  if (n == 0) {  // sub-chunk beyond the end of a short chunk: nothing to read, empty segment
    if (tid == 0) {
      Seg s;
      s.off = sub_id * sub_stride;
      s.size = 0u;
      segs[(size_t)chunk_id * segs_per_chunk + sub_id] = s;
    }
    return;
  }
  float mult[LANES];
#pragma unroll
  for (int k = 0; k < LANES; ++k) mult[k] = plan.ops[k].mult_f;

  for (uint32_t i = tid; i < RING_BYTES / 16u; i += T) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0u, 0u, 0u, 0u);

  // LOADW >= LANES dwords are loaded per point (one global_load_dwordx3/x4/...): the extra dwords carry the
  // adaptive-int fields that live right behind the floats, so their AoS->SoA split needs no second load.
  FloatVec<LOADW> cur[PPT], nxt[PPT];
  auto load_tile = [&](uint32_t base, FloatVec<LOADW>(&dst)[PPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int32_t idx = (int32_t)(base + (uint32_t)(j * NW + (int)wave) * 63u + lane) - 1;
      FloatVec<LOADW> z;
#pragma unroll
      for (int k = 0; k < LOADW; ++k) z.v[k] = 0.0f;
      if (idx >= idx_lo && idx < n && !(ablate & 8u)) {
        const uint8_t* a = gbase + (ptrdiff_t)idx * (ptrdiff_t)step;
        if (!UNAL) {
          z = *reinterpret_cast<const FloatVec<LOADW>*>(a);
        } else {
          const uint32_t mis = (uint32_t)((uintptr_t)a & 3u);
          const uint32_t* q = reinterpret_cast<const uint32_t*>(a - mis);
          uint32_t d[LOADW + 1];
          if (reinterpret_cast<const uint8_t*>(q + LOADW + 1) <= points_end) {
#pragma unroll
            for (int k = 0; k <= LOADW; ++k) d[k] = q[k];
          } else {  // last points of the batch: a dword is read only if it holds at least one byte of the buffer
#pragma unroll
            for (int k = 0; k <= LOADW; ++k) d[k] = (reinterpret_cast<const uint8_t*>(q + k) < points_end) ? q[k] : 0u;
          }
#pragma unroll
          for (int k = 0; k < LOADW; ++k) z.v[k] = __uint_as_float(__builtin_amdgcn_alignbyte(d[k + 1], d[k], mis));
        }
      }
      dst[j] = z;
    }
  };
  load_tile(0u, cur);
  __syncthreads();

  // Two copies of the tile loop. Clouds either have NaNs all over (organised depth images) or none (lidar), so the
  // first tile decides for the sub-chunk: with NaNs around, rows that fail the short-path test get a second try that
  // treats NaN lanes as marker bytes; without, that code is not even in the loop (it cost the NaN-free loop 7 %).
  uint32_t R = 0u, F = 0u;
  auto run_tiles = [&](auto nan_tier_tag) __attribute__((always_inline)) {
  constexpr bool NAN_TIER = decltype(nan_tier_tag)::value;
  for (uint32_t base = 0; base < (uint32_t)n; base += TILE) {
    const bool last = (base + TILE >= (uint32_t)n);
    if (PREFETCH && !last) load_tile(base + TILE, nxt);  // double buffer: in flight while this tile is encoded

    // Tokens of the row's points. Common case (no NaN in the wave row, every token <= 4 bytes, i.e. |delta| <
    // 2^27 ticks): one dword per token, built with the short formulas and kept until the scan is done. Rows with a
    // NaN or a 5-byte token (wave-uniform test) use the general formulas and are rebuilt at emission time.
    uint32_t tok[PPT][LANES], lens[PPT], plen[PPT], incl[PPT];
    uint32_t rare_rows = 0u;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
      const int32_t idx = (int32_t)(base + (uint32_t)(j * NW + (int)wave) * 63u + lane) - 1;
      const bool emits = (lane > 0u) && (idx < n);
      lens[j] = 0u;
      uint32_t total = 0u;
      bool rare = false;
#pragma unroll
      for (int k = 0; k < LANES; ++k) {
        // Common case in the float domain: r = rndne(v * m) is an integer-valued float; with |r| < 2^21 on both
        // sides, r - r_prev, 2d + 0.5 and |.| + 0.5 are all exact, and zigzag(d) + 1 == |2d + 0.5| + 0.5. Anything
        // else (NaN, Inf, |r| >= 2^21 ticks) marks the row rare. The neighbour's r travels negated so that the DPP
        // move folds into a commutative add (hipcc's v_subrev_*_dpp returned the operands swapped on gfx950).
        const float r = rintf(__fmul_rn(cur[j].v[(LANES == 4 && k == 3) ? L3 : k], mult[k]));
        rare |= !(fabsf(r) < 2097152.0f);
        // nd = r_prev - r with the DPP value as the first operand (a plain v_sub_f32_dpp; the reversed form is the one
        // that misbehaved); |2d + 0.5| == |(-2) nd + 0.5|
        const float nd = __fsub_rn(__uint_as_float(dpp_wave_shr1(__float_as_uint(r))), r);
        const float uf = fabsf(__fmaf_rn(nd, -2.0f, 0.5f)) + 0.5f;
        const uint32_t u = (uint32_t)uf;
        const uint32_t l = groups7((uint32_t)__builtin_amdgcn_frexp_expf(uf));  // frexp exponent == bit length of u
        tok[j][k] = token4(u, l);
        lens[j] |= l << (8 * k);
        total += l;
      }
      if (__builtin_expect(__ballot(rare) != 0ull, 0)) {
        // Second tier: the row only has NaNs (organised clouds with invalid pixels) next to in-range values. A NaN
        // is the marker byte and hands its neighbour a reference of 0; everything else stays in the float domain
        // and the tokens still fit one dword, so the row keeps the short emission path.
        bool hard = !NAN_TIER;
        uint32_t total2 = 0u, lens2 = 0u;
#pragma unroll
        for (int k = 0; k < LANES; ++k) {
          if (!NAN_TIER) break;
          const float v = cur[j].v[(LANES == 4 && k == 3) ? L3 : k];
          const bool isn = is_nan_f32(v);
          const float r = isn ? 0.0f : rintf(__fmul_rn(v, mult[k]));
          hard |= !(fabsf(r) < 2097152.0f);
          const float nrp = __uint_as_float(dpp_wave_shr1(__float_as_uint(r) ^ 0x80000000u));
          const float uf = fabsf(__fmaf_rn(__fadd_rn(r, nrp), 2.0f, 0.5f)) + 0.5f;
          const uint32_t l = isn ? 1u : groups7((uint32_t)__builtin_amdgcn_frexp_expf(uf));
          tok[j][k] = isn ? 0u : token4((uint32_t)uf, l);
          lens2 |= l << (8 * k);
          total2 += l;
        }
        if (NAN_TIER && __ballot(hard) == 0ull) {
          lens[j] = lens2;
          total = total2;
        } else {  // Inf, overflow or a 5-byte token somewhere in the row: general integer formulas
          rare_rows |= 1u << j;
          total = 0u;
#pragma unroll
          for (int k = 0; k < LANES; ++k) {
            const float v = cur[j].v[(LANES == 4 && k == 3) ? L3 : k];
            const bool isn = is_nan_f32(v);
            const int32_t q = quant_rne_i32(v, mult[k]);
            const uint32_t nqp = dpp_wave_shr1(isn ? 0u : (0u - (uint32_t)q));  // a NaN resets that lane's reference
            uint32_t a0, a1, l;
            floatn_token(isn, (int32_t)((uint32_t)q + nqp), a0, a1, l);
            total += l;
          }
        }
      }
      plen[j] = emits ? total : 0u;
    }
    if (PPT == 2) {  // both rows' byte counts (< 2^16 each) ride one wave scan
      const uint32_t pincl = wave_inclusive_scan(plen[0] | (plen[PPT - 1] << 16));
      incl[0] = pincl & 0xffffu;
      incl[PPT - 1] = pincl >> 16;
    } else {
#pragma unroll
      for (int j = 0; j < PPT; ++j) incl[j] = wave_inclusive_scan(plen[j]);
    }
    if (lane == 63u) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) wtot[j * NW + (int)wave] = incl[j];
    }
    __syncthreads();
    const uint32_t wt = (lane < (uint32_t)(NW * PPT)) ? wtot[lane] : 0u;
    const uint32_t wincl = wave_inclusive_scan(wt);
    const uint32_t tile_total = (uint32_t)__builtin_amdgcn_readlane((int)wincl, NW * PPT - 1);
    const uint32_t r_end = R + tile_total;
    const uint32_t target = last ? ((r_end + 15u) & ~15u) : (r_end & ~15u);

    if (plan.n_adaptive && LOADW > LANES && !(ablate & 1u)) {
      for (uint32_t a = 0; a < plan.n_adaptive; ++a) {
        if (!col_staged<L3 == 4>(plan, a, LOADW, LANES) || (ablate & 16u)) continue;
        const uint32_t bpv = plan.adaptive[a].bpv;
        uint8_t* st = colstage + (size_t)a * (TILE * 4u);
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
          const int32_t idx = (int32_t)(base + (uint32_t)(j * NW + (int)wave) * 63u + lane) - 1;
          if (lane > 0u && idx < n) {
            const uint32_t t = (uint32_t)idx - base;  // point index inside the tile
            const uint32_t rel = plan.adaptive[a].offset - plan.ops[0].offset;
            uint64_t raw;
            if (LOADW == LANES + 1) raw = __float_as_uint(cur[j].v[LOADW - 1]) >> ((rel & 3u) * 8u);  // the one extra dword
            else raw = field_from_regs<LOADW>(cur[j], rel);
            if (bpv == 2u) reinterpret_cast<uint16_t*>(st)[t] = (uint16_t)raw;
            else reinterpret_cast<uint32_t*>(st)[t] = (uint32_t)raw;
          }
        }
      }
    }

    auto emit_all = [&](auto windowed, uint32_t win_lo_dw) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) {
        const int f = j * NW + (int)wave;
        const uint32_t rowbase = (f == 0) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)wincl, f - 1);
        uint32_t off = R + rowbase + incl[j] - plen[j];
        if (__builtin_expect((rare_rows & (1u << j)) != 0u, 0)) {  // wave-uniform: rebuild the general tokens (all lanes take part in the DPP)
#pragma unroll
          for (int k = 0; k < LANES; ++k) {
            const float v = cur[j].v[(LANES == 4 && k == 3) ? L3 : k];
            const bool isn = is_nan_f32(v);
            const int32_t q = quant_rne_i32(v, mult[k]);
            const uint32_t nqp = dpp_wave_shr1(isn ? 0u : (0u - (uint32_t)q));
            uint32_t a0, a1, l;
            floatn_token(isn, (int32_t)((uint32_t)q + nqp), a0, a1, l);
            if (plen[j] && !(ablate & 2u)) ring_put5<RING_BYTES, decltype(windowed)::value>(ring, off, a0, a1, l, win_lo_dw);
            off += l;
          }
        } else if (plen[j] && !(ablate & 2u)) {
#pragma unroll
          for (int k = 0; k < LANES; ++k) {
            const uint32_t t = tok[j][k];
            const uint32_t l = (lens[j] >> (8 * k)) & 0xffu;
            if (decltype(windowed)::value) ring_put5<RING_BYTES, true>(ring, off, t, 0u, l, win_lo_dw);
            else ring_put4<RING_BYTES>(ring, off, t);
            off += l;
          }
        }
      }
    };

    // AoS -> SoA split of the adaptive-int fields. Fields covered by the point load are taken from registers;
    // 2/4-byte fields are staged in LDS (written after the scan barrier) and leave as 16-byte stores.
    auto write_columns = [&]() __attribute__((always_inline)) {
      if (plan.n_adaptive && !(ablate & 1u)) {
        const uint32_t tile_pts = min(TILE, (uint32_t)n - base);
        for (uint32_t a = 0; a < plan.n_adaptive; ++a) {
          const uint32_t bpv = plan.adaptive[a].bpv;
          uint8_t* gcol = cols.p[a] + (first_point + base) * bpv;
          const bool staged = col_staged<L3 == 4>(plan, a, LOADW, LANES) && (((uintptr_t)gcol & 15u) == 0u) && !(ablate & 16u);
          if (staged) {
            const uint8_t* st = colstage + (size_t)a * (TILE * 4u);
            const uint32_t bytes = tile_pts * bpv;
            for (uint32_t u = tid; u < (bytes >> 4); u += T)
              reinterpret_cast<uint4*>(gcol)[u] = reinterpret_cast<const uint4*>(st)[u];
            const uint32_t tail0 = bytes & ~15u;
            if (tid < (bytes & 15u)) gcol[tail0 + tid] = st[tail0 + tid];
            continue;
          }
  #pragma unroll
          for (int j = 0; j < PPT; ++j) {
            const int32_t idx = (int32_t)(base + (uint32_t)(j * NW + (int)wave) * 63u + lane) - 1;
            if (lane > 0u && idx < n) {
              const size_t gi = first_point + (size_t)idx;
              uint8_t* col = cols.p[a];
              uint64_t raw;
              if (LOADW > LANES && col_in_regs<L3 == 4>(plan, a, LOADW)) {
                raw = field_from_regs<LOADW>(cur[j], plan.adaptive[a].offset - plan.ops[0].offset);
              } else {
                const uint8_t* fp = points + gi * step + plan.adaptive[a].offset;
                raw = 0u;
                if (((uintptr_t)fp & (bpv - 1u)) == 0u) {
                  if (bpv == 2u) raw = *reinterpret_cast<const uint16_t*>(fp);
                  else if (bpv == 4u) raw = *reinterpret_cast<const uint32_t*>(fp);
                  else raw = *reinterpret_cast<const uint64_t*>(fp);
                } else {
                  for (uint32_t b = 0; b < bpv; ++b) raw |= ((uint64_t)fp[b]) << (8u * b);
                }
              }
              if (bpv == 2u) reinterpret_cast<uint16_t*>(col)[gi] = (uint16_t)raw;
              else if (bpv == 4u) reinterpret_cast<uint32_t*>(col)[gi] = (uint32_t)raw;
              else reinterpret_cast<uint64_t*>(col)[gi] = raw;
            }
          }
        }
      }
    };

    // Without the double buffer the next tile's points are requested as soon as this tile's registers are dead:
    // the loads fly during the ring flush and the next scan barrier.
    if (__builtin_expect(r_end - F <= RING_BYTES, 1)) {
      emit_all(std::false_type{}, 0u);
      __syncthreads();
      write_columns();
      if (!PREFETCH && !last) load_tile(base + TILE, cur);
      if (!(ablate & 4u)) ring_flush_n<T, RING_BYTES>(ring, slot, F, target);
      F = target;
    } else {
      for (;;) {
        emit_all(std::true_type{}, F >> 2);
        __syncthreads();
        const uint32_t nf = min(F + RING_BYTES, target);
        ring_flush_n<T, RING_BYTES>(ring, slot, F, nf);
        const bool done = (F + RING_BYTES >= r_end);
        F = nf;
        if (done) break;
        __syncthreads();
      }
      write_columns();
      if (!PREFETCH && !last) load_tile(base + TILE, cur);
    }
    R = r_end;

    if (PREFETCH && !last) {
#pragma unroll
      for (int j = 0; j < PPT; ++j) cur[j] = nxt[j];
    }
    // No barrier here: the next tile's wtot writes sit behind this tile's second barrier (every wave has read
    // wtot by then) and its ring ORs sit behind its own first barrier (every wave has finished this flush).
  }
  };
  if (LANES == 3) {  // the 4-lane instantiations would lose a wave of occupancy to the second copy
    bool any_nan = false;
#pragma unroll
    for (int j = 0; j < PPT; ++j)
#pragma unroll
      for (int k = 0; k < LANES; ++k) any_nan |= is_nan_f32(cur[j].v[(LANES == 4 && k == 3) ? L3 : k]);
    if (__syncthreads_or(any_nan ? 1 : 0)) run_tiles(std::true_type{});
    else run_tiles(std::false_type{});
  } else {
    run_tiles(std::false_type{});
  }

  if (tid == 0) {
    Seg s;
    s.off = sub_id * sub_stride;
    s.size = R;
    segs[(size_t)chunk_id * segs_per_chunk + sub_id] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_gorilla_tokens: FieldEncoderFloat_Gorilla<double> (include/cloudini_lib/field_encoder.hpp:156-312). The codec
// keeps a (leading, trailing) bit window that only changes at "new window" points; everything else about a point
// (its XOR with the predecessor, the bits it writes once the window is known) is independent of the other points.
// One workgroup per chunk works in passes of kGorPass points:
//   1. all threads load their points and the predecessors, XOR, count leading / trailing zero bits -> LDS (2 bytes);
//   2. wave 0 alone walks the pass 64 points at a time: every lane assumes the current window, the lowest lane that
//      would open a new one is resolved, the window is updated, and only the lanes behind it re-check; each point's
//      window (and whether it opened it) goes back to LDS;
//   3. all threads build the bytes of their points (the reference flushes to a byte boundary per point) and store
//      them as 16-byte tokens for k_encode_regular to place.
// Only step 2 is serial, and it touches nothing but LDS. grid = (n_chunks, n_gorilla), block = kGorThreads.
// ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kGorThreads = 512;
constexpr uint32_t kGorPPT = 16;
constexpr uint32_t kGorPass = kGorThreads * kGorPPT;  // 8192 points per pass

// the 8 bytes at p (any alignment); `end` = first byte that must not be read
__device__ __forceinline__ uint64_t gor_load64(const uint8_t* p, const uint8_t* end) {
  const uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
  const uint32_t* q = reinterpret_cast<const uint32_t*>(p - mis);
  const uint32_t d0 = q[0], d1 = q[1];
  const uint32_t d2 = (mis != 0u && reinterpret_cast<const uint8_t*>(q + 2) < end) ? q[2] : 0u;
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, mis);
  const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, mis);
  return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(kGorThreads) void k_gorilla_tokens(const DevPlan plan, const uint8_t* __restrict__ points,
                                                                const uint8_t* __restrict__ points_end,
                                                                const ChunkDesc* __restrict__ chunks,
                                                                uint4* const* out_tokens) {
  __shared__ uint16_t lt[kGorPass];  // in: lead | trail << 8 (lead 64 = no difference); out: window + bit 15 "opens"
  // find the blockIdx.y-th Gorilla op
  uint32_t opi = 0, seen = 0;
  for (; opi < plan.n_ops; ++opi) {
    if (plan.ops[opi].kind == OP_GORILLA64) {
      if (seen == blockIdx.y) break;
      ++seen;
    }
  }
  const uint32_t field_off = plan.ops[opi].offset;
  const ChunkDesc cd = chunks[blockIdx.x];
  const uint32_t n = cd.n_points;
  const uint32_t step = plan.point_step;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint8_t* base = points + (size_t)cd.first_point * step + field_off;
  uint4* out = out_tokens[blockIdx.y] + cd.first_point;

  uint32_t win_lead = 255u, win_trail = 0u;  // kLeadingSentinel: no window yet (wave 0 keeps the state)
  for (uint32_t p0 = 0; p0 < n; p0 += kGorPass) {
    // ---- 1: XOR with the predecessor (point 0 of the chunk has none: written raw)
    uint64_t cur[kGorPPT], x[kGorPPT];
#pragma unroll
    for (uint32_t k = 0; k < kGorPPT; ++k) {
      const uint32_t i = p0 + k * kGorThreads + tid;
      cur[k] = 0u;
      x[k] = 0u;
      if (i < n) {
        const uint8_t* q = base + (size_t)i * step;
        cur[k] = gor_load64(q, points_end);
        const uint64_t prev = i ? gor_load64(q - step, points_end) : 0u;
        x[k] = cur[k] ^ prev;
      }
      const uint32_t lead = x[k] ? (uint32_t)__builtin_clzll(x[k]) : 64u;
      const uint32_t trail = x[k] ? (uint32_t)__builtin_ctzll(x[k]) : 0u;
      lt[k * kGorThreads + tid] = (uint16_t)(lead | (trail << 8));
    }
    __syncthreads();
    // ---- 2: the windows (wave 0)
    if (tid < 64u) {
      const uint32_t np = min(kGorPass, n - p0);
      for (uint32_t b0 = 0; b0 < np; b0 += 64u) {
        const uint32_t j = b0 + lane;
        const uint32_t e16 = lt[j];
        const uint32_t lead = e16 & 0xffu, trail = e16 >> 8;
        bool pending = j < np && (p0 + j) > 0u && lead != 64u;
        bool opens = false;
        uint32_t my_lead = win_lead, my_trail = win_trail;
        for (;;) {
          const bool would_open = pending && (win_lead == 255u || lead < win_lead || trail < win_trail);
          const uint64_t ev = __ballot(would_open);
          if (ev == 0ull) {
            if (pending) {
              my_lead = win_lead;
              my_trail = win_trail;
            }
            break;
          }
          const uint32_t e = (uint32_t)__builtin_ctzll(ev);
          if (pending && lane <= e) {
            my_lead = win_lead;
            my_trail = win_trail;
            opens = (lane == e);
            pending = false;
          }
          const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)lead, (int)e);
          const uint32_t te = (uint32_t)__builtin_amdgcn_readlane((int)trail, (int)e);
          win_lead = le > 31u ? 31u : le;
          win_trail = te;
        }
        // a point that does not open one writes inside (my_lead <= 31, my_trail <= 63); an opener uses its own counts
        lt[j] = (uint16_t)((my_lead & 0x3fu) | ((my_trail & 0x7fu) << 6) | (opens ? 0x8000u : 0u));
      }
    }
    __syncthreads();
    // ---- 3: the bytes of every point
#pragma unroll
    for (uint32_t k = 0; k < kGorPPT; ++k) {
      const uint32_t i = p0 + k * kGorThreads + tid;
      if (i < n) {
        const uint32_t w16 = lt[k * kGorThreads + tid];
        const uint64_t xx = x[k];
        uint64_t lo = 0u, hi = 0u;
        uint32_t nbits;
        if (i == 0u) {  // first value of the chunk: raw 64 bits
          lo = cur[k];
          nbits = 64u;
        } else if (xx == 0u) {
          nbits = 1u;  // single '0' bit
        } else if (!(w16 & 0x8000u)) {
          const uint32_t my_lead = w16 & 0x3fu, my_trail = (w16 >> 6) & 0x7fu;
          const uint32_t m = 64u - my_lead - my_trail;  // '1','0', m bits of (x >> trailing)
          const uint64_t payload = xx >> my_trail;
          lo = 1u | (payload << 2);
          hi = payload >> 62;
          nbits = 2u + m;
        } else {
          const uint32_t lead = (uint32_t)__builtin_clzll(xx), trail = (uint32_t)__builtin_ctzll(xx);
          const uint32_t sl = lead > 31u ? 31u : lead;  // '1','1', leading(5), m-1 (6), m bits of (x >> trailing)
          const uint32_t m = 64u - sl - trail;
          const uint64_t payload = xx >> trail;
          lo = 3u | ((uint64_t)sl << 2) | ((uint64_t)(m - 1u) << 7) | (payload << 13);
          hi = payload >> 51;
          nbits = 13u + m;
        }
        out[i] = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (nbits + 7u) >> 3);
      }
    }
    __syncthreads();  // lt is rewritten by the next pass
  }
}

// k_gorilla_windows: steps 1 and 2 of k_gorilla_tokens for layouts whose Gorilla field the piece kernel encodes itself
// (TAIL instantiations of k_encode_fused): all it leaves behind is the window in effect in front of every piece of
// `piece_pts` points -- out[chunk * kGorWinStride + piece] = lead | trail << 8 (lead 255 = no window yet) -- instead of
// a 16-byte token per point written here and read back there. grid = (n_chunks), block = kGorThreads.
constexpr uint32_t kGorWinStride = 128;  // >= pieces per chunk (32768 / 378 = 87)

// min over the wave, valid in lane 63 (DPP row shifts and row broadcasts; lanes without a source keep their own value)
__device__ __forceinline__ uint32_t gor_wave_min(uint32_t x) {
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x111, 0xf, 0xf, false));  // row_shr:1
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x112, 0xf, 0xf, false));  // row_shr:2
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x114, 0xf, 0xf, false));  // row_shr:4
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x118, 0xf, 0xf, false));  // row_shr:8
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x142, 0xa, 0xf, false));  // row_bcast:15
  x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x143, 0xc, 0xf, false));  // row_bcast:31
  return x;
}

__global__ __launch_bounds__(kGorThreads) void k_gorilla_windows(const DevPlan plan, uint32_t opi, uint32_t piece_pts,
                                                                 const uint8_t* __restrict__ points,
                                                                 const uint8_t* __restrict__ points_end,
                                                                 const ChunkDesc* __restrict__ chunks,
                                                                 uint16_t* __restrict__ win_out) {
  constexpr uint32_t NB = kGorPass / 64u;  // batches of 64 points per pass: one per (k, wave)
  static_assert(NB <= 128u, "two batch summaries per lane of wave 0");
  __shared__ uint16_t lt[kGorPass];
  __shared__ uint32_t bmin[NB];  // per batch: min lead | min trail << 8 over its points that differ from their predecessor
  const uint32_t field_off = plan.ops[opi].offset;
  const ChunkDesc cd = chunks[blockIdx.x];
  const uint32_t n = cd.n_points;
  const uint32_t step = plan.point_step;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint8_t* base = points + (size_t)cd.first_point * step + field_off;
  uint16_t* out = win_out + (size_t)blockIdx.x * kGorWinStride;

  uint32_t win_lead = 255u, win_trail = 0u;
  // (round 6) a pass's 16 values per thread are requested up front, without a branch around any load (the two conditional
  // loads per point -- the value and its predecessor, each with a conditional third dword -- drained the memory pipe sixteen
  // times per pass and fetched every line twice); the predecessor is the neighbouring lane's value (DPP), a wave's first lane
  // takes it from the last lane of the wave in front (LDS, one barrier per pass)
  __shared__ unsigned long long wlast[kGorPPT * (kGorThreads / 64u) + 1u];  // [1 + k * 8 + wave]: value of that wave's lane 63; [0]: the pass in front
  if (tid == 0u) wlast[0] = 0ull;
  for (uint32_t p0 = 0; p0 < n; p0 += kGorPass) {
    uint64_t cur[kGorPPT];
#pragma unroll
    for (uint32_t k = 0; k < kGorPPT; ++k) {
      const uint32_t i = p0 + k * kGorThreads + tid;
      const uint8_t* q8 = base + (size_t)(i < n ? i : n - 1u) * step;  // (lanes behind the chunk: a valid address, the value is not used)
      const uint32_t mis = (uint32_t)((uintptr_t)q8 & 3u);
      const uint32_t* q = reinterpret_cast<const uint32_t*>(q8 - mis);
      const bool third = reinterpret_cast<const uint8_t*>(q + 2) < points_end;  // (a dword is read only if it holds a byte of the buffer)
      const uint32_t d0 = q[0], d1 = q[1], d2r = q[third ? 2 : 0];
      const uint32_t d2 = third ? d2r : 0u;
      cur[k] = (((uint64_t)__builtin_amdgcn_alignbyte(d2, d1, mis)) << 32) | __builtin_amdgcn_alignbyte(d1, d0, mis);
    }
#pragma unroll
    for (uint32_t k = 0; k < kGorPPT; ++k)
      if (lane == 63u) wlast[1u + k * (kGorThreads / 64u) + wave] = cur[k];
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < kGorPPT; ++k) {
      const uint32_t i = p0 + k * kGorThreads + tid;
      const uint64_t front = wlast[k * (kGorThreads / 64u) + wave];  // (lane 63 of the wave in front; [0]: the last value of the pass before)
      const uint32_t plo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)front, (int)(uint32_t)cur[k], 0x138, 0xf, 0xf, false);          // wave_shr:1
      const uint32_t phi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(front >> 32), (int)(uint32_t)(cur[k] >> 32), 0x138, 0xf, 0xf, false);
      const uint64_t prev = (((uint64_t)phi) << 32) | plo;
      const uint64_t x = (i < n && i != 0u) ? (cur[k] ^ prev) : 0ull;
      const bool counts = x != 0u;  // may open a window
      const uint32_t lead = x ? (uint32_t)__builtin_clzll(x) : 64u;
      const uint32_t trail = x ? (uint32_t)__builtin_ctzll(x) : 0u;
      lt[k * kGorThreads + tid] = (uint16_t)(lead | (trail << 8));
      // the batch of this (k, wave): the smallest counts any of its points brings -- a window at least that wide on both
      // sides is left alone by the whole batch, and the serial walk below skips it
      uint32_t ml = counts ? lead : 255u, mt = counts ? trail : 255u;
      ml = gor_wave_min(ml);
      mt = gor_wave_min(mt);
      if (lane == 63u) bmin[k * (kGorThreads / 64u) + wave] = ml | (mt << 8);
    }
    __syncthreads();
    if (tid == kGorThreads - 1u) wlast[0] = cur[kGorPPT - 1u];  // (read again behind the next pass's barrier)
    if (tid < 64u) {
      const uint32_t np = min(kGorPass, n - p0);
      const uint32_t my_bmin0 = lane < NB ? bmin[lane] : 0xffffu;
      const uint32_t my_bmin1 = lane + 64u < NB ? bmin[lane + 64u] : 0xffffu;
      for (uint32_t b = 0; b * 64u < np; ++b) {
        const uint32_t b0 = b * 64u;
        const uint32_t bm = (uint32_t)__builtin_amdgcn_readlane((int)(b < 64u ? my_bmin0 : my_bmin1), (int)(b & 63u));  // uniform
        const uint32_t entry_lead = win_lead, entry_trail = win_trail;
        const bool quiet = (bm & 0xffu) == 255u || (win_lead != 255u && (bm & 0xffu) >= win_lead && (bm >> 8) >= win_trail);
        uint64_t opened = 0ull;
        uint32_t lead = 0u, trail = 0u;
        if (!quiet) {
          const uint32_t j = b0 + lane;
          const uint32_t e16 = lt[j];
          lead = e16 & 0xffu;
          trail = e16 >> 8;
          bool pending = j < np && (p0 + j) > 0u && lead != 64u;
          for (;;) {
            const bool would_open = pending && (win_lead == 255u || lead < win_lead || trail < win_trail);
            const uint64_t ev = __ballot(would_open);
            if (ev == 0ull) break;
            const uint32_t e = (uint32_t)__builtin_ctzll(ev);
            opened |= 1ull << e;
            if (lane <= e) pending = false;
            const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)lead, (int)e);
            const uint32_t te = (uint32_t)__builtin_amdgcn_readlane((int)trail, (int)e);
            win_lead = le > 31u ? 31u : le;
            win_trail = te;
          }
        }
        // a piece that starts inside this batch: the window in front of its first point is the one the last opener
        // below that lane set, or the one the batch was entered with
        const uint32_t i0 = p0 + b0;                                   // chunk index of lane 0's point
        const uint32_t pc = (i0 + piece_pts - 1u) / piece_pts;         // first piece starting at or behind i0
        const uint32_t bnd = pc * piece_pts;
        if (bnd < i0 + 64u && bnd < n) {                               // uniform
          const uint32_t lb = bnd - i0;
          const uint64_t below = opened & ((1ull << lb) - 1ull);
          uint32_t wl = entry_lead, wt = entry_trail;
          if (below != 0ull) {
            const int e = 63 - (int)__builtin_clzll(below);
            const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)lead, e);
            wl = le > 31u ? 31u : le;
            wt = (uint32_t)__builtin_amdgcn_readlane((int)trail, e);
          }
          if (lane == 0u) out[pc] = (uint16_t)(wl | (wt << 8));
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_chunk_offsets: payload size of every chunk, exclusive scan of the framed sizes (4 + payload), per-cloud
// stream offsets. One workgroup.
// ------------------------------------------------------------------------------------------------------------

template <int T>
__global__ __launch_bounds__(T) void k_chunk_offsets(const Seg* __restrict__ segs, uint32_t segs_per_chunk,
                                                     uint32_t n_chunks, const uint32_t* __restrict__ cloud_first_chunk,
                                                     uint32_t n_clouds, uint32_t* __restrict__ chunk_payload,
                                                     uint64_t* __restrict__ chunk_dst,
                                                     uint64_t* __restrict__ stream_offsets) {
  __shared__ uint32_t wtot[32];
  uint64_t running = 0;
  for (uint32_t base = 0; base < n_chunks; base += T) {
    const uint32_t c = base + threadIdx.x;
    uint32_t framed = 0u;
    if (c < n_chunks) {
      uint32_t payload = 0u;
      for (uint32_t s = 0; s < segs_per_chunk; ++s) payload += segs[(size_t)c * segs_per_chunk + s].size;
      chunk_payload[c] = payload;
      framed = payload + 4u;
    }
    uint32_t total;
    const uint32_t excl = block_exclusive_scan<T>(framed, wtot, &total);
    if (c < n_chunks) chunk_dst[c] = running + excl;
    running += total;
    __syncthreads();
  }
  __syncthreads();
  // stream offset of cloud k = destination of its first chunk (clouds without chunks inherit the next one)
  for (uint32_t k = threadIdx.x; k <= n_clouds; k += T) {
    const uint32_t fc = (k < n_clouds) ? cloud_first_chunk[k] : n_chunks;
    stream_offsets[k] = (fc < n_chunks) ? chunk_dst[fc] : running;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_compact: final framed stream. grid = (n_chunks, splits). The chunk's segment table is read once into LDS and
// cut into work items of at most kCompactItemUnits 16-byte units; the waves of the chunk's workgroups take items
// round-robin, so many small segments (sub-chunked small batches) and one large segment (huge batches) both keep
// every wave busy without a dependent global load per segment. Source segments start 16-byte aligned; the
// destination position is arbitrary.
// ------------------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t funnel_bytes(uint32_t lo, uint32_t hi, uint32_t sb) {
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sb * 8u));
}

constexpr uint32_t kCompactMaxSegs = 96u + 2u * kMaxAdaptive;  // sub-chunk or piece segments + two per section
constexpr uint32_t kCompactItemUnits = 256u;  // 4 KiB per item
constexpr uint32_t kCompactMaxItems = 1024u;

template <int T>
__global__ __launch_bounds__(T) void k_compact(const uint8_t* __restrict__ slots, uint64_t slot_stride,
                                               const Seg* __restrict__ segs, uint32_t segs_per_chunk,
                                               const uint32_t* __restrict__ chunk_payload,
                                               const uint64_t* __restrict__ chunk_dst, uint8_t* __restrict__ out,
                                               uint64_t out_capacity, uint32_t* __restrict__ status) {
  __shared__ Seg seg_l[kCompactMaxSegs];
  __shared__ uint32_t doff_l[kCompactMaxSegs];               // destination offset of every segment behind the size word
  __shared__ uint32_t item_seg[kCompactMaxItems], item_u0[kCompactMaxItems];
  __shared__ uint32_t n_items_l;
  const uint32_t c = blockIdx.x;
  const uint32_t payload = chunk_payload[c];
  // work items of 4 KiB, larger when the chunk is so big (wide raw-copied points: up to 32768 * 1024 bytes) that 4 KiB
  // items would not fit the table
  const uint32_t item_units = max(kCompactItemUnits, (payload >> 4) / (kCompactMaxItems - kCompactMaxSegs - 8u) + 1u);
  const uint64_t dst0 = chunk_dst[c];
  if (dst0 + 4u + payload > out_capacity) {
    if (threadIdx.x == 0 && blockIdx.y == 0) atomicOr(status, (uint32_t)ST_OUT_OVERFLOW);
    return;
  }
  if (blockIdx.y == 0 && threadIdx.x < 4u) out[dst0 + threadIdx.x] = (uint8_t)(payload >> (8u * threadIdx.x));
  if (threadIdx.x < segs_per_chunk) seg_l[threadIdx.x] = segs[(size_t)c * segs_per_chunk + threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t d = 0u, n_items = 0u;
    for (uint32_t s = 0; s < segs_per_chunk; ++s) {
      doff_l[s] = d;
      const uint32_t size = seg_l[s].size;
      if (size) {
        // items cover the destination-aligned 16-byte units of the segment (+ one item for a segment without any)
        const uint32_t head = min(size, (uint32_t)((16u - (uint32_t)((dst0 + 4u + d) & 15u)) & 15u));
        const uint32_t units = (size - head) >> 4;
        uint32_t u0 = 0u;
        do {
          if (n_items < kCompactMaxItems) {
            item_seg[n_items] = s;
            item_u0[n_items] = u0;
            ++n_items;
          }
          u0 += item_units;
        } while (u0 < units);
      }
      d += size;
    }
    n_items_l = n_items;
  }
  __syncthreads();

  const uint8_t* slot = slots + (size_t)c * slot_stride;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wid = blockIdx.y * (T / 64) + (threadIdx.x >> 6);
  const uint32_t n_waves = gridDim.y * (T / 64);
  const uint32_t n_items = n_items_l;
  for (uint32_t it = wid; it < n_items; it += n_waves) {
    const uint32_t sidx = item_seg[it], u0 = item_u0[it];
    const Seg sg = seg_l[sidx];
    const uint8_t* src = slot + sg.off;
    uint8_t* dst = out + dst0 + 4u + doff_l[sidx];
    const uint32_t size = sg.size;
    // head: bytes until dst is 16-byte aligned; tail: bytes behind the last whole unit (first item of the segment)
    const uint32_t head = min(size, (uint32_t)((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u));
    const uint32_t body_units = (size - head) >> 4;
    const uint32_t tail = (size - head) & 15u;
    if (u0 == 0u) {
      if (lane < head) dst[lane] = src[lane];
      if (lane >= 32u && lane < 32u + tail) {
        const uint32_t k = head + body_units * 16u + (lane - 32u);
        dst[k] = src[k];
      }
    }
    // body: dst-aligned 16-byte units; source bytes [head + 16j, head + 16j + 16) straddle two aligned units
    const uint32_t sdw = head >> 2, sb = head & 3u;
    const uint4* src4 = reinterpret_cast<const uint4*>(src);
    uint4* dst4 = reinterpret_cast<uint4*>(dst + head);
    const uint32_t u1 = min(body_units, u0 + item_units);
    for (uint32_t j = u0 + lane; j < u1; j += 64u) {
      const uint4 a = src4[j];
      uint4 b = make_uint4(0u, 0u, 0u, 0u);
      if (head != 0u) b = src4[j + 1u];
      uint32_t w0, w1, w2, w3, w4;
      switch (sdw) {
        case 0: w0 = a.x; w1 = a.y; w2 = a.z; w3 = a.w; w4 = b.x; break;
        case 1: w0 = a.y; w1 = a.z; w2 = a.w; w3 = b.x; w4 = b.y; break;
        case 2: w0 = a.z; w1 = a.w; w2 = b.x; w3 = b.y; w4 = b.z; break;
        default: w0 = a.w; w1 = b.x; w2 = b.y; w3 = b.z; w4 = b.w; break;
      }
      uint4 o;
      o.x = funnel_bytes(w0, w1, sb);
      o.y = funnel_bytes(w1, w2, sb);
      o.z = funnel_bytes(w2, w3, sb);
      o.w = funnel_bytes(w3, w4, sb);
      dst4[j] = o;
    }
  }
}

}  // namespace cldn

// ------------------------------------------------------------------------------------------------------------
// V5 adaptive-int sections (src/v5_codec.cpp:258-491). One workgroup per (chunk, adaptive field); input is the
// SoA column written by k_encode_regular. The same device routines serve k_probe_modes (sizes only, over the
// first <= 4096 values of a cloud) and k_encode_sections (bytes, over a chunk).
// ------------------------------------------------------------------------------------------------------------

namespace cldn {

}  // namespace cldn

namespace cldn {

// ---------------------------------------------------------------------------------------------------------
// k_encode_fixed (round 4): regular streams whose per-point encoders all write a FIXED number of bytes --
// FieldEncoderFloat_XOR<float / double> (EncodingOptions::LOSSLESS: bits(cur) ^ bits(prev), prev = 0 at a chunk's first
// point, include/cloudini_lib/field_encoder.hpp:359-370) and FieldEncoderCopy (:56-60). Point i of a chunk then lies at byte
// i * P of the chunk's stream, P = the sum of the field sizes: no lengths, no scan, one thread per point (which also writes the
// point's integer fields to their SoA columns, the section kernels' input). The general kernel
// (op interpreter over an LDS tile, two passes) ran a lossless XYZI batch at 1.7 TB/s.
// grid (chunks, 32768 / 256) x 256. The stream leaves as the `subs` sub-streams the slot layout of the call reserves.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t fixed_load(const uint8_t* q, uint32_t nbytes) {
  uint64_t v = 0u;
  if (nbytes == 4u) {
    uint32_t w;
    __builtin_memcpy(&w, q, 4);
    v = w;
  } else if (nbytes == 8u) {
    __builtin_memcpy(&v, q, 8);
  } else if (nbytes == 2u) {
    uint16_t h;
    __builtin_memcpy(&h, q, 2);
    v = h;
  } else {
    for (uint32_t b = 0; b < nbytes; ++b) v |= (uint64_t)q[b] << (8u * b);
  }
  return v;
}
__device__ __forceinline__ void fixed_store(uint8_t* q, uint64_t v, uint32_t nbytes) {
  if (nbytes == 4u) {
    const uint32_t w = (uint32_t)v;
    __builtin_memcpy(q, &w, 4);
  } else if (nbytes == 8u) {
    __builtin_memcpy(q, &v, 8);
  } else if (nbytes == 2u) {
    const uint16_t h = (uint16_t)v;
    __builtin_memcpy(q, &h, 2);
  } else {
    for (uint32_t b = 0; b < nbytes; ++b) q[b] = (uint8_t)(v >> (8u * b));
  }
}

__global__ __launch_bounds__(256) void k_encode_fixed(const DevPlan plan, const uint8_t* __restrict__ points,
                                                      const ChunkDesc* __restrict__ chunks, uint8_t* __restrict__ slots,
                                                      uint64_t slot_stride, Seg* __restrict__ segs, uint32_t segs_per_chunk,
                                                      uint32_t subs, uint32_t sub_points, uint32_t sub_stride, uint32_t point_bytes,
                                                      const ColumnPtrs cols, uint8_t* __restrict__ direct_out,
                                                      uint32_t* __restrict__ chunk_payload, uint64_t* __restrict__ chunk_dst) {
  const uint32_t c = blockIdx.x;
  const uint32_t i = blockIdx.y * 256u + threadIdx.x;  // point of the chunk
  const ChunkDesc cd = chunks[c];
  const uint32_t n = cd.n_points;
  const uint32_t step = plan.point_step;
  uint8_t* slot = slots + (size_t)c * slot_stride;
  // DIRECT PLACEMENT (direct_out != NULL; schemas without integer columns): every chunk's payload is n * P bytes, so chunk c of
  // the batch begins at byte 4 c + P * (points in front of it) of the framed streams -- the bytes go straight to their final
  // place, [u32 size] included: no slot, no k_finish
  const uint64_t d0 = 4ull * c + (uint64_t)point_bytes * cd.first_point;
  if (direct_out != nullptr) {
    if (i == 0u) {
      const uint32_t payload = n * point_bytes;
      __builtin_memcpy(direct_out + d0, &payload, 4);
      chunk_payload[c] = payload;
      chunk_dst[c] = d0;
    }
  } else if (i < subs) {  // segment s: the points [s, s + 1) * sub_points of the chunk
    const uint32_t first = i * sub_points;
    Seg sg;
    sg.off = i * sub_stride;
    sg.size = (n > first ? min(sub_points, n - first) : 0u) * point_bytes;
    segs[(size_t)c * segs_per_chunk + i] = sg;
  }
  if (i >= n) return;
  const uint8_t* src = points + ((size_t)cd.first_point + i) * step;
  // the integer fields of the schema (V5 sections): AoS -> SoA columns for the section kernels (uniform loop)
  for (uint32_t a = 0; a < plan.n_adaptive; ++a) {
    const uint32_t bpv = plan.adaptive[a].bpv;
    fixed_store(cols.p[a] + ((size_t)cd.first_point + i) * bpv, fixed_load(src + plan.adaptive[a].offset, bpv), bpv);
  }
  const uint32_t s = i / sub_points;
  uint8_t* dst = direct_out != nullptr ? direct_out + d0 + 4u + (size_t)i * point_bytes
                                       : slot + (size_t)s * sub_stride + (size_t)(i - s * sub_points) * point_bytes;
  // four 32-bit XOR fields back to back in a 16-byte point (lossless XYZI): whole-point loads and one store
  const bool quad = step == 16u && point_bytes == 16u && plan.n_ops == 4u && plan.ops[0].kind == OP_XOR32 &&
                    plan.ops[1].kind == OP_XOR32 && plan.ops[2].kind == OP_XOR32 && plan.ops[3].kind == OP_XOR32 &&
                    plan.ops[0].offset == 0u && plan.ops[1].offset == 4u && plan.ops[2].offset == 8u && plan.ops[3].offset == 12u;  // (uniform)
  if (quad) {
    uint4 cur, pv = make_uint4(0u, 0u, 0u, 0u);
    __builtin_memcpy(&cur, src, 16);
    if (i != 0u) __builtin_memcpy(&pv, src - 16, 16);
    const uint4 o = make_uint4(cur.x ^ pv.x, cur.y ^ pv.y, cur.z ^ pv.z, cur.w ^ pv.w);
    __builtin_memcpy(dst, &o, 16);
    return;
  }
  uint32_t at = 0u;
  for (uint32_t k = 0; k < plan.n_ops; ++k) {  // (uniform)
    const uint32_t size = plan.ops[k].size, off = plan.ops[k].offset;
    uint64_t v = fixed_load(src + off, size);
    if (plan.ops[k].kind != OP_COPY && i != 0u) v ^= fixed_load(src - step + off, size);
    fixed_store(dst + at, v, size);
    at += size;
  }
}

// stream offset of every cloud for the direct placement of k_encode_fixed: where its first chunk begins (a cloud without chunks:
// where the next one does), and the batch's end
__global__ __launch_bounds__(256) void k_fixed_offsets(const ChunkDesc* __restrict__ chunks, const uint32_t* __restrict__ cloud_first_chunk,
                                                       uint32_t n_clouds, uint32_t n_chunks, uint32_t point_bytes, uint64_t total,
                                                       uint64_t* __restrict__ stream_offsets) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  if (k > n_clouds) return;
  const uint32_t fc = k < n_clouds ? cloud_first_chunk[k] : n_chunks;
  stream_offsets[k] = fc < n_chunks ? 4ull * fc + (uint64_t)point_bytes * chunks[fc].first_point : total;
}

}  // namespace cldn

#include "stage1_sections.h"
#include "stage1_fused.h"
#include "stage1_finish.h"

namespace cldn {

constexpr int kSecThreads = 1024;
constexpr uint32_t kPalSlots = 8192;      // LDS hash table slots (keys u64 + first-index u16)
constexpr uint32_t kPalCapacity = 6144;   // distinct values one table pass accepts (load factor 0.75)
constexpr uint32_t kPalEmpty = 0xffffu;

__device__ __forceinline__ uint64_t col_raw(const uint8_t* col, uint32_t bpv, uint32_t i) {
  if (bpv == 2u) return reinterpret_cast<const uint16_t*>(col)[i];
  if (bpv == 4u) return reinterpret_cast<const uint32_t*>(col)[i];
  return reinterpret_cast<const uint64_t*>(col)[i];
}

// Append up to two tokens per thread (a then b) to the stream, in thread order. EMIT=false only counts.
template <int T, bool EMIT>
__device__ __forceinline__ void stream_put2(StreamState& ss, uint32_t* ring, uint32_t* wtot, uint8_t* dst,
                                            const Tok a, const Tok b) {
  const uint32_t my_len = a.len + b.len;
  uint32_t total;
  const uint32_t excl = block_exclusive_scan<T>(my_len, wtot, &total);
  const uint32_t r_end = ss.R + total;
  if (EMIT) {
    const uint32_t target = r_end & ~15u;
    if (r_end - ss.F <= kRingBytes) {
      if (a.len) ring_put<false>(ring, ss.R + excl, a, 0u);
      if (b.len) ring_put<false>(ring, ss.R + excl + a.len, b, 0u);
      __syncthreads();
      ring_flush<T>(ring, dst, ss.F, target);
      ss.F = target;
    } else {
      for (;;) {
        if (a.len) ring_put<true>(ring, ss.R + excl, a, ss.F >> 2);
        if (b.len) ring_put<true>(ring, ss.R + excl + a.len, b, ss.F >> 2);
        __syncthreads();
        const uint32_t nf = min(ss.F + kRingBytes, target);
        ring_flush<T>(ring, dst, ss.F, nf);
        const bool done = (ss.F + kRingBytes >= r_end);
        ss.F = nf;
        if (done) break;
        __syncthreads();
      }
    }
  }
  ss.R = r_end;
  __syncthreads();
}

// flush the last partial 16-byte unit (the slot has slack behind every stream)
template <int T>
__device__ __forceinline__ void stream_finish(StreamState& ss, uint32_t* ring, uint8_t* dst) {
  const uint32_t target = (ss.R + 15u) & ~15u;
  ring_flush<T>(ring, dst, ss.F, target);
  ss.F = target;
  __syncthreads();
}

// ---- mode 0: DeltaVarint (appendDeltaVarintSection, v5_codec.cpp:423-432) ---------------------------------
template <int T, bool EMIT>
__device__ uint32_t section_delta_varint(const uint8_t* col, uint32_t bpv, uint32_t type, uint32_t n, uint32_t* ring,
                                         uint32_t* wtot, uint8_t* dst) {
  StreamState ss;
  ss.R = 1u;  // mode byte 0x00: the ring is zero-initialised
  ss.F = 0u;
  for (uint32_t base = 0; base < n; base += T) {
    const uint32_t i = base + threadIdx.x;
    Tok t = nan_tok();
    t.len = 0;
    if (i < n) {
      const int64_t v = int_field_as_i64(col_raw(col, bpv, i), type);
      const int64_t pv = i ? int_field_as_i64(col_raw(col, bpv, i - 1u), type) : 0;
      const int64_t d = (int64_t)((uint64_t)v - (uint64_t)pv);
      if (EMIT) t = varint64_tok(d);
      else t.len = varint64_len(d);
    }
    Tok none = t;
    none.len = 0;
    stream_put2<T, EMIT>(ss, ring, wtot, dst, t, none);
  }
  if (EMIT) stream_finish<T>(ss, ring, dst);
  return ss.R;
}

// ---- modes 2/3: Rle and DeltaRle (appendRleSection :471-491, appendDeltaRleSection :447-460) ---------------
// A run closes when the next run head is seen; heads are compacted into an LDS list per tile, entry 0 being
// the run left open by the previous tile.
template <int T, bool EMIT, bool DELTA>
__device__ uint32_t section_runs(const uint8_t* col, uint32_t bpv, uint32_t type, uint32_t n, uint32_t* ring,
                                 uint32_t* wtot, uint8_t* dst, uint32_t* list_pos, uint64_t* list_key) {
  StreamState ss;
  ss.R = 5u;  // [mode][u32 run_count]; both patched in after the runs are known
  ss.F = 0u;
  uint32_t run_count = 0u;
  uint32_t carry_pos = 0u;
  uint64_t carry_key = 0u;
  for (uint32_t base = 0; base < n; base += T) {
    const uint32_t i = base + threadIdx.x;
    const bool last_tile = (base + T >= n);
    uint64_t key = 0u;
    bool head = false;
    if (i < n) {
      if (DELTA) {  // keys are the first differences, values[-1] = 0 (forEachDeltaRun, :269-288)
        const int64_t v = int_field_as_i64(col_raw(col, bpv, i), type);
        const int64_t p1 = i >= 1u ? int_field_as_i64(col_raw(col, bpv, i - 1u), type) : 0;
        const int64_t p2 = i >= 2u ? int_field_as_i64(col_raw(col, bpv, i - 2u), type) : 0;
        key = (uint64_t)v - (uint64_t)p1;
        head = (i == 0u) || (key != ((uint64_t)p1 - (uint64_t)p2));
      } else {
        key = col_raw(col, bpv, i);
        head = (i == 0u) || (key != col_raw(col, bpv, i - 1u));
      }
    }
    uint32_t heads;
    const uint32_t rank = block_exclusive_scan<T>(head ? 1u : 0u, wtot, &heads);
    const uint32_t carry = base ? 1u : 0u;
    if (threadIdx.x == 0 && carry) {
      list_pos[0] = carry_pos;
      list_key[0] = carry_key;
    }
    if (head) {
      list_pos[carry + rank] = i;
      list_key[carry + rank] = key;
    }
    const uint32_t entries = carry + heads;
    if (threadIdx.x == 0 && last_tile) list_pos[entries] = n;  // sentinel closing the final run
    __syncthreads();
    const uint32_t n_emit = last_tile ? entries : entries - 1u;  // entries >= 1 always (point 0 is a head)
    for (uint32_t e0 = 0; e0 < n_emit; e0 += T) {
      const uint32_t e = e0 + threadIdx.x;
      Tok a = nan_tok(), b = nan_tok();
      a.len = 0;
      b.len = 0;
      if (e < n_emit) {
        const uint32_t run_len = list_pos[e + 1u] - list_pos[e];
        const uint64_t k = list_key[e];
        if (EMIT) {
          a = DELTA ? varint64_tok((int64_t)k) : raw_tok(k, bpv);
          b = uvarint32_tok(run_len);
        } else {
          a.len = DELTA ? varint64_len((int64_t)k) : bpv;
          b.len = uvarint32_len(run_len);
        }
      }
      stream_put2<T, EMIT>(ss, ring, wtot, dst, a, b);
    }
    run_count += n_emit;
    carry_pos = list_pos[entries - 1u];
    carry_key = list_key[entries - 1u];
    __syncthreads();
  }
  if (EMIT) {
    stream_finish<T>(ss, ring, dst);
    // patch [mode][run_count]; every earlier store of this workgroup has completed (barrier above)
    if (threadIdx.x < 5u) {
      const uint32_t v = threadIdx.x == 0u ? (DELTA ? 3u : 2u) : ((run_count >> (8u * (threadIdx.x - 1u))) & 0xffu);
      dst[threadIdx.x] = (uint8_t)v;
    }
  }
  return ss.R;
}

// ---- mode 1: Palette (appendPaletteSection :462-469, buildPaletteIndexes :369-379) -------------------------
struct PalTable {
  uint64_t* keys;   // [kPalSlots]
  uint16_t* first;  // [kPalSlots] index (within the chunk) of the first occurrence, kPalEmpty = free
};

__device__ __forceinline__ uint32_t pal_probe(const PalTable t, uint64_t v) {
  uint32_t slot = hash_u64(v) & (kPalSlots - 1u);
  for (;;) {
    const uint32_t f = t.first[slot];
    if (f == kPalEmpty) return kPalEmpty;
    if (t.keys[slot] == v) return f;
    slot = (slot + 1u) & (kPalSlots - 1u);
  }
}

// Build the table over values [0, n) whose hash partition is `part` (of `parts`), in index order; returns the
// number of distinct values, or 0xffffffff when the table overflowed. If first_out != nullptr, first_out[i]
// receives the first-occurrence index of value i (for the values of this partition).
template <int T>
__device__ uint32_t palette_pass(const uint8_t* col, uint32_t bpv, uint32_t n, uint32_t part, uint32_t parts,
                                 PalTable tab, uint64_t* tile_vals, uint64_t* miss_mask, uint32_t* flags,
                                 uint16_t* first_out) {
  constexpr int NW = T / 64;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (uint32_t s = threadIdx.x; s < kPalSlots; s += T) tab.first[s] = (uint16_t)kPalEmpty;
  if (threadIdx.x == 0) {
    flags[0] = 0u;  // any miss in this tile
    flags[1] = 0u;  // distinct count
  }
  __syncthreads();
  for (uint32_t base = 0; base < n; base += T) {
    const uint32_t i = base + threadIdx.x;
    uint64_t v = 0u;
    bool mine = false;
    if (i < n) {
      v = col_raw(col, bpv, i);
      mine = (parts == 1u) || (((hash_u64(v) >> 20) % parts) == part);
    }
    uint32_t f = kPalEmpty;
    if (mine) f = pal_probe(tab, v);
    const bool miss = mine && (f == kPalEmpty);
    const uint64_t mm = __ballot(miss);
    if (miss) tile_vals[threadIdx.x] = v;
    if (lane == 0u) {
      miss_mask[wave] = mm;
      if (mm) flags[0] = 1u;
    }
    __syncthreads();
    const bool any_miss = flags[0] != 0u;
    if (any_miss) {
      if (wave == 0u) {
        uint32_t count = flags[1];
        for (uint32_t w = 0; w < (uint32_t)NW; ++w) {
          const uint64_t m = miss_mask[w];
          if (m == 0u) continue;
          bool act = ((m >> lane) & 1u) != 0u;
          uint64_t val = 0u;
          if (act) {
            val = tile_vals[w * 64u + lane];
            act = pal_probe(tab, val) == kPalEmpty;  // an earlier batch may have inserted it
          }
          // leaders: lowest lane of each group of equal values
          uint64_t todo = __ballot(act);
          bool leader = false;
          while (todo) {
            const int j = __builtin_ctzll(todo);
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)val, j);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(val >> 32), j);
            const uint64_t vj = (((uint64_t)hi) << 32) | lo;
            const uint64_t same = __ballot(act && val == vj);
            if ((int)lane == j) leader = true;
            todo &= ~same;
          }
          const uint64_t leaders = __ballot(leader);
          count += (uint32_t)__builtin_popcountll(leaders);
          if (count > kPalCapacity) break;  // uniform
          if (leader) {
            // distinct new keys: claim the first free slot of the probe sequence, then publish the key
            const uint32_t idx = base + w * 64u + lane;
            uint32_t slot = hash_u64(val) & (kPalSlots - 1u);
            for (;;) {
              // 16-bit CAS emulated on the containing dword
              uint32_t* word = reinterpret_cast<uint32_t*>(tab.first) + (slot >> 1);
              const uint32_t shift = (slot & 1u) * 16u;
              const uint32_t old = *word;
              if (((old >> shift) & 0xffffu) == kPalEmpty) {
                const uint32_t want = (old & ~(0xffffu << shift)) | (idx << shift);
                if (atomicCAS(word, old, want) == old) {
                  tab.keys[slot] = val;
                  break;
                }
                continue;  // somebody changed the dword: re-read the same slot
              }
              slot = (slot + 1u) & (kPalSlots - 1u);
            }
          }
        }
        if (lane == 0u) flags[1] = count;
      }
      __syncthreads();
      if (flags[1] > kPalCapacity) return 0xffffffffu;  // uniform
      if (threadIdx.x == 0) flags[0] = 0u;  // every wave has read any_miss; next write is behind the barrier below
      if (miss) f = pal_probe(tab, v);
    }
    if (mine && first_out) first_out[i] = (uint16_t)f;
    __syncthreads();  // flags / miss_mask / tile_vals reuse
  }
  return flags[1];
}

// Distinct-value count of values [0, n) (mode analysis). Falls back to 8 hash partitions when one table
// cannot hold the distinct values.
template <int T>
__device__ uint32_t palette_count(const uint8_t* col, uint32_t bpv, uint32_t n, PalTable tab, uint64_t* tile_vals,
                                  uint64_t* miss_mask, uint32_t* flags) {
  uint32_t u = palette_pass<T>(col, bpv, n, 0u, 1u, tab, tile_vals, miss_mask, flags, nullptr);
  if (u != 0xffffffffu) return u;
  u = 0u;
  for (uint32_t p = 0; p < 8u; ++p) {
    __syncthreads();
    u += palette_pass<T>(col, bpv, n, p, 8u, tab, tile_vals, miss_mask, flags, nullptr);
  }
  return u;
}

// Full palette section of one chunk. Writes segment A ([0x01][u16 U][U values]) at dst and segment B (bit-packed
// indexes) at dst + kPaletteIndexOffset; returns their sizes.
template <int T>
__device__ void section_palette(const uint8_t* col, uint32_t bpv, uint32_t n, uint8_t* dst, uint16_t* first_idx,
                                uint8_t* lds, uint32_t* wtot, uint32_t* size_a, uint32_t* size_b) {
  // LDS carve (bytes): keys 64 KiB | first 16 KiB | tile_vals 8 KiB | miss_mask 128 | flags 16
  PalTable tab;
  tab.keys = reinterpret_cast<uint64_t*>(lds);
  tab.first = reinterpret_cast<uint16_t*>(lds + kPalSlots * 8u);
  uint64_t* tile_vals = reinterpret_cast<uint64_t*>(lds + kPalSlots * 10u);
  uint64_t* miss_mask = reinterpret_cast<uint64_t*>(lds + kPalSlots * 10u + T * 8u);
  uint32_t* flags = reinterpret_cast<uint32_t*>(lds + kPalSlots * 10u + T * 8u + 128u);

  uint32_t u = palette_pass<T>(col, bpv, n, 0u, 1u, tab, tile_vals, miss_mask, flags, first_idx);
  if (u == 0xffffffffu) {
    for (uint32_t p = 0; p < 8u; ++p) {
      __syncthreads();
      (void)palette_pass<T>(col, bpv, n, p, 8u, tab, tile_vals, miss_mask, flags, first_idx);
    }
  }
  __threadfence_block();
  __syncthreads();

  // ranks in first-occurrence order: rank_at[i] for every first-occurrence point i (LDS, reuses the table)
  uint16_t* rank_at = reinterpret_cast<uint16_t*>(lds);  // [32768]
  uint32_t running = 0u;
  for (uint32_t base = 0; base < n; base += T) {
    const uint32_t i = base + threadIdx.x;
    const bool is_first = (i < n) && (first_idx[i] == (uint16_t)i);
    uint32_t total;
    const uint32_t excl = block_exclusive_scan<T>(is_first ? 1u : 0u, wtot, &total);
    if (is_first) {
      const uint32_t r = running + excl;
      rank_at[i] = (uint16_t)r;
      // palette value r, little-endian, behind the 3-byte header
      const uint64_t v = col_raw(col, bpv, i);
      uint8_t* q = dst + 3u + (size_t)r * bpv;
      for (uint32_t k = 0; k < bpv; ++k) q[k] = (uint8_t)(v >> (8u * k));
    }
    running += total;
    __syncthreads();
  }
  const uint32_t U = running;
  if (threadIdx.x == 0) {
    dst[0] = 1u;
    dst[1] = (uint8_t)(U & 0xffu);
    dst[2] = (uint8_t)((U >> 8) & 0xffu);  // u16 truncation as in the reference (v5_codec.cpp:464)
  }
  const uint32_t bits = palette_bits(U);
  *size_a = 3u + U * bpv;
  *size_b = (bits * n + 7u) >> 3;
  if (bits == 0u) return;

  // bit-pack: thread g packs indexes [32g, 32g+32) into `bits` dwords (appendBitpackedIndexes, :209-227)
  uint32_t* idx_out = reinterpret_cast<uint32_t*>(dst + kPaletteIndexOffset);
  for (uint32_t g = threadIdx.x; g * 32u < n; g += T) {
    const uint32_t cnt = min(32u, n - g * 32u);
    uint64_t scratch = 0u;
    uint32_t held = 0u, w = 0u;
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t r = rank_at[first_idx[g * 32u + j]];
      scratch |= ((uint64_t)r) << held;
      held += bits;
      if (held >= 32u) {
        idx_out[g * bits + w] = (uint32_t)scratch;
        ++w;
        scratch >>= 32;
        held -= 32u;
      }
    }
    if (held > 0u) idx_out[g * bits + w] = (uint32_t)scratch;
  }
}

constexpr uint32_t kSecLdsPalette = kPalSlots * 10u + kSecThreads * 8u + 128u + 16u;  // 90256
constexpr uint32_t kSecLdsLists = (kSecThreads + 2u) * 12u + 16u;
constexpr uint32_t kSecLdsMain = (kSecLdsPalette > 65536u + 64u ? kSecLdsPalette : 65536u + 64u);
constexpr uint32_t kSecLdsTotal = ((kSecLdsMain + kSecLdsLists + 15u) & ~15u) + kRingBytes + 128u;

struct SecLds {
  uint8_t* main;       // palette table / rank_at
  uint32_t* list_pos;  // run lists
  uint64_t* list_key;
  uint32_t* ring;
  uint32_t* wtot;
};

__device__ __forceinline__ SecLds sec_lds_carve(uint8_t* smem) {
  SecLds l;
  l.main = smem;
  l.list_key = reinterpret_cast<uint64_t*>(smem + kSecLdsMain);
  l.list_pos = reinterpret_cast<uint32_t*>(smem + kSecLdsMain + (kSecThreads + 2u) * 8u);
  const uint32_t ring_off = (kSecLdsMain + kSecLdsLists + 15u) & ~15u;
  l.ring = reinterpret_cast<uint32_t*>(smem + ring_off);
  l.wtot = reinterpret_cast<uint32_t*>(smem + ring_off + kRingBytes);
  return l;
}

// k_probe_modes: grid = (n_clouds, n_adaptive). analyzeAdaptiveIntField + selectBestAdaptiveIntMode
// (v5_codec.cpp:387-412) over the first min(4096, n) values of the cloud's first chunk (window rule :934-949).
__global__ __launch_bounds__(kSecThreads) void k_probe_modes(const DevPlan plan, const ChunkDesc* __restrict__ chunks,
                                                             const uint32_t* __restrict__ cloud_first_chunk,
                                                             const ColumnPtrs cols, uint8_t* __restrict__ modes) {
  constexpr int T = kSecThreads;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SecLds l = sec_lds_carve(smem);
  const uint32_t cloud = blockIdx.x, a = blockIdx.y;
  if (modes[cloud * plan.n_adaptive + a] != 0xffu) return;  // decided by k_probe_fast
  __syncthreads();
  const uint32_t fc = cloud_first_chunk[cloud];
  if (fc == cloud_first_chunk[cloud + 1u]) {  // empty cloud
    if (threadIdx.x == 0) modes[cloud * plan.n_adaptive + a] = 0u;
    return;
  }
  const ChunkDesc cd = chunks[fc];
  const uint32_t n = cd.n_points > kProbePoints ? kProbePoints : cd.n_points;
  const uint32_t bpv = plan.adaptive[a].bpv, type = plan.adaptive[a].type;
  const uint8_t* col = cols.p[a] + (size_t)cd.first_point * bpv;

  const uint32_t delta = section_delta_varint<T, false>(col, bpv, type, n, l.ring, l.wtot, nullptr);
  __syncthreads();
  const uint32_t rle = section_runs<T, false, false>(col, bpv, type, n, l.ring, l.wtot, nullptr, l.list_pos, l.list_key);
  __syncthreads();
  const uint32_t drle = section_runs<T, false, true>(col, bpv, type, n, l.ring, l.wtot, nullptr, l.list_pos, l.list_key);
  __syncthreads();
  PalTable tab;
  tab.keys = reinterpret_cast<uint64_t*>(l.main);
  tab.first = reinterpret_cast<uint16_t*>(l.main + kPalSlots * 8u);
  uint64_t* tile_vals = reinterpret_cast<uint64_t*>(l.main + kPalSlots * 10u);
  uint64_t* miss_mask = reinterpret_cast<uint64_t*>(l.main + kPalSlots * 10u + T * 8u);
  uint32_t* flags = reinterpret_cast<uint32_t*>(l.main + kPalSlots * 10u + T * 8u + 128u);
  const uint32_t U = palette_count<T>(col, bpv, n, tab, tile_vals, miss_mask, flags);
  const uint32_t pal = 3u + U * bpv + ((palette_bits(U) * n + 7u) >> 3);

  uint32_t mode = 0u, best = delta;  // strict '<' in this order (selectBestAdaptiveIntMode)
  if (pal < best) { best = pal; mode = 1u; }
  if (rle < best) { best = rle; mode = 2u; }
  if (drle < best) { mode = 3u; }
  if (threadIdx.x == 0) modes[cloud * plan.n_adaptive + a] = (uint8_t)mode;
}

// k_encode_sections: the safety net behind the fast section kernels. grid = min(n_chunks * n_adaptive, kSecGrid) workgroups
// that take the (chunk, field) pairs round robin (round 6: the kernel's 119 KB of LDS allow one workgroup per CU anyway, and
// with the fused Palette nearly every pair is a check and nothing else: a thousand 1024-thread workgroups that return at
// once cost 8 us of dispatch on the 32-cloud batch)
constexpr uint32_t kSecGrid = 256;
__global__ __launch_bounds__(kSecThreads) void k_encode_sections(const DevPlan plan, const ChunkDesc* __restrict__ chunks,
                                                                 const ColumnPtrs cols, const uint8_t* __restrict__ modes,
                                                                 uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                                 uint64_t reg_stride, Seg* __restrict__ segs,
                                                                 uint32_t segs_per_chunk, const ColumnPtrs rank_cols,
                                                                 uint32_t subs, const uint8_t* __restrict__ handled_flags,
                                                                 uint32_t fused_field, uint32_t n_chunks) {
  constexpr int T = kSecThreads;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const SecLds l = sec_lds_carve(smem);
  const uint32_t n_pairs = n_chunks * plan.n_adaptive;
  for (uint32_t w = blockIdx.x; w < n_pairs; w += gridDim.x) {  // (every test below is uniform over the workgroup)
  const uint32_t a = w / n_chunks, c = w - a * n_chunks;
  if (handled_flags[(size_t)c * plan.n_adaptive + a]) continue;  // a fast-path kernel already wrote this section
  const ChunkDesc cd = chunks[c];
  // k_finish builds the Palette sections of this field itself (any number of distinct values)
  if (a == fused_field && modes[cd.cloud * plan.n_adaptive + a] == 1u) continue;
  __syncthreads();  // (the pair this workgroup did before is through with the LDS)
  const uint32_t n = cd.n_points;
  const uint32_t bpv = plan.adaptive[a].bpv, type = plan.adaptive[a].type;
  const uint8_t* col = cols.p[a] + (size_t)cd.first_point * bpv;
  const uint32_t sec_off = (uint32_t)reg_stride + a * kSectionStride;
  uint8_t* dst = slots + (size_t)c * slot_stride + sec_off;
  const uint32_t mode = modes[cd.cloud * plan.n_adaptive + a];

  for (uint32_t i = threadIdx.x; i < kRingU4; i += T) reinterpret_cast<uint4*>(l.ring)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  uint32_t size_a = 0u, size_b = 0u;
  if (mode == 0u) {
    size_a = section_delta_varint<T, true>(col, bpv, type, n, l.ring, l.wtot, dst);
  } else if (mode == 2u) {
    size_a = section_runs<T, true, false>(col, bpv, type, n, l.ring, l.wtot, dst, l.list_pos, l.list_key);
  } else if (mode == 3u) {
    size_a = section_runs<T, true, true>(col, bpv, type, n, l.ring, l.wtot, dst, l.list_pos, l.list_key);
  } else {
    uint16_t* first_idx = reinterpret_cast<uint16_t*>(rank_cols.p[a]) + cd.first_point;
    section_palette<T>(col, bpv, n, dst, first_idx, l.main, l.wtot, &size_a, &size_b);
  }
  if (threadIdx.x == 0) {
    Seg s;
    s.off = sec_off;
    s.size = size_a;
    segs[(size_t)c * segs_per_chunk + subs + 2u * a] = s;
    s.off = sec_off + kPaletteIndexOffset;
    s.size = size_b;
    segs[(size_t)c * segs_per_chunk + subs + 1u + 2u * a] = s;
  }
  }
}

}  // namespace cldn

#include "stage1_wide.h"

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
#include "cloudini_hip.h"
#include "stage1_launch.h"

namespace cldn {

namespace {
constexpr int kRegularThreads = 1024;
inline uint32_t regular_lds(uint32_t step) { return 2u * regular_tile_lds(kRegularThreads, step) + kRingBytes + 128u; }
constexpr uint32_t kRegularLdsMax = 2u * (((64u * kMaxPointStep + kMaxPointStep + 48u) + 15u) & ~15u) + kRingBytes + 128u;
constexpr uint32_t kFloatnRing = 16384;  // >= one tile of 3-lane points at 5 bytes per token (1008 * 15 B); wider tiles use windows
constexpr uint32_t kFloatnLds = kFloatnRing + 256u + kStagedCols * (4u * 63u * 2u) * 4u + 64u;  // ring, wtot, staged columns (TILE = 504)

// FloatN fast path: the regular stream is exactly one fused 3/4-lane float encoder on a 4-byte aligned layout
// lanes of the fused FloatN encoder the fast kernel can take (0 = none); *l3 = dword of the fourth lane
// `tail` (may be NULL): index of ONE more regular op behind the lanes that the piece kernel can append to every point
// (raw copy, scalar lossy float, Gorilla token), -1 if there is none. Callers that pass NULL get 0 for such plans.
int floatn_lanes(const DevPlan& p, const uint8_t* points, int* l3, int* tail = nullptr) {
  *l3 = 3;
  if (tail) *tail = -1;
  uint32_t lanes = 0;
  while (lanes < p.n_ops && lanes < 4u && p.ops[lanes].kind == OP_QF32) ++lanes;
  if (lanes != 3u && lanes != 4u) return 0;
  if (p.n_ops != lanes) {
    if (!tail || p.n_ops != lanes + 1u) return 0;
    const uint32_t k = p.ops[lanes].kind;
    // (XOR fields only exist in lossless schemas, which have no FloatN lanes)
    if (k != OP_COPY && k != OP_LOSSY_F32 && k != OP_LOSSY_F64 && k != OP_GORILLA64) return 0;
    if (p.ops[lanes].size > 8u || p.ops[lanes].offset < p.ops[0].offset) return 0;
    *tail = (int)lanes;
  }
  for (uint32_t k = 0; k < lanes; ++k) {
    if (k < 3u && p.ops[k].offset != p.ops[0].offset + 4u * k) return 0;
  }
  if (lanes == 4u) {
    if (p.ops[3].offset == p.ops[0].offset + 16u) *l3 = 4;
    else if (p.ops[3].offset != p.ops[0].offset + 12u) return 0;
  }
  return (int)lanes;
}

bool floatn_unaligned(const DevPlan& p, const uint8_t* points) {
  return (p.point_step & 3u) || (p.ops[0].offset & 3u) || ((uintptr_t)points & 3u);
}

// dwords to load per point so that every adaptive-int field (and the tail op's field) is covered by the point load
// (0 = not possible)
int floatn_loadw(const DevPlan& p, int lanes, bool unal, int l3, int tail = -1) {
  const uint32_t off0 = p.ops[0].offset;
  // bytes behind off0 the tail needs; an unaligned 8-byte field is read from three dwords
  uint32_t tail_need = 0;
  if (tail >= 0) {
    const uint32_t rel = p.ops[tail].offset - off0;
    tail_need = ((rel >> 2) + (p.ops[tail].size > 4u || (rel & 3u) + p.ops[tail].size > 4u ? ((rel & 3u) ? 3u : 2u) : 1u)) * 4u;
  }
  if (l3 == 4) return (!unal && off0 + 32u <= p.point_step && tail_need <= 32u) ? 8 : 0;  // one variant: aligned, 8 dwords
  if (p.n_adaptive == 0 && tail < 0) return unal && lanes == 3 ? 4 : lanes;
  uint32_t need = std::max((uint32_t)lanes * 4u, tail_need);
  for (uint32_t a = 0; a < p.n_adaptive; ++a) {
    const DevAdaptive& f = p.adaptive[a];
    // fields the point load cannot deliver: the aligned kernels then read every field directly (loadw == lanes);
    // the unaligned instantiations have no such mode -> 0 = generic kernel
    if (f.offset < off0) return (unal || tail >= 0) ? 0 : lanes;
    if (f.bpv == 8u && ((f.offset - off0) & 3u)) return (unal || tail >= 0) ? 0 : lanes;
    need = std::max(need, f.offset - off0 + f.bpv);
  }
  const int w = (int)((need + 3u) / 4u);
  if (tail >= 0) {  // TAIL instantiations: (3: 4, 8), (4: 8), aligned and unaligned
    if (w > 8) return 0;
    return (lanes == 3 && w <= 4) ? 4 : 8;  // (an aligned layout whose point is shorter takes the guarded UNAL variant)
  }
  if (unal) {  // realigned dword loads may reach into the next point; variants: (3: 4, 8), (4: 5, 8)
    if (w > 8) return 0;
    if (lanes == 3) return w <= 4 ? 4 : 8;
    return w <= 5 ? 5 : 8;
  }
  const int loadw = w <= lanes ? lanes : (w <= 4 ? 4 : (w <= 8 ? 8 : 0));
  if (loadw == 0 || off0 + (uint32_t)loadw * 4u > p.point_step) return lanes;  // would read past the point
  return loadw;
}

int hip_fail(hipError_t e, const char* what) { return launch_fail(e, what); }
}  // namespace

int stage1_configure_kernels() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_regular<kRegularThreads, false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)regular_lds(kWidePointStep));
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_encode_regular)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_regular<kRegularThreads, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRegularLdsMax);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_encode_regular wide)");
  const void* fk[] = {reinterpret_cast<const void*>(&k_encode_floatn<256, 3, 2, kFloatnRing, 3, false>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 3, 2, kFloatnRing, 4, false>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 3, 2, kFloatnRing, 8, false>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 4, 2, kFloatnRing, 4, false>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 4, 2, kFloatnRing, 8, false>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 3, 2, kFloatnRing, 4, false, true>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 3, 2, kFloatnRing, 8, false, true>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 4, 2, kFloatnRing, 5, false, true>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 4, 2, kFloatnRing, 8, false, true>),
                      reinterpret_cast<const void*>(&k_encode_floatn<256, 4, 2, kFloatnRing, 8, false, false, 4>)};
  for (const void* f : fk) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFloatnLds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_encode_floatn)");
  }
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe_modes), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kSecLdsTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_probe_modes)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_encode_sections),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSecLdsTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_encode_sections)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe_fast), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kProbeLds);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_probe_fast)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide_probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kProbeLds);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_wide_probe)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide_encode), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSecLdsTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_wide_encode)");
  const void* pk[] = {reinterpret_cast<const void*>(&k_section_palette<uint16_t>),
                      reinterpret_cast<const void*>(&k_section_palette<uint32_t>),
                      reinterpret_cast<const void*>(&k_section_palette<uint64_t>)};
  for (const void* f : pk) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS2PalLds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_section_palette)");
  }
  if (int rc = stage1_configure_decode()) return rc;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_section_fast), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kD32Lds);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_section_fast)");
  const void* pk32[] = {reinterpret_cast<const void*>(&k_section_palette32<uint16_t, kS2Threads>),
                        reinterpret_cast<const void*>(&k_section_palette32<uint16_t, 512>),
                        reinterpret_cast<const void*>(&k_section_palette32<uint32_t, 512>)};
  for (const void* f : pk32) {
    e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Pal32<uint32_t>::kLds);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_section_palette32)");
  }
  return CLDN_HIP_OK;
}

// ---- single-pass encoder ----
namespace {
struct FusedVariant {
  int lanes, loadw, l3;
  bool unal;
  int tail;  // index of the op appended behind the lanes, -1 = none
};
// the k_encode_floatn variant table decides whether the point load covers the plan
bool fused_variant(const DevPlan& p, const uint8_t* points, FusedVariant* v) {
  int l3 = 3, tail = -1;
  const int lanes = floatn_lanes(p, points, &l3, &tail);
  if (!lanes) return false;
  bool unal = floatn_unaligned(p, points);
  const int loadw = floatn_loadw(p, lanes, unal, l3, tail);
  if (loadw == 0) return false;
  // TAIL on an aligned layout whose loaded dwords reach into the next point: the UNAL instantiation (aligned dwords +
  // realignment, here by 0 bytes) has the guard for the last points of the batch
  if (tail >= 0 && !unal && l3 != 4 && p.ops[0].offset + (uint32_t)loadw * 4u > p.point_step) unal = true;
  bool ok;
  if (tail >= 0) ok = l3 == 4 ? loadw == 8 : ((lanes == 3 && (loadw == 4 || loadw == 8)) || (lanes == 4 && loadw == 8));
  else if (l3 == 4) ok = (loadw == 8);
  else if (unal) ok = (lanes == 3 && (loadw == 4 || loadw == 8)) || (lanes == 4 && (loadw == 5 || loadw == 8));
  else ok = (lanes == 3 && (loadw == 3 || loadw == 4 || loadw == 8)) || (lanes == 4 && (loadw == 4 || loadw == 8));
  if (!ok) return false;
  v->lanes = lanes;
  v->loadw = loadw;
  v->l3 = l3;
  v->unal = unal;
  v->tail = tail;
  return true;
}
}  // namespace

uint32_t stage1_piece_points(const DevPlan& plan, const uint8_t* points) {
  FusedVariant v;
  if (!fused_variant(plan, points, &v)) return 0u;
  return fused_piece_points(v.lanes);
}

// the piece kernel takes this plan and its tail op is the Gorilla field: points per piece, else 0
uint32_t stage1_gorilla_inline_piece_points(const DevPlan& plan, const uint8_t* points) {
  FusedVariant v;
  if (!fused_variant(plan, points, &v) || v.tail < 0 || plan.ops[v.tail].kind != OP_GORILLA64) return 0u;
  return fused_piece_points(v.lanes);
}

uint32_t stage1_piece_slot_stride(const DevPlan& plan, const uint8_t* points) {
  FusedVariant v;
  if (!fused_variant(plan, points, &v)) return 0u;
  const uint32_t per_point = 5u * (uint32_t)v.lanes + (v.tail >= 0 ? kTailMaxBytes : 0u);  // worst case, 5 bytes per token
  return (fused_piece_points(v.lanes) * per_point + 255u) & ~255u;
}

// instantiations without a TAIL op that prefetch at most 5 dwords per point: the 64-VGPR kernel (8 waves per SIMD)
template <int LL, int WW, bool UU, int L3>
static void launch_fused_variant(dim3 grid, dim3 block, uint32_t lds, hipStream_t stream, const DevPlan& plan, const FusedArgs& A) {
  if constexpr (WW <= 5)
    hipLaunchKernelGGL((k_encode_fused_w8<LL, WW, UU, L3>), grid, block, lds, stream, plan, A);
  else
    hipLaunchKernelGGL((k_encode_fused<LL, WW, UU, L3>), grid, block, lds, stream, plan, A);
}

static int launch_fused(const EncodeLaunch& L, hipStream_t stream, uint32_t piece0, uint32_t piece1, bool* probed) {
  FusedVariant v;
  if (!fused_variant(*L.plan, L.points, &v)) return launch_fail(hipErrorInvalidValue, "k_encode_fused (no variant)");
  FusedArgs A;
  A.points = L.points;
  A.points_end = L.points_end;
  A.pieces = L.pieces + piece0;
  A.cols = L.cols;
  static const uint32_t ablate_f = (uint32_t)dev_env_int("CLDN_HIP_ABLATE", 0);  // profiling only
  A.ablate = ablate_f;
  A.slots = L.slots;
  A.slot_stride = L.slot_stride;
  A.piece_stride = L.sub_stride / kFusedWaves;  // sub_stride = one workgroup's range (4 pieces)
  A.segs = L.segs;
  A.segs_per_chunk = L.segs_per_chunk;
  // mode probe next to the pieces: fields of 2 and 4 bytes (the distinct-value structure has to fit the launch's LDS)
  static const bool probe_in_piece_env = dev_env_int("CLDN_HIP_PROBE_IN_PIECE", 1) != 0;  // A/B switch
  bool probe_here = probe_in_piece_env && piece0 == 0u && L.plan->n_adaptive != 0u && !L.modes_forced && L.n_clouds != 0u &&
                    (uint64_t)L.n_clouds * L.plan->n_adaptive < (1u << 20);
  for (uint32_t a = 0; a < L.plan->n_adaptive && probe_here; ++a) probe_here = L.plan->adaptive[a].bpv <= 4u;
  A.n_probe = probe_here ? L.n_clouds * L.plan->n_adaptive : 0u;
  A.chunks = L.chunks;
  A.cloud_first_chunk = L.cloud_first_chunk;
  A.modes = L.modes;
  A.intra = L.intra ? 1u : 0u;
  A.epoch = L.fin_epoch;
  A.wgrec = L.wgrec;
  A.status = L.status;
  A.tail_kind = 0u;
  A.tail_rel = 0u;
  A.tail_size = 0u;
  A.tail_windows = nullptr;
  if (v.tail >= 0) {
    const DevOp& top = L.plan->ops[v.tail];
    A.tail_kind = top.kind;
    A.tail_rel = top.offset - L.plan->ops[0].offset;
    A.tail_size = top.size;
    if (top.kind == OP_GORILLA64) A.tail_windows = reinterpret_cast<const uint16_t*>(L.pre.p[top.type]);
  }
  const uint32_t region = v.tail >= 0 ? fused_region_bytes_tail(v.lanes)
                                      : fused_region_bytes_small(v.lanes);
  uint32_t lds = 16u + kFusedWaves * region;
  // the probe workgroups of the launch share its LDS size: a 16-bit field needs its 8 KiB value bitmap, a 32-bit field a
  // hash table of 6144 slots (24.8 KB: above the 18.3 KB of four 3-byte-per-token regions -- such launches keep 6
  // workgroups per CU instead of 8)
  if (A.n_probe) {
    bool wide = false;
    for (uint32_t a = 0; a < L.plan->n_adaptive; ++a) wide = wide || L.plan->adaptive[a].bpv == 4u;
    if (wide) lds = std::max(lds, 6144u * 4u + 272u);
  }
  A.probe_lds = lds;
  if (probed) *probed = A.n_probe != 0u;
  const dim3 grid(A.n_probe + (piece1 - piece0) / kFusedWaves), block(kFusedThreads);
#define LAUNCH_FUSED(LL, WW, UU, L3) launch_fused_variant<LL, WW, UU, L3>(grid, block, lds, stream, *L.plan, A)
#define LAUNCH_FUSED_TAIL(LL, WW, UU, L3)                                                                        \
  hipLaunchKernelGGL((k_encode_fused<LL, WW, UU, L3, true>), grid, block, lds, stream, *L.plan, A)
  if (v.tail >= 0) {
    if (v.l3 == 4) LAUNCH_FUSED_TAIL(4, 8, false, 4);
    else if (v.unal && v.lanes == 3 && v.loadw == 4) LAUNCH_FUSED_TAIL(3, 4, true, 3);
    else if (v.unal && v.lanes == 3) LAUNCH_FUSED_TAIL(3, 8, true, 3);
    else if (v.unal) LAUNCH_FUSED_TAIL(4, 8, true, 3);
    else if (v.lanes == 3 && v.loadw == 4) LAUNCH_FUSED_TAIL(3, 4, false, 3);
    else if (v.lanes == 3) LAUNCH_FUSED_TAIL(3, 8, false, 3);
    else LAUNCH_FUSED_TAIL(4, 8, false, 3);
  } else if (v.l3 == 4) LAUNCH_FUSED(4, 8, false, 4);
  else if (v.unal && v.lanes == 3 && v.loadw == 4) LAUNCH_FUSED(3, 4, true, 3);
  else if (v.unal && v.lanes == 3 && v.loadw == 8) LAUNCH_FUSED(3, 8, true, 3);
  else if (v.unal && v.lanes == 4 && v.loadw == 5) LAUNCH_FUSED(4, 5, true, 3);
  else if (v.unal && v.lanes == 4 && v.loadw == 8) LAUNCH_FUSED(4, 8, true, 3);
  else if (v.lanes == 3 && v.loadw == 3) LAUNCH_FUSED(3, 3, false, 3);
  else if (v.lanes == 3 && v.loadw == 4) LAUNCH_FUSED(3, 4, false, 3);
  else if (v.lanes == 3 && v.loadw == 8) LAUNCH_FUSED(3, 8, false, 3);
  else if (v.lanes == 4 && v.loadw == 4) LAUNCH_FUSED(4, 4, false, 3);
  else LAUNCH_FUSED(4, 8, false, 3);
#undef LAUNCH_FUSED
#undef LAUNCH_FUSED_TAIL
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "k_encode_fused");
  return CLDN_HIP_OK;
}

// section kernels of chunks [c0, c1) on `stream` (per-chunk tables are passed shifted to c0; columns, modes and the
// chunk descriptors' point indexes are batch-global)
static int launch_sections(const EncodeLaunch& L, hipStream_t stream, uint32_t c0, uint32_t c1, uint32_t fused_field) {
  hipError_t e;
  const uint32_t na = L.plan->n_adaptive;
  const uint32_t nch = c1 - c0;
  if (!na || !nch) return CLDN_HIP_OK;
  const ChunkDesc* chunks = L.chunks + c0;
  uint8_t* slots = L.slots + (size_t)c0 * L.slot_stride;
  Seg* segs = L.segs + (size_t)c0 * L.segs_per_chunk;
  uint8_t* flags = L.fallback_flags + (size_t)c0 * na;
  ColumnPtrs rank_cols;
  for (int a = 0; a < kMaxAdaptive; ++a) rank_cols.p[a] = reinterpret_cast<uint8_t*>(L.ranks[a]);
  static const bool no_fast = dev_env("CLDN_HIP_NO_FAST_SECTIONS") != nullptr;  // A/B switch: general kernels only
  // One launch per kernel type covers all the fields of that type (grid.y): the fields are independent and every one of
  // these kernels is latency-bound at one workgroup per chunk, so a schema with five integer channels gets five times
  // the workgroups in flight instead of five launches in a row.
  SectionFields run16, run32, pal16, pal32, pal64;
  run16.n = run32.n = pal16.n = pal32.n = pal64.n = 0u;
  for (uint32_t a = 0; a < na && !no_fast; ++a) {
    const uint32_t bpv = L.plan->adaptive[a].bpv;
    const uint32_t hint = L.mode_hint[a];
    if (hint & 0xDu) {  // DeltaVarint / Rle / DeltaRle expected somewhere
      if (bpv == 2u) run16.a[run16.n++] = (uint8_t)a;
      else if (bpv == 4u) run32.a[run32.n++] = (uint8_t)a;
    }
    if ((hint & 0x2u) && a != fused_field) {
      if (bpv == 2u) pal16.a[pal16.n++] = (uint8_t)a;
      else if (bpv == 4u) pal32.a[pal32.n++] = (uint8_t)a;
      else pal64.a[pal64.n++] = (uint8_t)a;
    }
  }
  // one adaptive field + one regular segment per chunk (intra-chunk placement): the section goes right behind the regular
  // stream, so that the chunk's payload is one contiguous run of its slot
  const uint32_t append = (L.intra && na == 1u && L.subs == 1u) ? 1u : 0u;
#define SEC_ARGS(FL) *L.plan, FL, chunks, L.cols, L.modes, slots, L.slot_stride, L.reg_stride, segs, L.segs_per_chunk, L.subs, flags, append
  {
    SectionFields runs;  // every 2- and 4-byte field that may be DeltaVarint / Rle / DeltaRle somewhere: one launch
    runs.n = 0u;
    for (uint32_t k = 0; k < run16.n; ++k) runs.a[runs.n++] = run16.a[k];
    for (uint32_t k = 0; k < run32.n; ++k) runs.a[runs.n++] = run32.a[k];
    if (runs.n) hipLaunchKernelGGL(k_section_fast, dim3(nch, runs.n), dim3(kS2Threads), kD32Lds, stream, SEC_ARGS(runs));
  }
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_section_fast");
  if (pal16.n)
  {
    // 512-thread workgroups (two bitmap words and two groups of 32 values per thread): four of them fit a CU, so a batch
    // of up to 1024 chunks is one generation (C2: sections 0.066 -> 0.062 ms); CLDN_HIP_PAL32_THREADS=1024 is the A/B switch
    static const bool pal1024 = dev_env_int("CLDN_HIP_PAL32_THREADS", 0) == 1024;
    if (pal1024)
      hipLaunchKernelGGL((k_section_palette32<uint16_t, kS2Threads>), dim3(nch, pal16.n), dim3(kS2Threads), Pal32<uint16_t>::kLds, stream, SEC_ARGS(pal16), rank_cols, L.status);
    else
      hipLaunchKernelGGL((k_section_palette32<uint16_t, 512>), dim3(nch, pal16.n), dim3(512), Pal32<uint16_t>::kLds, stream, SEC_ARGS(pal16), rank_cols, L.status);
  }
  if (pal32.n)
    hipLaunchKernelGGL((k_section_palette32<uint32_t, 512>), dim3(nch, pal32.n), dim3(512), Pal32<uint32_t>::kLds, stream, SEC_ARGS(pal32), rank_cols, L.status);
  if (pal64.n)
    hipLaunchKernelGGL(k_section_palette<uint64_t>, dim3(nch, pal64.n), dim3(kS2Threads), kS2PalLds, stream, SEC_ARGS(pal64));
#undef SEC_ARGS
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_section_palette");
  hipLaunchKernelGGL(k_encode_sections, dim3(std::min<uint32_t>(nch * na, kSecGrid)), dim3(kSecThreads), kSecLdsTotal, stream, *L.plan, chunks, L.cols,
                     L.modes, slots, L.slot_stride, L.reg_stride, segs, L.segs_per_chunk, rank_cols, L.subs, flags, fused_field, nch);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_encode_sections");
  return CLDN_HIP_OK;
}

constexpr uint32_t kNoFusedField = 0xffffffffu;

// profiling only: phase stamps of k_finish's leaders (CLDN_HIP_FINISH_TRACE=1), read back by tools/fintrace.py through
// cldn_hip_debug_finish_trace (not part of include/cloudini_hip.h)
static unsigned long long* g_fin_trace = nullptr;
static uint32_t g_fin_trace_chunks = 0u;
static unsigned long long* stage1_finish_trace_buffer(uint32_t n_chunks) {
  if (n_chunks > g_fin_trace_chunks) {
    if (g_fin_trace) (void)hipFree(g_fin_trace);
    g_fin_trace = nullptr;
    if (hipMalloc(&g_fin_trace, (size_t)n_chunks * 16u * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    g_fin_trace_chunks = n_chunks;
  }
  return g_fin_trace;
}
extern "C" __attribute__((visibility("default"))) int cldn_hip_debug_finish_trace(unsigned long long* host_out, uint32_t n_chunks) {
  if (!g_fin_trace || n_chunks > g_fin_trace_chunks) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpy(host_out, g_fin_trace, (size_t)n_chunks * 16u * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}

// bytes per point of a regular stream made of fixed-size encoders only (XOR-coded floats, raw copies), 0 otherwise
static uint32_t fixed_point_bytes(const DevPlan& P) {
  static const bool off = dev_env("CLDN_HIP_NO_FIXED_ENCODE") != nullptr;  // A/B switch
  if (off || P.n_ops == 0u || P.n_gorilla != 0u) return 0u;
  uint32_t bytes = 0u;
  for (uint32_t k = 0; k < P.n_ops; ++k) {
    const uint32_t kd = P.ops[k].kind;
    if (kd != OP_COPY && kd != OP_XOR32 && kd != OP_XOR64) return 0u;
    bytes += P.ops[k].size;
  }
  return bytes;
}

size_t stage1_wide_scratch_bytes() { return kWideScratchBytes; }

// WIDE route (stage1_wide.h): Gorilla pre-pass in groups, mode probe, one workgroup per chunk, framing
static int launch_encode_wide(const EncodeLaunch& L) {
  hipError_t e;
  const WidePlan& W = *L.wide;
  if (L.events) {
    (void)hipEventRecord(L.events[0], L.stream);
    (void)hipEventRecord(L.events[1], L.stream);
  }
  if (L.n_chunks) {
    if (W.n_gorilla) {
      // k_gorilla_tokens finds "the blockIdx.y-th Gorilla op of the plan": a plan of at most kMaxOps such ops per launch
      DevPlan mini;
      mini = DevPlan{};
      mini.point_step = W.point_step;
      uint32_t g0 = 0u;
      for (uint32_t k = 0; k <= W.n_ops; ++k) {
        if (k < W.n_ops && L.wide_ops_host[k].kind == OP_GORILLA64) mini.ops[mini.n_ops++] = L.wide_ops_host[k];
        if (mini.n_ops == (uint32_t)kMaxOps || (k == W.n_ops && mini.n_ops != 0u)) {
          mini.n_gorilla = mini.n_ops;
          hipLaunchKernelGGL(k_gorilla_tokens, dim3(L.n_chunks, mini.n_ops), dim3(kGorThreads), 0, L.stream, mini, L.points, L.points_end,
                             L.chunks, L.pre_out + g0);
          if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_gorilla_tokens (wide)");
          g0 += mini.n_ops;
          mini.n_ops = 0u;
        }
      }
    }
    WideEncodeArgs A;
    A.plan = W;
    A.points = L.points;
    A.points_end = L.points_end;
    A.chunks = L.chunks;
    A.cloud_first_chunk = L.cloud_first_chunk;
    A.modes = L.modes;
    A.slots = L.slots;
    A.slot_stride = L.slot_stride;
    A.segs = L.segs;
    A.scratch = L.wide_scratch;
    A.pre = L.wide_pre;
    if (W.n_adaptive && !L.modes_forced) {
      hipLaunchKernelGGL(k_wide_probe, dim3(L.n_clouds * W.n_adaptive), dim3(kS2Threads), kProbeLds, L.stream, A, L.n_clouds);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_wide_probe");
    }
    hipLaunchKernelGGL(k_wide_encode, dim3(L.n_chunks), dim3(kWideThreads), kSecLdsTotal, L.stream, A);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_wide_encode");
  }
  if (L.events) {
    (void)hipEventRecord(L.events[2], L.stream);
    (void)hipEventRecord(L.events[3], L.stream);
  }
  if (L.chunks_only) {
    if (L.n_chunks) {
      hipLaunchKernelGGL(k_chunk_sizes, dim3((L.n_chunks + 255u) / 256u), dim3(256), 0, L.stream, L.segs, 1u, L.n_chunks, L.chunk_payload,
                         L.contiguous_flag);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_chunk_sizes (wide)");
    }
  } else {
    FrameLaunch F;
    F.stream = L.stream;
    F.chunks = L.chunks;
    F.n_chunks = L.n_chunks;
    F.cloud_first_chunk = L.cloud_first_chunk;
    F.n_clouds = L.n_clouds;
    F.slots = L.slots;
    F.slot_stride = L.slot_stride;
    F.segs = L.segs;
    F.segs_per_chunk = 1u;
    F.rec = L.fin_rec;
    F.anchor = L.fin_anchor;
    F.epoch = L.fin_epoch;
    F.ticket = L.fin_ticket;
    F.use_ticket = L.use_ticket;
    F.test_timeout = L.test_timeout;
    F.chunk_payload = L.chunk_payload;
    F.chunk_dst = L.chunk_dst;
    F.stream_offsets = L.stream_offsets;
    F.out = L.out;
    F.out_capacity = L.out_capacity;
    F.status = L.status;
    const int rc = stage1_launch_frame(F);
    if (rc != CLDN_HIP_OK) return rc;
  }
  if (L.events) (void)hipEventRecord(L.events[4], L.stream);
  return CLDN_HIP_OK;
}

int stage1_launch_encode(const EncodeLaunch& L) {
  hipError_t e;
  if (L.wide) return launch_encode_wide(L);
  // CLDN_HIP_FINISH (A/B switch): 0 = the round-2 kernels (k_chunk_offsets + k_compact), 1 = k_finish without the fused
  // Palette section, 2 (default) = k_finish with it where the schema allows
  static const int finish_mode = dev_env_int("CLDN_HIP_FINISH", 2);
  // the field whose Palette sections k_finish builds itself: the first 2- or 4-byte adaptive field that may commit Palette
  uint32_t fused_field = kNoFusedField;
  if (finish_mode >= 2 && L.n_chunks && !L.chunks_only) {
    for (uint32_t a = 0; a < L.plan->n_adaptive && fused_field == kNoFusedField; ++a)
      if ((L.plan->adaptive[a].bpv == 2u || L.plan->adaptive[a].bpv == 4u) && (L.mode_hint[a] & 0x2u)) fused_field = a;
  }
  if (L.events) (void)hipEventRecord(L.events[0], L.stream);
  uint32_t gor_piece_pts = 0u;  // the piece kernel encodes the Gorilla field itself: points per piece
  bool modes_probed = false;    // the piece kernel's launch decided the adaptive-int modes
  if (L.events) (void)hipEventRecord(L.events[1], L.stream);
  gor_piece_pts = (L.n_chunks && L.pieces) ? stage1_gorilla_inline_piece_points(*L.plan, L.points) : 0u;
  if (gor_piece_pts) {
    const uint32_t opi = L.plan->n_ops - 1u;  // the tail op
    hipLaunchKernelGGL(k_gorilla_windows, dim3(L.n_chunks), dim3(kGorThreads), 0, L.stream, *L.plan, opi, gor_piece_pts,
                       L.points, L.points_end, L.chunks,
                       reinterpret_cast<uint16_t*>(const_cast<uint4*>(L.pre.p[L.plan->ops[opi].type])));
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_gorilla_windows");
  } else if (L.n_chunks && L.plan->n_gorilla) {
    hipLaunchKernelGGL(k_gorilla_tokens, dim3(L.n_chunks, L.plan->n_gorilla), dim3(kGorThreads), 0, L.stream, *L.plan,
                       L.points, L.points_end, L.chunks, L.pre_out);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_gorilla_tokens");
  }
  if (L.n_chunks && L.pieces) {  // slot pipeline, regular stream by the barrier-free piece kernel
    const int rc = launch_fused(L, L.stream, 0u, L.n_pieces, &modes_probed);
    if (rc != CLDN_HIP_OK) return rc;
  } else if (L.n_chunks && fixed_point_bytes(*L.plan) != 0u) {
    // every per-point encoder writes a fixed number of bytes (lossless floats, raw copies): one thread per point, which also
    // splits the integer fields off into their columns. CLDN_HIP_NO_FIXED_ENCODE=1: A/B switch (handled in fixed_point_bytes)
    const uint32_t pb = fixed_point_bytes(*L.plan);
    const uint64_t total_points = (uint64_t)(L.points_end - L.points) / L.plan->point_step;
    const uint64_t total = 4ull * L.n_chunks + (uint64_t)pb * total_points;
    // without integer columns every size is known here: the kernel writes the framed streams themselves (an output that is too
    // small takes the slot path, whose k_finish reports it). CLDN_HIP_NO_FIXED_DIRECT=1: A/B switch
    static const bool no_direct = dev_env("CLDN_HIP_NO_FIXED_DIRECT") != nullptr;
    const bool direct = !no_direct && !L.chunks_only && finish_mode != 0 && L.plan->n_adaptive == 0u && total <= L.out_capacity;
    hipLaunchKernelGGL(k_encode_fixed, dim3(L.n_chunks, kPointsPerChunk / 256u), dim3(256), 0, L.stream, *L.plan, L.points, L.chunks,
                       L.slots, L.slot_stride, L.segs, L.segs_per_chunk, L.subs, L.sub_points, L.sub_stride, pb, L.cols,
                       direct ? L.out : (uint8_t*)nullptr, L.chunk_payload, L.chunk_dst);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_encode_fixed");
    if (direct) {
      hipLaunchKernelGGL(k_fixed_offsets, dim3((L.n_clouds + 256u) / 256u), dim3(256), 0, L.stream, L.chunks, L.cloud_first_chunk, L.n_clouds,
                         L.n_chunks, pb, total, L.stream_offsets);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_fixed_offsets");
      if (L.events) {
        (void)hipEventRecord(L.events[2], L.stream);
        (void)hipEventRecord(L.events[3], L.stream);
        (void)hipEventRecord(L.events[4], L.stream);
      }
      return CLDN_HIP_OK;
    }
  } else if (L.n_chunks) {
    int l3 = 3;
    const int lanes = floatn_lanes(*L.plan, L.points, &l3);
    static const uint32_t ablate = (uint32_t)dev_env_int("CLDN_HIP_ABLATE", 0);  // profiling only
    const bool unal = lanes && floatn_unaligned(*L.plan, L.points);
    const int loadw = lanes ? floatn_loadw(*L.plan, lanes, unal, l3) : 0;
    // LDS: ring, scan scratch and one staging area per adaptive field that can be staged (at most kStagedCols)
    const uint32_t floatn_lds = kFloatnRing + 256u + std::min<uint32_t>(L.plan->n_adaptive, kStagedCols) * (4u * 63u * 2u) * 4u + 64u;
#define LAUNCH_FLOATN(TT, LL, PP, ...)                                                                                          \
  hipLaunchKernelGGL((k_encode_floatn<TT, LL, PP, kFloatnRing, __VA_ARGS__>), dim3(L.n_chunks * L.subs), dim3(TT), floatn_lds, \
                     L.stream, *L.plan, L.points, L.points_end, L.chunks, L.slots, L.slot_stride, L.segs,          \
                     L.segs_per_chunk, L.cols, L.subs, L.sub_points, L.sub_stride, ablate)
    if (l3 == 4 && loadw == 8) LAUNCH_FLOATN(256, 4, 2, 8, false, false, 4);
    else if (l3 == 4 || loadw == 0) goto generic_regular;
    else if (unal && lanes == 3 && loadw == 4) LAUNCH_FLOATN(256, 3, 2, 4, false, true);
    else if (unal && lanes == 3 && loadw == 8) LAUNCH_FLOATN(256, 3, 2, 8, false, true);
    else if (unal && lanes == 4 && loadw == 5) LAUNCH_FLOATN(256, 4, 2, 5, false, true);
    else if (unal && lanes == 4 && loadw == 8) LAUNCH_FLOATN(256, 4, 2, 8, false, true);
    else if (unal) goto generic_regular;
    else if (lanes == 3 && loadw == 3) LAUNCH_FLOATN(256, 3, 2, 3, false);
    else if (lanes == 3 && loadw == 4) LAUNCH_FLOATN(256, 3, 2, 4, false);
    else if (lanes == 3 && loadw == 8) LAUNCH_FLOATN(256, 3, 2, 8, false);
    else if (lanes == 4 && loadw == 4) LAUNCH_FLOATN(256, 4, 2, 4, false);
    else if (lanes == 4 && loadw == 8) LAUNCH_FLOATN(256, 4, 2, 8, false);
    else {
    generic_regular:
      if (L.plan->point_step <= kWidePointStep)
        hipLaunchKernelGGL((k_encode_regular<kRegularThreads, false>), dim3(L.n_chunks * L.subs), dim3(kRegularThreads),
                           regular_lds(L.plan->point_step), L.stream, *L.plan, L.points, L.points_end, L.chunks, L.slots,
                           L.slot_stride, L.segs, L.segs_per_chunk, L.cols, L.subs, L.sub_points, L.sub_stride, L.pre);
      else
        hipLaunchKernelGGL((k_encode_regular<kRegularThreads, true>), dim3(L.n_chunks * L.subs), dim3(kRegularThreads),
                           regular_lds(L.plan->point_step), L.stream, *L.plan, L.points, L.points_end, L.chunks, L.slots,
                           L.slot_stride, L.segs, L.segs_per_chunk, L.cols, L.subs, L.sub_points, L.sub_stride, L.pre);
    }
#undef LAUNCH_FLOATN
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_encode_regular/floatn");
  }
  if (L.events) (void)hipEventRecord(L.events[2], L.stream);
  const uint32_t na = L.plan->n_adaptive;
  if (na && L.n_chunks) {
    static const bool no_fast = dev_env("CLDN_HIP_NO_FAST_SECTIONS") != nullptr;  // A/B switch: general kernels only
    if (!L.modes_forced && !modes_probed) {
      if (!no_fast) {
        hipLaunchKernelGGL(k_probe_fast, dim3(L.n_clouds, na), dim3(kS2Threads), kProbeLds, L.stream, *L.plan, L.chunks,
                           L.cloud_first_chunk, L.cols, L.modes);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_probe_fast");
      } else {  // A/B switch: the general probe decides every mode
        (void)hipMemsetAsync(L.modes, 0xff, (size_t)L.n_clouds * na, L.stream);
        hipLaunchKernelGGL(k_probe_modes, dim3(L.n_clouds, na), dim3(kSecThreads), kSecLdsTotal, L.stream, *L.plan,
                           L.chunks, L.cloud_first_chunk, L.cols, L.modes);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_probe_modes");
      }
    }
    const int rc_sec = launch_sections(L, L.stream, 0u, L.n_chunks, fused_field);
    if (rc_sec != CLDN_HIP_OK) return rc_sec;
  }
  if (L.events) (void)hipEventRecord(L.events[3], L.stream);
  if (L.chunks_only) {
    if (L.n_chunks) {
      hipLaunchKernelGGL(k_chunk_sizes, dim3((L.n_chunks + 255u) / 256u), dim3(256), 0, L.stream, L.segs, L.segs_per_chunk, L.n_chunks,
                         L.chunk_payload, L.contiguous_flag);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_chunk_sizes");
    }
    if (L.events) (void)hipEventRecord(L.events[4], L.stream);
    return CLDN_HIP_OK;
  }
  if (finish_mode != 0) {
    if (L.n_chunks == 0u) {  // no chunk, no workgroup: every cloud's stream is empty
      if ((e = hipMemsetAsync(L.stream_offsets, 0, (size_t)(L.n_clouds + 1u) * sizeof(uint64_t), L.stream)) != hipSuccess)
        return hip_fail(e, "hipMemsetAsync(stream_offsets)");
    } else {
      FinishArgs F;
      F.chunks = L.chunks;
      F.n_chunks = L.n_chunks;
      F.cloud_first_chunk = L.cloud_first_chunk;
      F.n_clouds = L.n_clouds;
      F.slots = L.slots;
      F.slot_stride = L.slot_stride;
      F.segs = L.segs;
      F.segs_per_chunk = L.segs_per_chunk;
      F.subs = L.subs;
      F.rec = L.fin_rec;
      F.rec2 = L.fin_rec2;
      F.anchor = L.fin_anchor;
      F.epoch = L.fin_epoch;
      F.ticket = L.fin_ticket;
      static const uint32_t use_ticket = (uint32_t)dev_env_int("CLDN_HIP_FINISH_TICKET", 0);
      static const uint32_t order = (uint32_t)dev_env_int("CLDN_HIP_FINISH_ORDER", 0);  // A/B switch
      F.use_ticket = (use_ticket || L.use_ticket) ? 1u : 0u;
      F.test_timeout = L.test_timeout;
      F.order = order;
      static const uint32_t copy_mode = (uint32_t)dev_env_int("CLDN_HIP_FINISH_COPY", 0);  // A/B switch
      static const uint32_t fin_ablate = (uint32_t)dev_env_int("CLDN_HIP_FINISH_ABLATE", 0);  // profiling only
      F.copy_mode = copy_mode;
      F.trace = nullptr;
      static const bool fin_trace = dev_env("CLDN_HIP_FINISH_TRACE") != nullptr;  // profiling only
      if (fin_trace) F.trace = stage1_finish_trace_buffer(L.n_chunks);
      F.ablate = fin_ablate;
      F.chunk_payload = L.chunk_payload;
      F.chunk_dst = L.chunk_dst;
      F.stream_offsets = L.stream_offsets;
      F.out = L.out;
      F.out_capacity = L.out_capacity;
      F.status = L.status;
      F.modes = L.modes;
      F.n_adaptive = na;
      F.fuse_field = fused_field;
      F.fuse_col = nullptr;
      F.fuse_first = nullptr;
      if (fused_field != kNoFusedField) {
        F.fuse_col = L.cols.p[fused_field];
        F.fuse_first = L.ranks[fused_field];
        // small batches: 1024-thread workgroups (a chunk's Palette section is latency-bound: twice the threads, 0.6x the time)
        static const uint32_t big_at = (uint32_t)dev_env_int("CLDN_HIP_FINISH_1024_BELOW", 200);  // A/B switch
        const bool big = L.n_chunks < big_at;
        static const uint32_t splits_env = (uint32_t)dev_env_int("CLDN_HIP_FINISH_SPLITS", 0);  // A/B switch
        const uint32_t splits = splits_env ? splits_env : (big ? 4u : (L.n_chunks >= 512u ? 1u : 2u));
        F.splits = splits;
        const bool u16 = L.plan->adaptive[fused_field].bpv == 2u;
        if (big && u16)
          hipLaunchKernelGGL((k_finish<1024, 2>), dim3(L.n_chunks * splits), dim3(1024), Pal32<uint16_t>::kLds, L.stream, F);
        else if (big)
          hipLaunchKernelGGL((k_finish<1024, 4>), dim3(L.n_chunks * splits), dim3(1024), Pal32<uint32_t>::kLds, L.stream, F);
        else if (u16)
          hipLaunchKernelGGL((k_finish<512, 2>), dim3(L.n_chunks * splits), dim3(512), Pal32<uint16_t>::kLds, L.stream, F);
        else
          hipLaunchKernelGGL((k_finish<512, 4>), dim3(L.n_chunks * splits), dim3(512), Pal32<uint32_t>::kLds, L.stream, F);
      } else {
        const uint32_t splits = L.n_chunks >= 1024u ? 1u : (L.n_chunks >= 256u ? 4u : 16u);
        F.splits = splits;
        static const uint32_t ldspad = (uint32_t)dev_env_int("CLDN_HIP_FINISH_LDSPAD", 0);  // experiment: occupancy
        hipLaunchKernelGGL((k_finish<256, 0>), dim3(L.n_chunks * splits), dim3(256), ldspad, L.stream, F);
      }
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_finish");
    }
    if (L.events) (void)hipEventRecord(L.events[4], L.stream);
    return CLDN_HIP_OK;
  }
#ifdef CLDN_DEV  // (CLDN_HIP_FINISH=0: the round-2 kernels, an A/B reference of the development build)
  hipLaunchKernelGGL(k_chunk_offsets<1024>, dim3(1), dim3(1024), 0, L.stream, L.segs, L.segs_per_chunk, L.n_chunks,
                     L.cloud_first_chunk, L.n_clouds, L.chunk_payload, L.chunk_dst, L.stream_offsets);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_chunk_offsets");
  if (L.n_chunks) {
    const uint32_t splits = L.n_chunks >= 1024u ? 1u : (L.n_chunks >= 256u ? 4u : 16u);
    hipLaunchKernelGGL(k_compact<256>, dim3(L.n_chunks, splits), dim3(256), 0, L.stream, L.slots, L.slot_stride,
                       L.segs, L.segs_per_chunk, L.chunk_payload, L.chunk_dst, L.out, L.out_capacity, L.status);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_compact");
  }
  if (L.events) (void)hipEventRecord(L.events[4], L.stream);
#endif
  return CLDN_HIP_OK;
}

int stage1_launch_frame(const FrameLaunch& L) {
  hipError_t e;
  if (L.n_chunks == 0u) {
    if ((e = hipMemsetAsync(L.stream_offsets, 0, (size_t)(L.n_clouds + 1u) * sizeof(uint64_t), L.stream)) != hipSuccess)
      return hip_fail(e, "hipMemsetAsync(stream_offsets)");
    return CLDN_HIP_OK;
  }
  FinishArgs F;
  F = FinishArgs{};
  F.chunks = L.chunks;
  F.n_chunks = L.n_chunks;
  F.cloud_first_chunk = L.cloud_first_chunk;
  F.n_clouds = L.n_clouds;
  F.slots = L.slots;
  F.slot_stride = L.slot_stride;
  F.segs = L.segs;
  F.segs_per_chunk = L.segs_per_chunk;
  F.subs = L.segs_per_chunk;
  F.rec = L.rec;
  F.rec2 = nullptr;
  F.anchor = L.anchor;
  F.epoch = L.epoch;
  F.ticket = L.ticket;
  F.use_ticket = L.use_ticket;
  F.test_timeout = L.test_timeout;
  F.order = 0u;
  F.copy_mode = 0u;
  F.ablate = 0u;
  F.trace = nullptr;
  F.chunk_payload = L.chunk_payload;
  F.chunk_dst = L.chunk_dst;
  F.stream_offsets = L.stream_offsets;
  F.out = L.out;
  F.out_capacity = L.out_capacity;
  F.status = L.status;
  F.fuse_field = kNoFusedField;
  F.splits = L.n_chunks >= 1024u ? 1u : (L.n_chunks >= 256u ? 4u : 16u);
  hipLaunchKernelGGL((k_finish<256, 0>), dim3(L.n_chunks * F.splits), dim3(256), 0, L.stream, F);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_finish (frame)");
  return CLDN_HIP_OK;
}

}  // namespace cldn
