// stage1_fused.h -- single-pass stage-1 encoder for schemas whose regular stream is one fused FloatN encoder
// (included by stage1_kernels.hip behind the section kernels).
//
// The slot pipeline (k_encode_floatn -> sections -> k_chunk_offsets -> k_compact) writes every regular byte twice:
// once into a worst-case-strided slot, once more when the chunks are packed. Here every regular byte is written
// ONCE, at its final position in the framed stream [u32 size][payload] (src/chunk_writer.cpp:27-48):
//
//   * a chunk is cut into PIECES of kPieceRows * 63 points; one wave encodes one piece, barrier-free, into its own
//     LDS region (sized for the worst case, 5 bytes per token), with the row arithmetic of k_encode_floatn
//     (src/field_encoder.cpp:42-91): lane l loads point l-1 of the row, the delta reference is lane l-1's value by DPP;
//   * the byte position of a piece is the sum of everything before it in the batch: the sizes of all earlier pieces,
//     4 bytes per chunk header and the V5 sections of every earlier chunk. That is a prefix sum over the pieces, done
//     inside the launch by decoupled look-back (aggregate / inclusive-prefix records, one 8-byte agent-scope word per
//     piece); pieces take their index from a ticket counter, so every predecessor of a piece has started;
//   * the section SIZES a chunk contributes are known before any section byte exists: every piece leaves per-field
//     statistics (sum of DeltaVarint token lengths; run heads, key bytes and long-run counts for Rle / DeltaRle; a
//     presence bitmap of its 16-bit values for Palette), and the chunk's last piece folds them with the size formulas
//     of src/v5_codec.cpp:258-316 (A.4 of SURVEY.md) once its siblings have arrived;
//   * integer fields still leave as SoA columns (the "AoS -> SoA channel split"); the section kernels encode them as
//     before and k_place_sections moves each section to the place the fused kernel reserved for it -- and checks
//     that its size is the size the statistics promised.
//
// Inter-workgroup protocol (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility"): every word
// another workgroup reads is written with an agent-scope atomic store / RMW (write-through) and read with an
// agent-scope atomic load; payload words are complete (s_waitcnt vmcnt(0)) before the word that announces them.
// Every spin is bounded: a wave that waits too long raises ST_FUSED_TIMEOUT and gives up (the call then fails
// loudly in cldn_hip_codec_status / on the host path) instead of hanging the GPU.
#pragma once

namespace cldn {

constexpr uint32_t kRowPts = 63;          // new points per wave row (lane 0 holds the point before them)
constexpr uint32_t kFusedWaves = 4;       // pieces per workgroup (all of one chunk: piece counts are padded to 4)
constexpr uint32_t kFusedThreads = kFusedWaves * 64;
constexpr uint32_t kBitmapWords = 2048;   // presence bitmap of a 16-bit field: 65536 bits
constexpr uint32_t kSpinLimit = 1u << 22; // polls (each >= ~0.3 us) before a wave gives up

__host__ __device__ constexpr uint32_t fused_piece_rows(int lanes) { return lanes == 3 ? 8u : 6u; }
__host__ __device__ constexpr uint32_t fused_piece_points(int lanes) { return fused_piece_rows(lanes) * kRowPts; }
// LDS bytes of one wave's stream region: worst case 5 bytes per token + slack for the aligned 16-byte reads of the
// copy-out
__host__ __device__ constexpr uint32_t fused_region_bytes(int lanes) {
  return ((fused_piece_points(lanes) * 5u * (uint32_t)lanes + 15u) & ~15u) + 32u;
}
// ... with one more token of up to kTailMaxBytes behind the FloatN tokens of every point (TAIL instantiations)
constexpr uint32_t kTailMaxBytes = 10;  // varint of an int64 delta, Gorilla token (13 + 64 bits), raw 8 bytes
__host__ __device__ constexpr uint32_t fused_region_bytes_tail(int lanes) {
  return ((fused_piece_points(lanes) * (5u * (uint32_t)lanes + kTailMaxBytes) + 15u) & ~15u) + 32u;
}

// look-back record: [63:62] state, [61:0] value
constexpr unsigned long long kLbAggregate = 1ull << 62;
constexpr unsigned long long kLbPrefix = 2ull << 62;
constexpr unsigned long long kLbValueMask = (1ull << 62) - 1ull;

// piece record, u32 units: [0] regular bytes, [1] spare, then 8 words per adaptive field
constexpr uint32_t kPrecHead = 2;
constexpr uint32_t kPrecField = 8;
enum : uint32_t { PF_LEN = 0, PF_HEADS = 1, PF_FIRST1 = 2, PF_LAST1 = 3, PF_L128 = 4, PF_L16K = 5 };

using gu32 = __attribute__((address_space(1))) uint32_t;
using gu64 = __attribute__((address_space(1))) unsigned long long;

__device__ __forceinline__ void ag_store64(void* p, unsigned long long v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ag_load64(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ag_store32(void* p, uint32_t v) {
  __hip_atomic_store(reinterpret_cast<uint32_t*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ag_load32(const void* p) {
  return __hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t x) {  // identity 0
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false));
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false));
  return x;
}

// value of lane l-1; lane 0 receives `carry` (wave_shr:1 leaves lane 0 untouched, so it keeps the `old` operand)
__device__ __forceinline__ uint32_t shr1_carry(uint32_t x, uint32_t carry) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint64_t shr1_carry64(uint64_t x, uint64_t carry) {
  const uint32_t lo = shr1_carry((uint32_t)x, (uint32_t)carry);
  const uint32_t hi = shr1_carry((uint32_t)(x >> 32), (uint32_t)(carry >> 32));
  return (((uint64_t)hi) << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t x, int lane) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), lane);
  return (((uint64_t)hi) << 32) | lo;
}

// one adaptive-int field of one point, straight from the AoS buffer (any alignment)
__device__ __forceinline__ uint64_t aos_field(const uint8_t* fp, uint32_t bpv) {
  uint64_t raw = 0u;
  if (((uintptr_t)fp & (bpv - 1u)) == 0u) {
    if (bpv == 2u) raw = *reinterpret_cast<const uint16_t*>(fp);
    else if (bpv == 4u) raw = *reinterpret_cast<const uint32_t*>(fp);
    else raw = *reinterpret_cast<const uint64_t*>(fp);
  } else {
    for (uint32_t b = 0; b < bpv; ++b) raw |= ((uint64_t)fp[b]) << (8u * b);
  }
  return raw;
}

// plan.adaptive[a] as two dwords (scalar loads: a byte-sized member at a run-time index would be fetched with a
// vector load from the kernel-argument segment, and every such load drags an s_waitcnt vmcnt(0) along)
__device__ __forceinline__ void adaptive_field(const DevPlan& plan, uint32_t a, uint32_t& offset, uint32_t& type, uint32_t& bpv) {
  static_assert(sizeof(DevAdaptive) == 8, "two dwords");
  uint32_t w[2];
  __builtin_memcpy(w, &plan.adaptive[a], 8);
  offset = w[0];
  type = w[1] & 0xffu;
  bpv = (w[1] >> 8) & 0xffu;
}

// token of <= 4 bytes OR-ed into a zeroed, linear LDS byte region at byte address `off`
__device__ __forceinline__ void lds_or4(uint8_t* region, uint32_t off, uint32_t t) {
  const uint64_t v = ((uint64_t)t) << ((off << 3) & 31u);
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  uint32_t* w = reinterpret_cast<uint32_t*>(region + (off & ~3u));
  atomicOr(w, lo);
  if (hi) atomicOr(w + 1, hi);
}
// token of <= 5 bytes (w1 holds byte 4)
__device__ __forceinline__ void lds_or5(uint8_t* region, uint32_t off, uint32_t w0, uint32_t w1, uint32_t len) {
  const uint32_t sh = (off & 3u) * 8u;
  const uint64_t v = ((((uint64_t)w1) << 32) | w0) << sh;
  const uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
  uint32_t* w = reinterpret_cast<uint32_t*>(region + (off & ~3u));
  if (lo) atomicOr(w, lo);
  if (((off & 3u) + len) > 4u) {
    if (hi) atomicOr(w + 1, hi);
    if (((off & 3u) + len) > 8u) {  // 5-byte token starting at byte 3 of a dword: its last byte is byte 0 of dword 2
      const uint32_t top = (uint32_t)(((uint64_t)w1 << sh) >> 32);
      if (top) atomicOr(w + 2, top);
    }
  }
}

// token of <= 12 bytes (Tok::w0..w2) OR-ed in at byte address `off`
__device__ __forceinline__ void lds_or12(uint8_t* region, uint32_t off, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t len) {
  const uint32_t sh = (off & 3u) * 8u;
  uint32_t* w = reinterpret_cast<uint32_t*>(region + (off & ~3u));
  const uint32_t o0 = w0 << sh;
  const uint32_t o1 = (uint32_t)((((uint64_t)w1 << 32) | w0) << sh >> 32);
  const uint32_t o2 = (uint32_t)((((uint64_t)w2 << 32) | w1) << sh >> 32);
  const uint32_t o3 = (uint32_t)(((uint64_t)w2 << sh) >> 32);
  const uint32_t span = (off & 3u) + len;  // bytes from the first dword's start to the token's end
  if (o0) atomicOr(w, o0);
  if (span > 4u && o1) atomicOr(w + 1, o1);
  if (span > 8u && o2) atomicOr(w + 2, o2);
  if (span > 12u && o3) atomicOr(w + 3, o3);
}

// bytes [rel, rel + 8) of the dwords loaded for one point, any rel (three dwords)
template <int LOADW>
__device__ __forceinline__ uint64_t field64_from_regs(const FloatVec<LOADW>& pt, uint32_t rel) {
  const uint32_t di = rel >> 2;
  uint32_t d0 = 0u, d1 = 0u, d2 = 0u;
#pragma unroll
  for (int k = 0; k < LOADW; ++k) {
    if ((uint32_t)k == di) d0 = __float_as_uint(pt.v[k]);
    if ((uint32_t)k == di + 1u) d1 = __float_as_uint(pt.v[k]);
    if ((uint32_t)k == di + 2u) d2 = __float_as_uint(pt.v[k]);
  }
  const uint32_t mis = rel & 3u;
  const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, mis), hi = __builtin_amdgcn_alignbyte(d2, d1, mis);
  return (((uint64_t)hi) << 32) | lo;
}

// One wave copies R bytes from its LDS region (16-byte aligned, 32 bytes of slack behind it) to dst (any alignment):
// byte stores up to the first aligned unit and behind the last one, in between 16-byte units assembled from two
// aligned LDS reads with a wave-uniform byte funnel.
__device__ __forceinline__ void copy_region_out(const uint8_t* region, uint32_t R, uint8_t* dst, uint32_t lane) {
  const uint32_t head = min(R, (uint32_t)((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u));
  const uint32_t body_units = (R - head) >> 4;
  const uint32_t tail = (R - head) & 15u;
  if (lane < head) dst[lane] = region[lane];
  if (lane >= 32u && lane < 32u + tail) {
    const uint32_t kb = head + body_units * 16u + (lane - 32u);
    dst[kb] = region[kb];
  }
  const uint32_t sdw = head >> 2, sb = head & 3u;  // wave-uniform
  const uint4* src4 = reinterpret_cast<const uint4*>(region);
  uint4* dst4 = reinterpret_cast<uint4*>(dst + head);
  for (uint32_t j = lane; j < body_units; j += 64u) {
    const uint4 a = src4[j];
    const uint4 b = src4[j + 1u];
    uint32_t w0, w1, w2, w3, w4;
    switch (sdw) {
      case 0: w0 = a.x; w1 = a.y; w2 = a.z; w3 = a.w; w4 = b.x; break;
      case 1: w0 = a.y; w1 = a.z; w2 = a.w; w3 = b.x; w4 = b.y; break;
      case 2: w0 = a.z; w1 = a.w; w2 = b.x; w3 = b.y; w4 = b.z; break;
      default: w0 = a.w; w1 = b.x; w2 = b.y; w3 = b.z; w4 = b.w; break;
    }
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(w1, w0, sb);
    o.y = __builtin_amdgcn_alignbyte(w2, w1, sb);
    o.z = __builtin_amdgcn_alignbyte(w3, w2, sb);
    o.w = __builtin_amdgcn_alignbyte(w4, w3, sb);
    dst4[j] = o;
  }
}

// Destination of every section, left by the fused kernel for k_place_sections
struct SecPlace {
  unsigned long long dst;  // byte offset in the output buffer
  uint32_t size;           // bytes the statistics promised
  uint32_t pad;
};

struct FusedArgs {
  const uint8_t* points;
  const uint8_t* points_end;
  const ChunkDesc* chunks;
  const PieceDesc* pieces;
  uint32_t n_pieces;
  FusedCtrl* ctrl;
  uint32_t* arrivals;            // [n_chunks], zero at launch
  unsigned long long* lb;        // [n_pieces], zero at launch: kLbAggregate | regular bytes of the piece
  unsigned long long* lbc;       // [n_chunks], zero at launch: chunk records of the look-back
  unsigned long long* start1;    // [n_chunks], zero at launch: chunk start + 1
  uint32_t* prec;                // [n_pieces * prec_stride]
  uint32_t prec_stride;
  uint32_t* bitmaps;             // [n_chunks * n_bm_fields * kBitmapWords], zero at launch, left zero
  uint32_t n_bm_fields;          // adaptive fields of 2 bytes (each has a bitmap slot)
  const uint8_t* modes;          // [n_clouds * n_adaptive]
  ColumnPtrs cols;
  uint8_t* out;
  unsigned long long out_capacity;
  uint32_t* chunk_payload;       // [n_chunks]
  unsigned long long* chunk_dst; // [n_chunks]
  SecPlace* secplace;            // [n_chunks * n_adaptive]
  uint32_t* status;
  uint32_t use_ticket;           // 1: piece order from a ticket counter instead of the workgroup index
  uint32_t slot_mode;            // 1: no inter-piece protocol -- every piece leaves its stream in its own range of the chunk slot
  uint8_t* slots;                //    (slot pipeline: sections, k_chunk_offsets and k_compact follow)
  unsigned long long slot_stride;
  uint32_t piece_stride;
  Seg* segs;
  uint32_t segs_per_chunk;
  // TAIL instantiations: one more regular op behind the FloatN lanes (raw copy, scalar lossy float, Gorilla token)
  uint32_t tail_kind;            // OP_* of plan.ops[LANES]
  uint32_t tail_rel;             // its offset behind plan.ops[0].offset (inside the LOADW dwords loaded per point)
  uint32_t tail_size;            // field bytes
  const uint16_t* tail_windows;  // OP_GORILLA64: k_gorilla_windows' window in front of every piece, [chunk * 128 + piece]
  uint32_t ablate;               // profiling only (CLDN_HIP_ABLATE): 1 no statistics pass, 2 no inter-piece protocol (fake positions), 4 no column stores
};

// UNAL / L3 / LOADW as in k_encode_floatn.
//
// Memory-level parallelism is explicit: the point loads of ALL rows of the piece are issued before the first row is
// touched, from straight-line code without a branch around any load (out-of-range lanes load a clamped address), so
// that the compiler's s_waitcnt counting stays exact -- a conditional load inside a loop made it fall back to
// vmcnt(0) before every row, which serialised the whole kernel on memory latency. The rows are unrolled for the same
// reason.
template <int LANES, int LOADW, bool UNAL, int L3, bool TAIL = false>
__global__ __launch_bounds__(kFusedThreads) void k_encode_fused(const DevPlan plan, const FusedArgs A) {
  constexpr uint32_t ROWS = fused_piece_rows(LANES);
  constexpr uint32_t PIECE = fused_piece_points(LANES);
  constexpr uint32_t REGION = TAIL ? fused_region_bytes_tail(LANES) : fused_region_bytes(LANES);
  constexpr uint32_t ROWS_B = (PIECE + 63u) / 64u;  // 64-wide rows of the statistics pass
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* wg_misc = reinterpret_cast<uint32_t*>(smem);                       // [0] ticket
  uint8_t* regions = smem + 16u;
  uint32_t* bitmap = reinterpret_cast<uint32_t*>(smem + 16u + kFusedWaves * REGION);  // [kBitmapWords] if n_bm_fields

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t na = plan.n_adaptive;

  // My piece. Workgroups are dispatched in index order, so every piece before mine has started (what the waits below
  // rely on); A.use_ticket replaces that assumption by a ticket counter (one contended atomic per workgroup).
  uint32_t wg = blockIdx.x;
  if (A.use_ticket) {
    if (tid == 0) wg_misc[0] = atomicAdd(&A.ctrl->ticket, 1u);
    __syncthreads();
    wg = __builtin_amdgcn_readfirstlane(wg_misc[0]);
  }
  const uint32_t g = wg * kFusedWaves + wave;
  const PieceDesc pd = A.pieces[g];           // one 32-byte record: no dependent second load
  const uint32_t P = pd.P;                    // pieces of the chunk (multiple of 4)
  const uint32_t p = pd.p;
  const uint32_t first = p * PIECE;           // chunk-relative index of my first point
  const uint32_t nc = pd.n_chunk_points;
  const uint32_t n = min(PIECE, nc - min(nc, first));  // my points (0 for a padding piece)
  const uint32_t step = plan.point_step;
  const size_t first_point = (size_t)pd.chunk_first_point + first;
  const uint8_t* gbase = A.points + first_point * step + plan.ops[0].offset;
  uint8_t* region = regions + wave * REGION;
  const uint8_t* cloud_modes = A.modes + (size_t)pd.cloud * na;

  // ---------------------------------------------------------------------------------------------------------
  // phase A: regular stream of the piece into `region`, integer fields into their columns
  // ---------------------------------------------------------------------------------------------------------
  FloatVec<LOADW> rows[ROWS];
  if (n) {
    // lane l of row r holds point r*63 + l - 1 of the piece (lane 0 = the point before the row). Out-of-range lanes
    // read point 0 of the piece instead: they never emit, and what they feed their neighbour is never emitted either.
    const bool guard = UNAL && (gbase + (size_t)(n + 1u) * step + 4u * (LOADW + 2) > A.points_end);  // uniform: batch tail
#pragma unroll
    for (uint32_t r = 0; r < ROWS; ++r) {
      const int32_t idx = (int32_t)(r * kRowPts + lane) - 1;
      const bool in_range = (idx >= 0 || first > 0u) && idx < (int32_t)n;
      // uniform base (one point before the piece) + 32-bit byte offset: scalar-base addressing, no 64-bit lane math
      const uint8_t* a = (gbase - step) + (in_range ? (uint32_t)(idx + 1) : 1u) * step;
      if (!UNAL) {
        rows[r] = *reinterpret_cast<const FloatVec<LOADW>*>(a);
      } else {
        const uint32_t mis = (uint32_t)((uintptr_t)a & 3u);
        const uint32_t* q = reinterpret_cast<const uint32_t*>(a - mis);
        uint32_t d[LOADW + 1];
        if (!guard) {
#pragma unroll
          for (int k = 0; k <= LOADW; ++k) d[k] = q[k];
        } else {  // last points of the batch: a dword is read only if it holds at least one byte of the buffer
#pragma unroll
          for (int k = 0; k <= LOADW; ++k) d[k] = (reinterpret_cast<const uint8_t*>(q + k) < A.points_end) ? q[k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < LOADW; ++k) rows[r].v[k] = __uint_as_float(__builtin_amdgcn_alignbyte(d[k + 1], d[k], mis));
      }
    }
  }

  // TAIL, Gorilla (FieldEncoderFloat_Gorilla<double>, include/cloudini_lib/field_encoder.hpp:156-312): the (leading,
  // trailing) window in front of my piece comes from k_gorilla_windows; from there the wave carries it row by row
  uint32_t gw_lead = 255u, gw_trail = 0u;
  if (TAIL && n && A.tail_kind == (uint32_t)OP_GORILLA64) {
    const uint32_t w16 = A.tail_windows[(size_t)pd.chunk * 128u + p];
    gw_lead = w16 & 0xffu;
    gw_trail = w16 >> 8;
  }

  // zero my stream region (tokens are OR-ed in) and, together, the bitmap -- while the loads fly
  {
    uint4* z = reinterpret_cast<uint4*>(region);
    for (uint32_t i = lane; i < REGION / 16u; i += 64u) z[i] = make_uint4(0u, 0u, 0u, 0u);
    if (A.n_bm_fields) {
      uint4* zb = reinterpret_cast<uint4*>(bitmap);
      for (uint32_t i = tid; i < kBitmapWords / 4u; i += kFusedThreads) zb[i] = make_uint4(0u, 0u, 0u, 0u);
    }
  }
  float mult[LANES];
#pragma unroll
  for (int k = 0; k < LANES; ++k) mult[k] = plan.ops[k].mult_f;

  uint32_t R = 0u;  // bytes of my stream so far (wave-uniform)
  auto run_rows = [&](auto nan_tier_tag) __attribute__((always_inline)) {
    constexpr bool NAN_TIER = decltype(nan_tier_tag)::value;
#pragma unroll
    for (uint32_t r = 0; r < ROWS; ++r) {
      if (r * kRowPts >= n) break;  // uniform: rows behind the piece's last point
      FloatVec<LOADW> cur = rows[r];
      if (r == 0u && first == 0u) {  // chunk start: the reference of the first point is 0 (lane 0 loaded point 0 instead)
#pragma unroll
        for (int k = 0; k < LOADW; ++k) cur.v[k] = lane == 0u ? 0.0f : cur.v[k];
      }
      const int32_t idx = (int32_t)(r * kRowPts + lane) - 1;
      const bool emits = (lane > 0u) && (idx < (int32_t)n);

      // tokens of the row (common case in the float domain, see k_encode_floatn)
      uint32_t tok[LANES], lens = 0u, total = 0u;
      bool rare = false;
#pragma unroll
      for (int k = 0; k < LANES; ++k) {
        const float rr = rintf(__fmul_rn(cur.v[(LANES == 4 && k == 3) ? L3 : k], mult[k]));
        rare |= !(fabsf(rr) < 2097152.0f);
        const float nd = __fsub_rn(__uint_as_float(dpp_wave_shr1(__float_as_uint(rr))), rr);
        const float uf = fabsf(__fmaf_rn(nd, -2.0f, 0.5f)) + 0.5f;
        const uint32_t u = (uint32_t)uf;
        const uint32_t l = groups7((uint32_t)__builtin_amdgcn_frexp_expf(uf));
        tok[k] = token4(u, l);
        lens |= l << (8 * k);
        total += l;
      }
      bool hard_row = false;
      if (__builtin_expect(__ballot(rare) != 0ull, 0)) {
        bool hard = !NAN_TIER;
        uint32_t total2 = 0u, lens2 = 0u;
#pragma unroll
        for (int k = 0; k < LANES; ++k) {
          if (!NAN_TIER) break;
          const float v = cur.v[(LANES == 4 && k == 3) ? L3 : k];
          const bool isn = is_nan_f32(v);
          const float rr = isn ? 0.0f : rintf(__fmul_rn(v, mult[k]));
          hard |= !(fabsf(rr) < 2097152.0f);
          const float nrp = __uint_as_float(dpp_wave_shr1(__float_as_uint(rr) ^ 0x80000000u));
          const float uf = fabsf(__fmaf_rn(__fadd_rn(rr, nrp), 2.0f, 0.5f)) + 0.5f;
          const uint32_t l = isn ? 1u : groups7((uint32_t)__builtin_amdgcn_frexp_expf(uf));
          tok[k] = isn ? 0u : token4((uint32_t)uf, l);
          lens2 |= l << (8 * k);
          total2 += l;
        }
        if (NAN_TIER && __ballot(hard) == 0ull) {
          lens = lens2;
          total = total2;
        } else {  // Inf, overflow or a 5-byte token somewhere in the row: general integer formulas
          hard_row = true;
          total = 0u;
#pragma unroll
          for (int k = 0; k < LANES; ++k) {
            const float v = cur.v[(LANES == 4 && k == 3) ? L3 : k];
            const bool isn = is_nan_f32(v);
            const int32_t q = quant_rne_i32(v, mult[k]);
            const uint32_t nqp = dpp_wave_shr1(isn ? 0u : (0u - (uint32_t)q));  // a NaN resets that lane's reference
            uint32_t a0, a1, l;
            floatn_token(isn, (int32_t)((uint32_t)q + nqp), a0, a1, l);
            total += l;
          }
        }
      }
      // TAIL: the op behind the FloatN lanes (include/cloudini_lib/field_encoder.hpp:56-60, :342-370, :156-312), from
      // the dwords already loaded for the point; lane l-1 holds the previous point (zeros in front of a chunk)
      Tok tt;
      tt.w0 = tt.w1 = tt.w2 = 0u;
      tt.len = 0u;
      if (TAIL) {
        const uint32_t kind = A.tail_kind;  // uniform
        const uint64_t raw = field64_from_regs<LOADW>(cur, A.tail_rel);
        if (kind == (uint32_t)OP_COPY) {
          tt = raw_tok(A.tail_size >= 8u ? raw : (raw & ((1ull << (8u * A.tail_size)) - 1ull)), A.tail_size);
        } else if (kind == (uint32_t)OP_LOSSY_F64) {
          const double v = __longlong_as_double((long long)raw);
          const bool isn = is_nan_f64(v);
          const int64_t q = isn ? 0 : quant_away_i64_f64(v, plan.ops[LANES].mult_d);  // a NaN resets the reference to 0
          const int64_t pq = (int64_t)shr1_carry64((uint64_t)q, 0u);
          if (isn) tt.len = 1u;  // marker byte 0x00
          else tt = varint64_tok((int64_t)((uint64_t)q - (uint64_t)pq));
        } else if (kind == (uint32_t)OP_LOSSY_F32) {
          const float v = __uint_as_float((uint32_t)raw);
          const bool isn = is_nan_f32(v);
          const int64_t q = isn ? 0 : quant_away_i64_f32(v, plan.ops[LANES].mult_f);
          const int64_t pq = (int64_t)shr1_carry64((uint64_t)q, 0u);
          if (isn) tt.len = 1u;
          else tt = varint64_tok((int64_t)((uint64_t)q - (uint64_t)pq));
        } else {  // OP_GORILLA64
          const bool chunk_first = first == 0u && idx == 0;          // the chunk's first value is written raw
          const uint64_t x = raw ^ shr1_carry64(raw, 0u);
          const uint32_t lead = x ? (uint32_t)__builtin_clzll(x) : 64u;
          const uint32_t trail = x ? (uint32_t)__builtin_ctzll(x) : 0u;
          // every lane assumes the current window; the lowest lane that would open a new one is resolved, the window is
          // updated, and only the lanes behind it re-check (lanes = points in order; lane 0 is the point before the row)
          bool pending = emits && !chunk_first && x != 0u;
          bool opens = false;
          uint32_t my_lead = gw_lead, my_trail = gw_trail;
          for (;;) {
            const bool would_open = pending && (gw_lead == 255u || lead < gw_lead || trail < gw_trail);
            const uint64_t ev = __ballot(would_open);
            if (ev == 0ull) {
              if (pending) {
                my_lead = gw_lead;
                my_trail = gw_trail;
              }
              break;
            }
            const uint32_t e = (uint32_t)__builtin_ctzll(ev);
            if (pending && lane <= e) {
              my_lead = gw_lead;
              my_trail = gw_trail;
              opens = (lane == e);
              pending = false;
            }
            const uint32_t le = (uint32_t)__builtin_amdgcn_readlane((int)lead, (int)e);
            gw_lead = le > 31u ? 31u : le;
            gw_trail = (uint32_t)__builtin_amdgcn_readlane((int)trail, (int)e);
          }
          uint64_t lo = 0u, hi = 0u;
          uint32_t nbits;
          if (chunk_first) {
            lo = raw;
            nbits = 64u;
          } else if (x == 0u) {
            nbits = 1u;  // single '0' bit
          } else if (!opens) {
            const uint32_t m = 64u - my_lead - my_trail;  // '1','0', m bits of (x >> trailing)
            const uint64_t payload = x >> my_trail;
            lo = 1u | (payload << 2);
            hi = payload >> 62;
            nbits = 2u + m;
          } else {
            const uint32_t sl = lead > 31u ? 31u : lead;  // '1','1', leading(5), m-1 (6), m bits of (x >> trailing)
            const uint32_t m = 64u - sl - trail;
            const uint64_t payload = x >> trail;
            lo = 3u | ((uint64_t)sl << 2) | ((uint64_t)(m - 1u) << 7) | (payload << 13);
            hi = payload >> 51;
            nbits = 13u + m;
          }
          tt.w0 = (uint32_t)lo;
          tt.w1 = (uint32_t)(lo >> 32);
          tt.w2 = (uint32_t)hi;
          tt.len = (nbits + 7u) >> 3;
        }
        total += tt.len;
      }
      const uint32_t plen = emits ? total : 0u;
      const uint32_t incl = wave_inclusive_scan(plen);
      uint32_t off = R + incl - plen;
      if (__builtin_expect(hard_row, 0)) {  // wave-uniform: rebuild the general tokens (all lanes take part in the DPP)
#pragma unroll
        for (int k = 0; k < LANES; ++k) {
          const float v = cur.v[(LANES == 4 && k == 3) ? L3 : k];
          const bool isn = is_nan_f32(v);
          const int32_t q = quant_rne_i32(v, mult[k]);
          const uint32_t nqp = dpp_wave_shr1(isn ? 0u : (0u - (uint32_t)q));
          uint32_t a0, a1, l;
          floatn_token(isn, (int32_t)((uint32_t)q + nqp), a0, a1, l);
          if (plen) lds_or5(region, off, a0, a1, l);
          off += l;
        }
      } else if (plen) {
#pragma unroll
        for (int k = 0; k < LANES; ++k) {
          lds_or4(region, off, tok[k]);
          off += (lens >> (8 * k)) & 0xffu;
        }
      }
      if (TAIL && plen) lds_or12(region, off, tt.w0, tt.w1, tt.w2, tt.len);
      R += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);

      // AoS -> SoA split of the adaptive-int fields (uniform loop: the plan is read with scalar loads)
      if (!(A.ablate & 4u)) {
        for (uint32_t a = 0; a < na; ++a) {
          uint32_t f_off, f_type, bpv;
          adaptive_field(plan, a, f_off, f_type, bpv);
          const uint32_t rel = f_off - plan.ops[0].offset;
          // uniform: the field lies inside the dwords loaded for the point (always, except for the padded-fourth-lane layout)
          const bool in_regs = LOADW > LANES && (L3 != 4 || (f_off >= plan.ops[0].offset && rel + bpv <= (uint32_t)LOADW * 4u));
          uint64_t raw;
          if (in_regs) {
            if (LOADW == LANES + 1) raw = __float_as_uint(cur.v[LOADW - 1]) >> ((rel & 3u) * 8u);
            else raw = field_from_regs<LOADW>(cur, rel);
          } else {
            raw = emits ? aos_field(A.points + first_point * step + (uint32_t)idx * step + f_off, bpv) : 0u;
          }
          // uniform base + 32-bit element index: the store takes the scalar-base addressing form, no 64-bit lane math
          uint8_t* colp = A.cols.p[a] + first_point * bpv;
          if (emits) {
            if (bpv == 2u) reinterpret_cast<uint16_t*>(colp)[idx] = (uint16_t)raw;
            else if (bpv == 4u) reinterpret_cast<uint32_t*>(colp)[idx] = (uint32_t)raw;
            else reinterpret_cast<uint64_t*>(colp)[idx] = raw;
          }
        }
      }
    }
  };
  if (n) {
    if (LANES == 3) {  // the 4-lane instantiations would lose occupancy to the second copy
      bool any_nan = false;
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < LANES; ++k) any_nan |= is_nan_f32(rows[b].v[k]);
      if (__ballot(any_nan) != 0ull) run_rows(std::true_type{});
      else run_rows(std::false_type{});
    } else {
      run_rows(std::false_type{});
    }
  }

  // ---------------------------------------------------------------------------------------------------------
  // phase B: per-field statistics of the piece (what the chunk's section of that field will weigh), from the column
  // values this wave has just written (workgroup-scope release/acquire: same CU, same L1)
  // ---------------------------------------------------------------------------------------------------------
  uint32_t* my_rec = A.prec + (size_t)g * A.prec_stride;
  if (na && !A.slot_mode && !(A.ablate & 1u)) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    drain_vmem();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (A.n_bm_fields) __syncthreads();  // the bitmap is zero for everybody
    uint32_t bm_slot = 0u;
    for (uint32_t a = 0; a < na; ++a) {
      uint32_t f_off, type, bpv;
      adaptive_field(plan, a, f_off, type, bpv);
      const uint32_t mode = cloud_modes[a];
      const uint32_t my_bm = bm_slot;
      if (bpv == 2u) ++bm_slot;
      // all values of the piece first (branch-free: lanes behind the end read value 0 of the piece) ...
      const uint8_t* col = A.cols.p[a] + (n ? first_point : (size_t)pd.chunk_first_point) * bpv;  // padding piece: any valid address
      uint64_t xs[ROWS_B];
      if (bpv == 2u) {  // one uniform branch around straight-line loads
#pragma unroll
        for (uint32_t rb = 0; rb < ROWS_B; ++rb) xs[rb] = reinterpret_cast<const uint16_t*>(col)[(rb * 64u + lane < n) ? rb * 64u + lane : 0u];
      } else if (bpv == 4u) {
#pragma unroll
        for (uint32_t rb = 0; rb < ROWS_B; ++rb) xs[rb] = reinterpret_cast<const uint32_t*>(col)[(rb * 64u + lane < n) ? rb * 64u + lane : 0u];
      } else {
#pragma unroll
        for (uint32_t rb = 0; rb < ROWS_B; ++rb) xs[rb] = reinterpret_cast<const uint64_t*>(col)[(rb * 64u + lane < n) ? rb * 64u + lane : 0u];
      }
      // ... and the two values before it, from the AoS input (their column entries belong to another workgroup)
      uint64_t vm1 = 0u, vm2 = 0u;
      if (n) {
        const uint8_t* fp = A.points + first_point * step + f_off;
        if (first >= 1u) vm1 = aos_field(fp - step, bpv);
        if (first >= 2u) vm2 = aos_field(fp - 2u * (size_t)step, bpv);
      }
      uint64_t carry_raw = vm1;
      uint64_t carry_diff = (uint64_t)int_field_as_i64(vm1, type) - (uint64_t)int_field_as_i64(vm2, type);
      uint32_t acc_len = 0u;                                                  // per lane
      uint32_t n_heads = 0u, first1 = 0u, last1 = 0u, l128 = 0u, l16k = 0u;  // wave-uniform
#pragma unroll
      for (uint32_t rb = 0; rb < ROWS_B; ++rb) {
        const uint32_t b0 = rb * 64u;
        if (b0 >= n) break;  // uniform
        const uint32_t i = b0 + lane;
        const bool valid = i < n;
        const uint64_t x = valid ? xs[rb] : 0u;
        const uint64_t px = shr1_carry64(x, carry_raw);
        const uint64_t d = (uint64_t)int_field_as_i64(x, type) - (uint64_t)int_field_as_i64(px, type);
        if (mode == 0u) {
          acc_len += valid ? varint64_len((int64_t)d) : 0u;
        } else if (mode == 1u) {
          if (bpv == 2u && valid) atomicOr(&bitmap[(uint32_t)x >> 5], 1u << ((uint32_t)x & 31u));
        } else {
          bool head;
          if (mode == 2u) {
            head = valid && ((first + i) == 0u || x != px);
          } else {
            const uint64_t pd_ = shr1_carry64(d, carry_diff);
            head = valid && ((first + i) == 0u || d != pd_);
            acc_len += head ? varint64_len((int64_t)d) : 0u;
          }
          const uint64_t M = __ballot(head);
          if (M != 0ull) {
            const uint32_t f1 = first + b0 + (uint32_t)__builtin_ctzll(M) + 1u;         // chunk index + 1
            const uint32_t l1 = first + b0 + (63u - (uint32_t)__builtin_clzll(M)) + 1u;
            if (last1 != 0u) {
              const uint32_t gap = f1 - last1;
              l128 += gap >= 128u ? 1u : 0u;
              l16k += gap >= 16384u ? 1u : 0u;
            } else {
              first1 = f1;
            }
            last1 = l1;
            n_heads += (uint32_t)__builtin_popcountll(M);
          }
        }
        carry_raw = readlane64(x, 63);
        carry_diff = readlane64(d, 63);
      }
      const uint32_t len_sum = wave_sum(acc_len);
      if (lane == 0u) {
        uint32_t* fr = my_rec + kPrecHead + a * kPrecField;
        ag_store64(fr + PF_LEN, ((unsigned long long)n_heads << 32) | len_sum);       // PF_LEN, PF_HEADS
        ag_store64(fr + PF_FIRST1, ((unsigned long long)last1 << 32) | first1);       // PF_FIRST1, PF_LAST1
        ag_store64(fr + PF_L128, ((unsigned long long)l16k << 32) | l128);            // PF_L128, PF_L16K
      }
      if (mode == 1u && bpv == 2u) {
        // fold the workgroup's bitmap into the chunk's (and leave the LDS copy zero for the next palette field)
        __syncthreads();
        uint32_t* gbm = A.bitmaps + ((size_t)pd.chunk * A.n_bm_fields + my_bm) * kBitmapWords;
        for (uint32_t w = tid; w < kBitmapWords; w += kFusedThreads) {
          const uint32_t bits = bitmap[w];
          if (bits) {
            __hip_atomic_fetch_or(gbm + w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bitmap[w] = 0u;
          }
        }
        __syncthreads();
      }
    }
  }
  if (A.slot_mode) {
    // slot pipeline: the four streams of the workgroup go back to back into the workgroup's range of the chunk slot
    // (one segment of ~11 KB for k_compact instead of four small ones)
    if (lane == 0u) wg_misc[wave] = R;  // the ticket word is not used in this mode
    __syncthreads();
    uint32_t before = 0u, total = 0u;
#pragma unroll
    for (uint32_t w = 0; w < kFusedWaves; ++w) {
      const uint32_t rw = wg_misc[w];
      before += w < wave ? rw : 0u;
      total += rw;
    }
    const uint32_t quad = p >> 2;
    const uint32_t seg_off = quad * kFusedWaves * A.piece_stride;
    if (R) copy_region_out(region, R, A.slots + (size_t)pd.chunk * A.slot_stride + seg_off + before, lane);
    if (wave == 0u && lane == 0u) {
      Seg sg;
      sg.off = seg_off;
      sg.size = total;
      A.segs[(size_t)pd.chunk * A.segs_per_chunk + quad] = sg;
    }
    return;
  }
  if (A.ablate & 2u) {  // profiling only: no protocol, every piece writes into a private worst-case range
    if (R == 0u) return;
    uint8_t* dst = A.out + (size_t)g * (REGION / 2u);
    const uint4* src4 = reinterpret_cast<const uint4*>(region);
    for (uint32_t j = lane; j < ((R + 15u) >> 4); j += 64u) reinterpret_cast<uint4*>(dst)[j] = src4[j];
    return;
  }
  if (lane == 0u) ag_store64(A.lb + g, kLbAggregate | (unsigned long long)R);  // my regular bytes, for my chunk's siblings
  drain_vmem();  // statistics, bitmap ORs and the size word are performed before the arrival is announced

  // bounded spin helper: true = the awaited condition never came (status raised)
  uint32_t spins = 0u;
  auto spin_fail = [&]() -> bool {
    __builtin_amdgcn_s_sleep(8);
    if (++spins < kSpinLimit && ((spins & 1023u) != 0u || ag_load32(&A.ctrl->timeout) == 0u)) return false;
    if (lane == 0u) {
      ag_store32(&A.ctrl->timeout, 1u);
      atomicOr(A.status, (uint32_t)ST_FUSED_TIMEOUT);
    }
    return true;
  };

  // ---------------------------------------------------------------------------------------------------------
  // Positions, two levels (a flat look-back over ~4000 pieces in flight would walk dozens of 64-record windows):
  //   inside the chunk   piece p starts behind the regular bytes of pieces 0..p-1 of its chunk (<= 2 windows of lb)
  //   between chunks     the chunk's LAST piece waits for its siblings, turns the statistics into section sizes,
  //                      publishes the chunk total, finds the chunk's start by decoupled look-back over the chunk
  //                      records (aggregate = total, prefix = start + total) and hands the start to its siblings
  //                      through start1[chunk] (one word, polled by one lane per wave)
  // ---------------------------------------------------------------------------------------------------------
  const uint32_t g0 = g - p;  // first piece of my chunk
  const uint32_t c = pd.chunk;
  uint32_t reg_sum = 0u, sec_total = 0u, sec_rel = 0u, sec_size = 0u;
  unsigned long long chunk_start = 0ull;

  // regular bytes of the pieces before me in the chunk (for the last piece: of all its siblings, which have arrived)
  uint32_t before_me = 0u;
  auto sum_siblings = [&](uint32_t count, bool may_wait) -> bool {  // sum of R over pieces [g0, g0 + count)
    uint32_t acc = 0u;
    for (uint32_t b = 0; b < count; b += 64u) {
      const uint32_t q = b + lane;
      unsigned long long x = 0ull;
      for (;;) {
        x = q < count ? ag_load64(A.lb + g0 + q) : kLbAggregate;
        if (__ballot((x >> 62) == 0ull) == 0ull) break;
        if (!may_wait || spin_fail()) return false;
      }
      acc += wave_sum((uint32_t)(x & 0xffffffffull));
    }
    before_me = acc;
    return true;
  };

  if (p + 1u < P) {
    if (lane == 0u) __hip_atomic_fetch_add(A.arrivals + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    for (;;) {
      if ((A.ablate & 64u) || ag_load32(A.arrivals + c) == P - 1u) break;
      if (spin_fail()) return;
    }
    if (!(A.ablate & 64u) && !sum_siblings(P - 1u, true)) return;  // all published (they arrived); the wait path is only a safety net
    reg_sum = before_me + R;
    uint32_t bm_slot = 0u;
    for (uint32_t a = 0; a < ((A.ablate & 16u) ? 0u : na); ++a) {
      uint32_t f_off, f_type, bpv;
      adaptive_field(plan, a, f_off, f_type, bpv);
      const uint32_t mode = cloud_modes[a];
      const uint32_t my_bm = bm_slot;
      if (bpv == 2u) ++bm_slot;
      uint32_t size = 0u;
      if (mode == 1u) {  // Palette: 3 + U*bpv + ceil(bits*n/8)   (v5_codec.cpp:298-306)
        uint32_t* gbm = A.bitmaps + ((size_t)c * A.n_bm_fields + my_bm) * kBitmapWords;
        unsigned long long bits[kBitmapWords / 128u];
#pragma unroll
        for (uint32_t k = 0; k < kBitmapWords / 128u; ++k) bits[k] = ag_load64(gbm + lane * 2u + k * 128u);  // all in flight
        uint32_t cnt = 0u;
#pragma unroll
        for (uint32_t k = 0; k < kBitmapWords / 128u; ++k) {
          cnt += (uint32_t)__builtin_popcountll(bits[k]);
          if (bits[k]) ag_store64(gbm + lane * 2u + k * 128u, 0ull);  // leave the bitmap zero for the next call
        }
        const uint32_t U = wave_sum(cnt);
        size = 3u + U * bpv + ((palette_bits(U) * nc + 7u) >> 3);
      } else {
        uint32_t len_sum = 0u, heads = 0u, l128 = 0u, l16k = 0u, carry_last1 = 0u;
        for (uint32_t b = 0; b < P; b += 64u) {
          const uint32_t q = b + lane;
          unsigned long long w0 = 0ull, w1 = 0ull, w2 = 0ull;
          if (q < P) {
            const uint32_t* fr = A.prec + (size_t)(g0 + q) * A.prec_stride + kPrecHead + a * kPrecField;
            w0 = ag_load64(fr + PF_LEN);
            w1 = ag_load64(fr + PF_FIRST1);
            w2 = ag_load64(fr + PF_L128);
          }
          len_sum += wave_sum((uint32_t)w0);
          heads += wave_sum((uint32_t)(w0 >> 32));
          const uint32_t f1 = (uint32_t)w1, l1 = (uint32_t)(w1 >> 32);
          const uint32_t incl = wave_inclusive_max(l1);
          const uint32_t prev1 = max(carry_last1, shr1_carry(incl, 0u));  // newest head before my piece (+1), 0 = none
          uint32_t c128 = (uint32_t)w2, c16k = (uint32_t)(w2 >> 32);
          if (f1 != 0u && prev1 != 0u) {  // the run that was open when my piece began ends at my first head
            const uint32_t gap = f1 - prev1;
            c128 += gap >= 128u ? 1u : 0u;
            c16k += gap >= 16384u ? 1u : 0u;
          }
          l128 += wave_sum(c128);
          l16k += wave_sum(c16k);
          carry_last1 = max(carry_last1, (uint32_t)__builtin_amdgcn_readlane((int)incl, 63));
        }
        if (mode == 0u) {
          size = 1u + len_sum;  // v5_codec.cpp:263-267
        } else {
          const uint32_t tail = nc + 1u - carry_last1;  // the last run ends with the chunk (point 0 is always a head)
          l128 += tail >= 128u ? 1u : 0u;
          l16k += tail >= 16384u ? 1u : 0u;
          const uint32_t keys = (mode == 2u) ? heads * bpv : len_sum;
          size = 5u + keys + heads + l128 + l16k;  // v5_codec.cpp:269-296, run lengths <= 32768: 1..3 bytes
        }
      }
      if (lane == a) {  // lane a keeps section a's place (na <= kMaxAdaptive <= 64 lanes), written once the chunk start is known
        sec_rel = sec_total;
        sec_size = size;
      }
      sec_total += size;
    }
    const unsigned long long total = 4ull + reg_sum + sec_total;

    // decoupled look-back over the chunks
    if (c == 0u || (A.ablate & 32u)) {
      chunk_start = 0ull;
    } else {
      if (lane == 0u) ag_store64(A.lbc + c, kLbAggregate | total);
      uint32_t base = c;  // the window is [base - 64, base)
      for (;;) {
        const bool valid = lane < base;
        const unsigned long long* slot = A.lbc + (valid ? (base - 1u - lane) : 0u);
        unsigned long long x;
        uint64_t m_prefix;
        for (;;) {
          x = valid ? ag_load64(slot) : kLbPrefix;  // before the first chunk: prefix 0
          const uint32_t st = (uint32_t)(x >> 62);
          m_prefix = __ballot(st == 2u);
          const uint64_t m_ready = __ballot(st != 0u);
          const uint64_t need = m_prefix ? ((1ull << __builtin_ctzll(m_prefix)) - 1ull) : ~0ull;  // lanes closer than the prefix
          if ((m_ready & need) == need) break;
          if (spin_fail()) return;
        }
        const uint32_t k = m_prefix ? (uint32_t)__builtin_ctzll(m_prefix) : 64u;  // totals of lanes < k, prefix of lane k
        // chunk totals are < 2^26 (32768 points * (regular + sections)): 64 of them fit 32 bits
        chunk_start += (unsigned long long)wave_sum(lane < k ? (uint32_t)(x & kLbValueMask) : 0u);
        if (k < 64u) {
          chunk_start += readlane64(x, (int)k) & kLbValueMask;
          break;
        }
        base -= 64u;
      }
    }
    if (lane == 0u) {
      ag_store64(A.lbc + c, kLbPrefix | (chunk_start + total));
      ag_store64(A.start1 + c, chunk_start + 1ull);
    }
  }

  if ((A.ablate & 8u) && p + 1u < P) {  // profiling only: no waiting for siblings / chunk start
    if (R == 0u) return;
    uint8_t* dst = A.out + (size_t)g * (REGION / 2u);
    const uint4* src4 = reinterpret_cast<const uint4*>(region);
    for (uint32_t j = lane; j < ((R + 15u) >> 4); j += 64u) reinterpret_cast<uint4*>(dst)[j] = src4[j];
    return;
  }
  if (p + 1u < P) {
    if (!sum_siblings(p, true)) return;  // regular bytes in front of me inside the chunk
    unsigned long long s1 = 0ull;
    for (;;) {
      s1 = ag_load64(A.start1 + c);  // one word; every lane reads the same address (one request)
      if (s1 != 0ull) break;
      if (spin_fail()) return;
    }
    chunk_start = s1 - 1ull;
  }

  // ---------------------------------------------------------------------------------------------------------
  // placement: chunk header and tables (last piece), then my bytes from LDS to their final position
  // ---------------------------------------------------------------------------------------------------------
  const unsigned long long my_dst = chunk_start + 4ull + before_me;
  if (p + 1u == P) {
    const uint32_t payload = reg_sum + sec_total;
    if (chunk_start + 4ull + payload > A.out_capacity) {
      if (lane == 0u) atomicOr(A.status, (uint32_t)ST_OUT_OVERFLOW);
    } else {
      if (lane < 4u) A.out[chunk_start + lane] = (uint8_t)(payload >> (8u * lane));
      if (lane == 0u) {
        A.chunk_payload[c] = payload;
        A.chunk_dst[c] = chunk_start;
      }
      if (lane < na) {
        SecPlace sp;
        sp.dst = chunk_start + 4ull + reg_sum + sec_rel;
        sp.size = sec_size;
        sp.pad = 0u;
        A.secplace[(size_t)c * na + lane] = sp;
      }
    }
  }
  if (R == 0u) return;
  if (my_dst + R > A.out_capacity) {
    if (lane == 0u) atomicOr(A.status, (uint32_t)ST_OUT_OVERFLOW);
    return;
  }
  copy_region_out(region, R, A.out + my_dst, lane);
}

// ------------------------------------------------------------------------------------------------------------
// k_probe_extract: the probe decides the adaptive-int modes on the first <= 4096 values of a cloud
// (src/v5_codec.cpp:934-949) BEFORE the fused kernel runs, so those values are copied into the columns ahead of it
// (the fused kernel writes the same values again). grid = (n_clouds, n_adaptive), 1024 threads.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_probe_extract(const DevPlan plan, const uint8_t* __restrict__ points,
                                                        const ChunkDesc* __restrict__ chunks,
                                                        const uint32_t* __restrict__ cloud_first_chunk, const ColumnPtrs cols) {
  const uint32_t cloud = blockIdx.x, a = blockIdx.y;
  const uint32_t fc = cloud_first_chunk[cloud];
  if (fc == cloud_first_chunk[cloud + 1u]) return;  // empty cloud
  const ChunkDesc cd = chunks[fc];
  const uint32_t n = cd.n_points > kProbePoints ? kProbePoints : cd.n_points;
  const uint32_t bpv = plan.adaptive[a].bpv;
  for (uint32_t i = threadIdx.x; i < n; i += 1024u) {
    const size_t gi = (size_t)cd.first_point + i;
    const uint64_t raw = aos_field(points + gi * plan.point_step + plan.adaptive[a].offset, bpv);
    uint8_t* col = cols.p[a];
    if (bpv == 2u) reinterpret_cast<uint16_t*>(col)[gi] = (uint16_t)raw;
    else if (bpv == 4u) reinterpret_cast<uint32_t*>(col)[gi] = (uint32_t)raw;
    else reinterpret_cast<uint64_t*>(col)[gi] = raw;
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_place_sections: grid = (n_chunks, n_adaptive). Moves the two segments of a section (slot layout of the section
// kernels) to the position the fused kernel reserved, byte-exact at any destination alignment, and checks the size
// promise. Block (0, 0) also derives the per-cloud stream offsets from the chunk destinations.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void copy_bytes_wg(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t size,
                                              uint32_t tid, uint32_t nthreads) {
  // src 16-byte aligned, dst arbitrary
  const uint32_t head = min(size, (uint32_t)((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u));
  const uint32_t body_units = (size - head) >> 4;
  const uint32_t tail = (size - head) & 15u;
  if (tid < head) dst[tid] = src[tid];
  if (tid >= 32u && tid < 32u + tail) {
    const uint32_t kb = head + body_units * 16u + (tid - 32u);
    dst[kb] = src[kb];
  }
  const uint32_t sdw = head >> 2, sb = head & 3u;
  const uint4* src4 = reinterpret_cast<const uint4*>(src);
  uint4* dst4 = reinterpret_cast<uint4*>(dst + head);
  for (uint32_t j = tid; j < body_units; j += nthreads) {
    const uint4 a = src4[j];
    uint4 b = make_uint4(0u, 0u, 0u, 0u);
    if (head != 0u) b = src4[j + 1u];  // the slot has slack behind every segment
    uint32_t w0, w1, w2, w3, w4;
    switch (sdw) {
      case 0: w0 = a.x; w1 = a.y; w2 = a.z; w3 = a.w; w4 = b.x; break;
      case 1: w0 = a.y; w1 = a.z; w2 = a.w; w3 = b.x; w4 = b.y; break;
      case 2: w0 = a.z; w1 = a.w; w2 = b.x; w3 = b.y; w4 = b.z; break;
      default: w0 = a.w; w1 = b.x; w2 = b.y; w3 = b.z; w4 = b.w; break;
    }
    uint4 o;
    o.x = __builtin_amdgcn_alignbyte(w1, w0, sb);
    o.y = __builtin_amdgcn_alignbyte(w2, w1, sb);
    o.z = __builtin_amdgcn_alignbyte(w3, w2, sb);
    o.w = __builtin_amdgcn_alignbyte(w4, w3, sb);
    dst4[j] = o;
  }
}

__global__ __launch_bounds__(256) void k_place_sections(const uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                        const Seg* __restrict__ segs, uint32_t segs_per_chunk, uint32_t subs,
                                                        uint32_t n_adaptive, const SecPlace* __restrict__ secplace,
                                                        const uint32_t* __restrict__ chunk_payload,
                                                        const unsigned long long* __restrict__ chunk_dst,
                                                        uint32_t n_chunks, const uint32_t* __restrict__ cloud_first_chunk,
                                                        uint32_t n_clouds, unsigned long long* __restrict__ stream_offsets,
                                                        uint8_t* __restrict__ out, unsigned long long out_capacity,
                                                        const FusedCtrl* __restrict__ ctrl, uint32_t* __restrict__ status) {
  if (ctrl->need_fallback != 0u || ctrl->timeout != 0u) return;
  const uint32_t c = blockIdx.x, a = blockIdx.y;
  if (c == 0u && a == 0u) {
    // stream offset of cloud k = destination of its first chunk (clouds without chunks inherit the next one)
    const unsigned long long total = n_chunks ? chunk_dst[n_chunks - 1u] + 4ull + chunk_payload[n_chunks - 1u] : 0ull;
    for (uint32_t k = threadIdx.x; k <= n_clouds; k += 256u) {
      const uint32_t fc = (k < n_clouds) ? cloud_first_chunk[k] : n_chunks;
      stream_offsets[k] = (fc < n_chunks) ? chunk_dst[fc] : total;
    }
  }
  if (c >= n_chunks || a >= n_adaptive) return;
  const SecPlace sp = secplace[(size_t)c * n_adaptive + a];
  const Seg s0 = segs[(size_t)c * segs_per_chunk + subs + 2u * a];
  const Seg s1 = segs[(size_t)c * segs_per_chunk + subs + 1u + 2u * a];
  if (s0.size + s1.size != sp.size) {  // the statistics and the section kernel disagree: never silently
    if (threadIdx.x == 0) atomicOr(status, (uint32_t)ST_FUSED_MISMATCH);
    return;
  }
  if (sp.dst + sp.size > out_capacity) {
    if (threadIdx.x == 0) atomicOr(status, (uint32_t)ST_OUT_OVERFLOW);
    return;
  }
  const uint8_t* slot = slots + (size_t)c * slot_stride;
  copy_bytes_wg(slot + s0.off, out + sp.dst, s0.size, threadIdx.x, 256u);
  copy_bytes_wg(slot + s1.off, out + sp.dst + s0.size, s1.size, threadIdx.x, 256u);
}

}  // namespace cldn
