// stage1_fused.h -- piece kernel: the regular stream of schemas whose per-point encoders are one fused FloatN encoder
// (optionally followed by one more op), included by stage1_kernels.hip behind the section kernels.
//
//   * a chunk is cut into PIECES of kPieceRows * 63 points; one wave encodes one piece, barrier-free, into its own
//     LDS region (sized for the worst case, 5 bytes per token), with the row arithmetic of
//     FieldEncoderFloatN_Lossy::encode (src/field_encoder.cpp:42-91): lane l loads point l-1 of the row, the delta
//     reference is lane l-1's value by DPP;
//   * the four pieces of a workgroup leave as ONE segment of the chunk's slot ({offset, size} in the segment table);
//     k_finish (stage1_finish.h) places the segments and the sections behind the chunk's [u32 size];
//   * integer fields leave as SoA columns (the "AoS -> SoA channel split") for the section kernels.
//
// The single-pass variant of round 2 (every byte placed by an in-kernel look-back over pieces, section sizes from
// per-piece statistics) was removed in round 3: it was byte-exact but 2.5x slower than slots + compaction, because a
// piece cannot be placed before the Palette sizes of every earlier chunk of its cloud are known (negative
// results).
#pragma once

namespace cldn {

constexpr uint32_t kRowPts = 63;          // new points per wave row (lane 0 holds the point before them)
constexpr uint32_t kFusedWaves = 4;       // pieces per workgroup (all of one chunk: piece counts are padded to 4)
constexpr uint32_t kFusedThreads = kFusedWaves * 64;

__host__ __device__ constexpr uint32_t fused_piece_rows(int lanes) { return lanes == 3 ? 8u : 6u; }
__host__ __device__ constexpr uint32_t fused_piece_points(int lanes) { return fused_piece_rows(lanes) * kRowPts; }
// LDS bytes of one wave's stream region: worst case 5 bytes per token + slack for the aligned 16-byte reads of the
// copy-out
__host__ __device__ constexpr uint32_t fused_region_bytes(int lanes) {
  return ((fused_piece_points(lanes) * 5u * (uint32_t)lanes + 15u) & ~15u) + 32u;
}
// Round 5: the instantiations without a TAIL op size the region for 3 bytes per token (18.3 KB per workgroup instead of 30.4:
// more workgroups per CU). A piece whose tokens do not fit -- deltas of 2^20 ticks and more on average, i.e. noise over
// kilometres at 1 mm -- is written by fused_slow_piece straight from the input instead (same bytes, slowly).
__host__ __device__ constexpr uint32_t fused_region_cap_small(int lanes) {
  return (fused_piece_points(lanes) * 3u * (uint32_t)lanes + 15u) & ~15u;
}
__host__ __device__ constexpr uint32_t fused_region_bytes_small(int lanes) { return fused_region_cap_small(lanes) + 32u; }
// ... with one more token of up to kTailMaxBytes behind the FloatN tokens of every point (TAIL instantiations)
constexpr uint32_t kTailMaxBytes = 10;  // varint of an int64 delta, Gorilla token (13 + 64 bits), raw 8 bytes
__host__ __device__ constexpr uint32_t fused_region_bytes_tail(int lanes) {
  return ((fused_piece_points(lanes) * (5u * (uint32_t)lanes + kTailMaxBytes) + 15u) & ~15u) + 32u;
}

// ceil(bits / 7) for bits < 2^16 on the full-rate 24-bit multiply-add (groups7's 32-bit product becomes a v_mad_u64_u32,
// which issues at a quarter of the rate)
__device__ __forceinline__ uint32_t groups7_u24(uint32_t bits) {
  return (__umul24(bits, 37u) + 222u) >> 8;
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
  atomicOr(w + 1, hi);  // (a zero OR costs an LDS cycle; the test around it a compare, two exec-mask instructions and a branch)
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

struct FusedArgs {
  const uint8_t* points;
  const uint8_t* points_end;
  const PieceDesc* pieces;
  ColumnPtrs cols;
  uint8_t* slots;                // every workgroup leaves its four pieces back to back in its range of the chunk slot
  unsigned long long slot_stride;
  uint32_t piece_stride;         // bytes reserved per piece inside the slot
  Seg* segs;
  uint32_t segs_per_chunk;
  // TAIL instantiations: one more regular op behind the FloatN lanes (raw copy, scalar lossy float, Gorilla token)
  uint32_t tail_kind;            // OP_* of plan.ops[LANES]
  uint32_t tail_rel;             // its offset behind plan.ops[0].offset (inside the LOADW dwords loaded per point)
  uint32_t tail_size;            // field bytes
  const uint16_t* tail_windows;  // OP_GORILLA64: k_gorilla_windows' window in front of every piece, [chunk * 128 + piece]
  uint32_t ablate;               // profiling only (CLDN_HIP_ABLATE): 4 no column stores
  // n_probe != 0: the first n_probe workgroups of the grid are not pieces: workgroup b decides the adaptive-int mode of
  // (cloud b / n_adaptive, field b % n_adaptive) on the first <= 4096 values of the cloud, read from the AoS input
  // (analyzeAdaptiveIntField + selectBestAdaptiveIntMode, src/v5_codec.cpp:387-412, :934-949) -- next to the pieces
  // instead of in a launch of its own behind them (10 us + a launch boundary per call)
  uint32_t n_probe;
  uint32_t probe_lds;            // bytes of dynamic LDS the launch has
  const ChunkDesc* chunks;
  const uint32_t* cloud_first_chunk;
  uint8_t* modes;
  // intra != 0: the workgroups of a chunk place their streams back to back at the start of the chunk's slot (ONE
  // regular segment per chunk): every workgroup publishes its byte count as {epoch, bytes} and adds up the records of
  // its chunk's workgroups before it (at most 16, one load per lane; agent-scope store / loads, bounded spin)
  uint32_t intra;
  uint32_t epoch;
  unsigned long long* wgrec;     // [n_chunks * 32]
  uint32_t* status;
};

// UNAL / L3 / LOADW as in k_encode_floatn.
//
// Memory-level parallelism is explicit: the point loads of ALL rows of the piece are issued before the first row is
// touched, from straight-line code without a branch around any load (out-of-range lanes load a clamped address), so
// that the compiler's s_waitcnt counting stays exact -- a conditional load inside a loop made it fall back to
// vmcnt(0) before every row, which serialised the whole kernel on memory latency. The rows are unrolled for the same
// reason.
template <int LANES, int LOADW, bool UNAL, int L3, bool TAIL>
__device__ __forceinline__ void fused_body(const DevPlan& plan, const FusedArgs& A) {
  constexpr uint32_t ROWS = fused_piece_rows(LANES);
  constexpr uint32_t PIECE = fused_piece_points(LANES);
  constexpr bool SMALL = !TAIL;  // 3 bytes per token, overflowing pieces go to the slow path below
  constexpr uint32_t REGION = TAIL ? fused_region_bytes_tail(LANES) : (SMALL ? fused_region_bytes_small(LANES) : fused_region_bytes(LANES));
  constexpr uint32_t CAP = REGION - 32u;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* wg_misc = reinterpret_cast<uint32_t*>(smem);                       // [wave] bytes of the wave's stream
  uint8_t* regions = smem + 16u;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t na = plan.n_adaptive;

  if (blockIdx.x < A.n_probe) {  // uniform: a mode-probe workgroup
    const uint32_t cloud = blockIdx.x / na, a = blockIdx.x - cloud * na;
    uint8_t* mode_out = A.modes + (size_t)cloud * na + a;
    const uint32_t fc = A.cloud_first_chunk[cloud];
    if (fc == A.cloud_first_chunk[cloud + 1u]) {  // empty cloud
      if (tid == 0u) *mode_out = 0u;
      return;
    }
    const ChunkDesc cd = A.chunks[fc];
    const uint32_t n = cd.n_points > kProbePoints ? kProbePoints : cd.n_points;
    uint32_t f_off, f_type, bpv;
    adaptive_field(plan, a, f_off, f_type, bpv);
    const uint8_t* fp = A.points + (size_t)cd.first_point * plan.point_step + f_off;
    const uint32_t pstep = plan.point_step;
    uint32_t* wtot = reinterpret_cast<uint32_t*>(smem + ((A.probe_lds - 256u) & ~15u));
    const uint32_t slots = (A.probe_lds - 272u) / 4u;  // 32-bit keys: >= 6144 slots for <= 4096 values
    uint8_t mode;
    if (bpv == 2u)
      mode = probe_mode_t<uint16_t, (int)kFusedThreads>([&](uint32_t i) { return (uint16_t)aos_field(fp + (size_t)i * pstep, 2u); }, n, f_type,
                                                        smem, 0u, wtot);
    else
      mode = probe_mode_t<uint32_t, (int)kFusedThreads>([&](uint32_t i) { return (uint32_t)aos_field(fp + (size_t)i * pstep, 4u); }, n, f_type,
                                                        smem, slots, wtot);
    if (tid == 0u) *mode_out = mode;
    return;
  }
  const uint32_t g = (blockIdx.x - A.n_probe) * kFusedWaves + wave;  // my piece
  const PieceDesc pd = A.pieces[g];           // one 32-byte record: no dependent second load
  const uint32_t p = pd.p;
  const uint32_t first = p * PIECE;           // chunk-relative index of my first point
  const uint32_t nc = pd.n_chunk_points;
  const uint32_t n = min(PIECE, nc - min(nc, first));  // my points (0 for a padding piece)
  const uint32_t step = plan.point_step;
  const size_t first_point = (size_t)pd.chunk_first_point + first;
  const uint8_t* gbase = A.points + first_point * step + plan.ops[0].offset;
  uint8_t* region = regions + wave * REGION;

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
        // non-temporal: the points are read once, and left out of L2 / Infinity Cache they do not push out the slot
        // and column data this kernel writes for k_finish (same-box A/B, C2: step 0.296 -> 0.288 ms, k_finish -6 %;
        // non-temporal STORES of slots or columns cost what this gains). Scalar loads: the compiler merges them.
        // Not for the padded layouts (L3 == 4): their integer fields may lie outside the loaded dwords and are then
        // read from the same lines a second time (Ouster-style 48-byte points: 0.336 -> 0.410 ms with these loads bypassing).
        const float* f = reinterpret_cast<const float*>(a);
#pragma unroll
        for (int k = 0; k < LOADW; ++k) rows[r].v[k] = (L3 != 4) ? __builtin_nontemporal_load(f + k) : f[k];
      } else {
        const uint32_t mis = (uint32_t)((uintptr_t)a & 3u);
        const uint32_t* q = reinterpret_cast<const uint32_t*>(a - mis);
        uint32_t d[LOADW + 1];
        if (!guard) {
#pragma unroll
          for (int k = 0; k <= LOADW; ++k) d[k] = __builtin_nontemporal_load(q + k);
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

  // zero my stream region (tokens are OR-ed in) -- while the loads fly
  {
    uint4* z = reinterpret_cast<uint4*>(region);
    for (uint32_t i = lane; i < REGION / 16u; i += 64u) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  float mult[LANES];
#pragma unroll
  for (int k = 0; k < LANES; ++k) mult[k] = plan.ops[k].mult_f;

  // AoS -> SoA split of one adaptive-int field for the row's points; all arguments but `cur`, `emits`, `idx` are uniform.
  // rel = field offset behind plan.ops[0].offset, colp = the column at the piece's first point.
  auto col_store = [&](uint32_t f_off, uint32_t rel, uint32_t bpv, bool in_regs, uint8_t* colp, const FloatVec<LOADW>& cur, bool emits,
                       int32_t idx) __attribute__((always_inline)) {
    uint64_t raw;
    if (in_regs) {
      if (LOADW == LANES + 1) raw = __float_as_uint(cur.v[LOADW - 1]) >> ((rel & 3u) * 8u);
      else raw = field_from_regs<LOADW>(cur, rel);
    } else {
      raw = emits ? aos_field(A.points + first_point * step + (uint32_t)idx * step + f_off, bpv) : 0u;
    }
    // uniform base + 32-bit element index: the store takes the scalar-base addressing form, no 64-bit lane math
    if (emits) {
      if (bpv == 2u) reinterpret_cast<uint16_t*>(colp)[idx] = (uint16_t)raw;
      else if (bpv == 4u) reinterpret_cast<uint32_t*>(colp)[idx] = (uint32_t)raw;
      else reinterpret_cast<uint64_t*>(colp)[idx] = raw;
    }
  };
  auto col_args = [&](uint32_t a, uint32_t& f_off, uint32_t& rel, uint32_t& bpv, bool& in_regs, uint8_t*& colp) __attribute__((always_inline)) {
    uint32_t f_type;
    adaptive_field(plan, a, f_off, f_type, bpv);
    rel = f_off - plan.ops[0].offset;
    // uniform: the field lies inside the dwords loaded for the point (always, except for the padded-fourth-lane layout)
    in_regs = LOADW > LANES && (L3 != 4 || (f_off >= plan.ops[0].offset && rel + bpv <= (uint32_t)LOADW * 4u));
    colp = A.cols.p[a] + first_point * bpv;
  };
  // the first fields' arguments are worked out once (round 5: the row loop fetched the plan entry with scalar loads and
  // decoded it again for every row -- 45 SALU instructions and a scalar-cache round trip per row and field)
  constexpr uint32_t kHoistCols = 2;
  uint32_t hc_off[kHoistCols ? kHoistCols : 1], hc_rel[kHoistCols ? kHoistCols : 1], hc_bpv[kHoistCols ? kHoistCols : 1];
  bool hc_in[kHoistCols ? kHoistCols : 1];
  uint8_t* hc_ptr[kHoistCols ? kHoistCols : 1];
#pragma unroll
  for (uint32_t a = 0; a < kHoistCols; ++a) {
    hc_off[a] = hc_rel[a] = hc_bpv[a] = 0u;
    hc_in[a] = false;
    hc_ptr[a] = nullptr;
    if (a < na) col_args(a, hc_off[a], hc_rel[a], hc_bpv[a], hc_in[a], hc_ptr[a]);
  }

  uint32_t R = 0u;  // bytes of my stream so far (wave-uniform)
  bool ovf = false; // SMALL: the piece's tokens do not fit the region (wave-uniform); nothing more is written to it
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
      const bool emits = (lane > 0u) && (lane < n + 1u - r * kRowPts);  // idx < n, as one compare against a uniform limit

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
        const uint32_t l = groups7_u24((uint32_t)__builtin_amdgcn_frexp_expf(uf));
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
          const uint32_t l = isn ? 1u : groups7_u24((uint32_t)__builtin_amdgcn_frexp_expf(uf));
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
      const uint32_t row_bytes = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      if (SMALL) ovf |= (R + row_bytes > CAP);
      if (SMALL && __builtin_expect(ovf, 0)) {
        // (the piece is rewritten by the slow path below: only its size is still counted)
      } else if (__builtin_expect(hard_row, 0)) {  // wave-uniform: rebuild the general tokens (all lanes take part in the DPP)
#pragma unroll
        for (int k = 0; k < LANES; ++k) {
          const float v = cur.v[(LANES == 4 && k == 3) ? L3 : k];
          const bool isn = is_nan_f32(v);
          const int32_t q = quant_rne_i32(v, mult[k]);
          const uint32_t nqp = dpp_wave_shr1(isn ? 0u : (0u - (uint32_t)q));
          uint32_t a0, a1, l;
          floatn_token(isn, (int32_t)((uint32_t)q + nqp), a0, a1, l);
          if (emits) lds_or5(region, off, a0, a1, l);
          off += l;
        }
      } else if (emits) {
#pragma unroll
        for (int k = 0; k < LANES; ++k) {
          lds_or4(region, off, tok[k]);
          off += (lens >> (8 * k)) & 0xffu;
        }
      }
      if (TAIL && emits) lds_or12(region, off, tt.w0, tt.w1, tt.w2, tt.len);
      R += row_bytes;

      // AoS -> SoA split of the adaptive-int fields
      if (!(A.ablate & 4u)) {
#pragma unroll
        for (uint32_t a = 0; a < kHoistCols; ++a)
          if (a < na) col_store(hc_off[a], hc_rel[a], hc_bpv[a], hc_in[a], hc_ptr[a], cur, emits, idx);
        for (uint32_t a = kHoistCols; a < na; ++a) {  // uniform loop: the plan is read with scalar loads
          uint32_t f_off, rel, bpv;
          bool in_regs;
          uint8_t* colp;
          col_args(a, f_off, rel, bpv, in_regs, colp);
          col_store(f_off, rel, bpv, in_regs, colp, cur, emits, idx);
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

  // SMALL, a piece that did not fit its region: the same R bytes once more, straight from the input to `dst` with the
  // general integer formulas (FieldEncoderFloatN_Lossy::encode, src/field_encoder.cpp:42-91), one point per lane and
  // row, byte stores. Rare by construction (more than 3 bytes per token on average over 378 / 504 points).
  auto slow_piece = [&](uint8_t* dst) __attribute__((always_inline)) {
    uint32_t RR = 0u;
    for (uint32_t r0 = 0; r0 < n; r0 += kRowPts) {
      const uint32_t idx = r0 + lane - 1u;
      const bool emits = (lane > 0u) && (idx < n);
      uint32_t w0[LANES], w1[LANES], l[LANES], tot = 0u;
#pragma unroll
      for (int k = 0; k < LANES; ++k) {
        w0[k] = w1[k] = l[k] = 0u;
        if (emits) {
          const uint8_t* fp = A.points + (first_point + idx) * step + plan.ops[k].offset;
          const float v = __uint_as_float((uint32_t)aos_field(fp, 4u));
          int32_t qp = 0;
          if (first + idx > 0u) {  // the chunk's first point has the reference 0; a NaN resets it to 0
            const float pv = __uint_as_float((uint32_t)aos_field(fp - step, 4u));
            qp = is_nan_f32(pv) ? 0 : quant_rne_i32(pv, mult[k]);
          }
          const int32_t q = quant_rne_i32(v, mult[k]);
          floatn_token(is_nan_f32(v), (int32_t)((uint32_t)q - (uint32_t)qp), w0[k], w1[k], l[k]);
        }
        tot += l[k];
      }
      const uint32_t incl = wave_inclusive_scan(tot);
      uint32_t off = RR + incl - tot;
#pragma unroll
      for (int k = 0; k < LANES; ++k) {
        const uint64_t t = (((uint64_t)w1[k]) << 32) | w0[k];
        for (uint32_t b = 0; b < l[k]; ++b) dst[off + b] = (uint8_t)(t >> (8u * b));
        off += l[k];
      }
      RR += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  };
  auto write_out = [&](uint8_t* dst) __attribute__((always_inline)) {
    if (SMALL && __builtin_expect(ovf, 0)) slow_piece(dst);
    else copy_region_out(region, R, dst, lane);
  };

  // the four streams of the workgroup go back to back into the workgroup's range of the chunk slot
  // (one segment of ~11 KB for k_compact instead of four small ones)
  if (lane == 0u) wg_misc[wave] = R;
  __syncthreads();
  uint32_t before = 0u, total = 0u;
#pragma unroll
  for (uint32_t w = 0; w < kFusedWaves; ++w) {
    const uint32_t rw = wg_misc[w];
    before += w < wave ? rw : 0u;
    total += rw;
  }
  const uint32_t quad = p >> 2;
  if (A.intra) {
    // The piece table lists the chunk's workgroups in ascending order, and the hardware hands out a grid's workgroups
    // in index order: the records this workgroup waits for belong to workgroups that have started -- with the
    // quad-major table of a large batch, a whole generation of workgroups earlier. The spin is bounded all the same.
    unsigned long long* rec = A.wgrec + (size_t)pd.chunk * 32u;  // my chunk's records (<= 22 workgroups per chunk)
    if (wave == 0u && lane == 0u)
      __hip_atomic_store(rec + quad, ((unsigned long long)A.epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t mine = 0u;
    for (uint32_t spins = 0;; ++spins) {
      const unsigned long long x = lane < quad ? __hip_atomic_load(rec + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                               : ((unsigned long long)A.epoch << 32);
      mine = (uint32_t)x;
      if (__ballot((uint32_t)(x >> 32) != A.epoch) == 0ull) break;
      if (spins >= (1u << 22)) {
        if (lane == 0u) atomicOr(A.status, (uint32_t)ST_FINISH_TIMEOUT);
        return;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    const uint32_t wg_off = wave_sum(mine);  // bytes of the chunk's workgroups before mine
    if (R) write_out(A.slots + (size_t)pd.chunk * A.slot_stride + wg_off + before);
    if (wave == 0u && lane == 0u && quad + 1u == (pd.P >> 2)) {  // the chunk's last workgroup: the regular stream is complete
      Seg sg;
      sg.off = 0u;
      sg.size = wg_off + total;
      A.segs[(size_t)pd.chunk * A.segs_per_chunk] = sg;
    }
    return;
  }
  const uint32_t seg_off = quad * kFusedWaves * A.piece_stride;
  if (R) write_out(A.slots + (size_t)pd.chunk * A.slot_stride + seg_off + before);
  if (wave == 0u && lane == 0u) {
    Seg sg;
    sg.off = seg_off;
    sg.size = total;
    A.segs[(size_t)pd.chunk * A.segs_per_chunk + quad] = sg;
  }
}

template <int LANES, int LOADW, bool UNAL, int L3, bool TAIL = false>
__global__ __launch_bounds__(kFusedThreads) void k_encode_fused(const DevPlan plan, const FusedArgs A) {
  fused_body<LANES, LOADW, UNAL, L3, TAIL>(plan, A);
}
// Round 5: the instantiations that prefetch at most 5 dwords per point and row and have no TAIL op fit 64 VGPRs without a
// spill; with the 18.3 KB of four 3-byte-per-token regions a CU then holds 8 workgroups = 8 waves per SIMD instead of 5
// (same-box A/B, C2: 136 -> 130 us). The wider ones would spill (up to 88 VGPRs' worth) and keep the compiler's choice.
template <int LANES, int LOADW, bool UNAL, int L3>
__global__ __launch_bounds__(kFusedThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_encode_fused_w8(const DevPlan plan,
                                                                                                            const FusedArgs A) {
  fused_body<LANES, LOADW, UNAL, L3, false>(plan, A);
}

}  // namespace cldn
