// stage1_decode_wave.h -- k_decode_points_w: the barrier-free point decoder (round 4; token path rebuilt in round 6). Chunks whose
// regular stream is one fused FloatN encoder (3 or 4 int32-delta varint tokens per point; FieldDecoderFloatN_Lossy::decode,
// src/field_decoder.cpp:43-86; decodeVarint, include/cloudini_lib/encoding_utils.hpp:98-148), Palette sections folded in /
// integer fields taken from columns.
//
// A chunk's payload is cut into PIECES of 62 aligned 16-byte units (992 bytes) at fixed addresses and ONE WAVE decodes a piece
// from its first byte to its last store without meeting a barrier; the waves of the workgroup take the pieces round robin. Lanes
// 1..62 hold the piece's units, lane 0 the unit in FRONT of it (a token that ends in lane 1 may begin there; lane l takes what
// it needs of lane l - 1 by DPP), lane 63 the first unit of the NEXT piece (the halo: the last point a piece owns may end there).
//   phase V  (round 6) BYTE-PARALLEL token values. A token is owned by the unit its LAST byte lies in. Per lane: the four dwords'
//            7-bit groups are packed once (28 bits per dword); for each of the 16 byte positions the 4 groups that end there are
//            one v_alignbit of two packed dwords, the token's length comes from the 3 end flags below the position through a
//            v_perm table look-up, one shift leaves the token's value u = zigzag + 1. The value goes to LDS slot
//            (ends in front of the lane) + (ends below the position): the piece's tokens lie in stream order, one dword each,
//            and positions that end no token write to a per-lane dummy word (v_bfi on the address: no exec changes). Nothing is
//            parsed serially, no point-start list, no per-token find-first-set / funnel shift chain as in rounds 4-5.
//   chain 1  T0 = how many token ends lie in front of the piece (popcount + one DPP scan; known before phase V ends). A piece OWNS
//            the points whose FIRST token ends in it: point j of the piece has its tokens in slots k0 + NOPS * j + o, with
//            k0 = (-T0) mod NOPS; the last of them may be halo tokens (at most NOPS - 1, within 4 * (NOPS - 1) bytes).
//   phase B  one point per lane: NOPS slot reads -> zigzag -> DPP prefix sums (a DPP segmented scan when a row holds a NaN
//            marker, u == 0) -> values relative to the piece's start.
//   chain 2  the running values in front of the piece's first point (int32 per lane of the FloatN encoder, NaN markers
//            reset a lane). A piece publishes its aggregate and adds the carry right before the conversion to float and
//            the stores. A lane stores its own point, so a store instruction covers 64 consecutive points.
// Every chain record is one aligned 8-byte LDS word {tag = piece + 1, value}: no ordering between LDS operations is assumed.
// A wave waits only for the piece in front of its own, which belongs to the neighbouring wave at the same step of its
// loop -- all waves of a workgroup are resident, so the waits cannot deadlock; they are bounded all the same.
// Irregular chunks are handed back (reg_end = kDecRedo -> k_decode_varint / the serial decoder decode them or raise the errors):
// a token longer than 4 bytes, a 0x00 byte that ends a token of more than one byte (an overlong zero -- decodeVarint rejects
// it -- or another non-canonical form the encoder never writes), a short stream. Both are seen on the units' end / zero masks.
#pragma once

namespace cldn {

constexpr uint32_t kWpUnits = 62u;                // owned 16-byte units per piece: lanes 1..62 (lane 0: the unit in front, lane 63: the halo)
constexpr uint32_t kWpPiece = kWpUnits * 16u;     // bytes of stream per piece
constexpr uint32_t kWpSlots = 1024u;              // token slots per wave: <= 992 owned tokens + <= 16 of the halo unit
constexpr uint32_t kWpRing = 64u;                 // chain records (slot = piece % kWpRing, tagged)
constexpr uint32_t kWpSpinLimit = 1u << 18;
constexpr int kWpSleep = 4;                       // x 64 cycles between two polls of a record (64 / 128 cycles: no difference, round 6)
// the hop from a chain record's arrival to the piece's own record runs at the highest wave priority (round 6: C4 -3 %, C5 -2 %)
#define WP_BOOST(P) __builtin_amdgcn_s_setprio(P)
#define WP_BOOST_BACK() __builtin_amdgcn_s_setprio(3)

template <int NOPS>
struct WpGeom {
  static constexpr uint32_t kMaxPts = (kWpPiece + (uint32_t)NOPS - 1u) / (uint32_t)NOPS;  // points a piece can own
  static constexpr uint32_t kRows = (kMaxPts + 63u) / 64u;                                  // 6 (3 lanes), 4 (4 lanes)
};

template <int NOPS, int NF, int NW>
struct WpLds {
  static constexpr uint32_t kValsOff = 0u;                                       // u32 [NW][kWpSlots]
  static constexpr uint32_t kDummyOff = (uint32_t)NW * kWpSlots * 4u;            // u32 [64]: where positions that end no token write
  static constexpr uint32_t kTrecOff = kDummyOff + 256u;                         // u64 [kWpRing]
  static constexpr uint32_t kVrecOff = kTrecOff + kWpRing * 8u;                  // u64 [kWpRing][NOPS]: running values behind a piece
  static constexpr uint32_t kPalOff = kVrecOff + kWpRing * (uint32_t)NOPS * 8u;
  static constexpr uint32_t kMiscOff = kPalOff + (uint32_t)(NF > 2 ? 0 : NF) * kFastPalEntries * 4u;  // (NF > 2: columns only, nothing is folded)
  static constexpr uint32_t kTotal = kMiscOff + 512u;
};

// bit j = byte j of the 16 ends a token (MSB clear): one v_dot4_u32_u8 per dword gathers the four MSBs
__device__ __forceinline__ uint32_t wp_ends16(const uint32_t (&b)[4]) {
  uint32_t lo = __builtin_amdgcn_udot4(b[0] & 0x80808080u, 0x08040201u, 0u, false);
  lo = __builtin_amdgcn_udot4(b[1] & 0x80808080u, 0x80402010u, lo, false);
  uint32_t hi = __builtin_amdgcn_udot4(b[2] & 0x80808080u, 0x08040201u, 0u, false);
  hi = __builtin_amdgcn_udot4(b[3] & 0x80808080u, 0x80402010u, hi, false);
  return ~((lo >> 7) | (hi << 1)) & 0xffffu;  // (the sums are 128 x the masks of the bytes that CONTINUE)
}

// bit j = byte j of the 16 is NOT 0x00 (same gather; (x & 0x7f) + 0x7f carries into the MSB of a byte with value bits)
__device__ __forceinline__ uint32_t wp_nonzero16(const uint32_t (&b)[4]) {
  uint32_t nz[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) nz[k] = (((b[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | b[k]) & 0x80808080u;
  uint32_t lo = __builtin_amdgcn_udot4(nz[0], 0x08040201u, 0u, false);
  lo = __builtin_amdgcn_udot4(nz[1], 0x80402010u, lo, false);
  uint32_t hi = __builtin_amdgcn_udot4(nz[2], 0x08040201u, 0u, false);
  hi = __builtin_amdgcn_udot4(nz[3], 0x80402010u, hi, false);
  return ((lo >> 7) | (hi << 1)) & 0xffffu;
}

// the four 7-bit groups of a dword side by side: byte 0's group in bits 0..6, byte 3's in bits 21..27
__device__ __forceinline__ uint32_t wp_pack7(uint32_t w) {
  const uint32_t lo = w & 0x7f7f7f7fu;
  const uint32_t x1 = lo - ((lo & 0x7f007f00u) >> 1);  // 7-bit groups -> 14-bit groups
  return x1 - __umul24(x1 >> 16, 49152u);              // -> 28 bits
}

// value of lane l - 1 (wave_shr:1; lane 0 reads 0)
__device__ __forceinline__ uint32_t wp_from_lane_below(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true);
}

// compiler-level ordering of a wave's own LDS traffic (the hardware keeps a wave's LDS operations in order)
__device__ __forceinline__ void wp_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long wp_rec_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void wp_rec_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// inclusive segmented scan over the 64 lanes: inc[o] = sum of my lane's and the lanes' in front of me since the last lane
// whose bit o of `fin` is set (that lane's own value included); fin becomes the OR of the flags up to my lane
template <int NOPS>
__device__ __forceinline__ void wp_seg_scan(int32_t (&inc)[NOPS], uint32_t& fin) {
#define WP_SEG_STEP(CTRL, RMASK, BC)                                                                              \
  {                                                                                                               \
    const uint32_t of = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)fin, CTRL, RMASK, 0xf, BC);                 \
    int32_t ov[NOPS];                                                                                             \
    _Pragma("unroll") for (int o = 0; o < NOPS; ++o) ov[o] = __builtin_amdgcn_update_dpp(0, inc[o], CTRL, RMASK, 0xf, BC); \
    _Pragma("unroll") for (int o = 0; o < NOPS; ++o)                                                              \
      if (!(fin & (1u << o))) inc[o] = (int32_t)((uint32_t)inc[o] + (uint32_t)ov[o]);                             \
    fin |= of;                                                                                                    \
  }
  WP_SEG_STEP(0x111, 0xf, true)   // row_shr:1 (lanes without a source read 0: nothing to add, no flag)
  WP_SEG_STEP(0x112, 0xf, true)   // row_shr:2
  WP_SEG_STEP(0x114, 0xf, true)   // row_shr:4
  WP_SEG_STEP(0x118, 0xf, true)   // row_shr:8
  WP_SEG_STEP(0x142, 0xa, false)  // row_bcast:15 -> rows 1, 3
  WP_SEG_STEP(0x143, 0xc, false)  // row_bcast:31 -> rows 2, 3
#undef WP_SEG_STEP
}

// phase V, byte position I of the lane's unit: the value of the token that ends there (if one does) -> its slot.
//   pk_lo / pk_hi  the packed groups of dword I / 4 - 1 and I / 4 as one 56-bit number (low / high dword)
//   e20            end flags: bit 4 + j = byte j of the unit, bits 0..3 = the four bytes in front of it
//   addr           LDS byte address of the slot of the lane's next token; advanced when position I ends one
//   dummy          LDS byte address of the lane's dummy word
// The 4 groups that end with byte I are bits 7j + 7 .. 7j + 34 of the 56-bit number (j = I % 4): one v_alignbit leaves them in
// bits 4..31. The token has 1 + (continuation bytes right below I, at most 3) bytes: a v_perm look-up turns the three end flags
// below I into the shift that drops the groups of other tokens (32 - 7 * length; length 4 also when the token is longer --
// such chunks are handed back).
//   ev             the ends of the unit that are tokens of the payload (bit j = byte j)
template <int I>
__device__ __forceinline__ void wp_value_at(const uint32_t (&pk_lo)[4], const uint32_t (&pk_hi)[4], uint32_t e20, uint32_t ev,
                                            uint32_t& addr, uint32_t dummy) {
  constexpr int K = I / 4, J = I % 4;
  const uint32_t x = __builtin_amdgcn_alignbit(pk_hi[K], pk_lo[K], 7 * J + 3);
  const uint32_t below = __builtin_amdgcn_ubfe(e20, I + 1, 3);  // bit 2: byte I - 1 ends a token, bit 1: I - 2, bit 0: I - 3
  const uint32_t sh = __builtin_amdgcn_perm(0x19191919u, 0x12120b04u, below);  // {4, 11, 18, 18, 25, 25, 25, 25}[below]
  const uint32_t u = x >> (sh & 31u);
  const uint32_t is_end = (uint32_t)__builtin_amdgcn_sbfe((int)ev, I, 1);  // all ones: byte I ends a token
  uint32_t where;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(where) : "v"(is_end), "v"(addr), "v"(dummy));  // is_end ? addr : dummy
  *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(where) = u;
  asm("v_mad_i32_i24 %0, %1, -4, %0" : "+v"(addr) : "v"(is_end));  // + 4 behind an end
}

// phase V for one unit: the packed groups of the unit's four dwords (pk) -> the values of the tokens that end in the unit, each
// into its slot (addr: the slot of the unit's first token)
__device__ __forceinline__ void wp_scatter16(const uint32_t (&pk)[4], uint32_t e20, uint32_t ev, uint32_t addr, uint32_t dummy) {
  uint32_t pk_lo[4], pk_hi[4];
  const uint32_t pk_front = wp_from_lane_below(pk[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    pk_lo[k] = (pk[k] << 28) | (k ? pk[k ? k - 1 : 0] : pk_front);
    pk_hi[k] = pk[k] >> 4;
  }
  wp_value_at<0>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<1>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<2>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<3>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<4>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<5>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<6>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<7>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<8>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<9>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<10>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<11>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<12>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<13>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<14>(pk_lo, pk_hi, e20, ev, addr, dummy);
  wp_value_at<15>(pk_lo, pk_hi, e20, ev, addr, dummy);
}

// the bytes of a unit that hand the chunk back (see the head of the file): the end of a token of 5 bytes or more, a 0x00 byte
// behind a byte that continues a token. e20: the end flags of the unit and of the four bytes in front of it; ev: the ends asked about
__device__ __forceinline__ uint32_t wp_flaws16(const uint32_t (&b)[4], uint32_t e20, uint32_t ev) {
  const uint32_t c20 = ~e20;  // bit 4 + j: byte j continues a token
  const uint32_t run2 = c20 & (c20 << 1);
  const uint32_t run4 = run2 & (run2 << 2);                     // bit i: bytes i - 3 .. i of the 20 continue
  const uint32_t longer = e20 & (run4 << 1);                    // an end behind four of them: a token of 5 bytes or more
  const uint32_t zero_end = (~wp_nonzero16(b) & 0xffffu) << 4;  // 0x00 bytes ...
  return ((longer | (zero_end & (c20 << 1))) >> 4) & ev;        // ... behind a byte that continues
}

// SPLIT launches (round 5, small batches: fewer chunks than the chip has room for): the pieces of a chunk are spread over
// gridDim.y workgroups and the two chains are replaced by three light kernels' results in global memory --
//   k_wp_counts   t0[piece] = token ends in front of the piece (the popcounts of chain 1, one block scan per chunk)
//   PASS 1        this kernel up to the pieces' aggregates: agg[piece] = {value of every lane behind the piece relative to
//                 its start, reset flags} (what chain 2 would combine)
//   k_wp_carry    carry[piece] = the values in front of the piece (a segmented scan of the aggregates, one wave per chunk)
//   PASS 2        this kernel from its first byte to its stores, T0 and the carry read instead of waited for.
// About 1.5 x the arithmetic of the chained launch, but no piece waits for another one: a lone cloud's 31 chunks are 5456
// pieces on 256 CUs instead of 31 chains of 176 hops.
struct WpSplit {
  uint32_t* t0;       // [chunk * maxp + piece]
  int32_t* agg;       // [(chunk * maxp + piece) * (NOPS + 1)]
  int32_t* carry;     // [(chunk * maxp + piece) * NOPS]
  uint32_t* flags;    // [chunk * 4]: irregular, palette index out of range, end of the regular stream, workgroups done
  uint32_t maxp;      // pieces a chunk may have (more: the chunk is left to the kernels behind)
};

template <int NOPS, int NF, int NW, int WPE = 8, int SM = 0, int PASS = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, 8))) void k_decode_points_w(
    const DevPlan plan, const uint8_t* __restrict__ streams, const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
    uint32_t* __restrict__ reg_end, uint8_t* __restrict__ sec_done, uint32_t uses_v5, uint32_t* __restrict__ status,
    const uint8_t* __restrict__ col0, const uint8_t* __restrict__ col1, const uint32_t* __restrict__ reg_end_pre,
    const uint8_t* __restrict__ sec_cols, uint32_t fill_zero, const DecColumns many, const WpSplit sp) {
  using L = WpLds<NOPS, NF, NW>;
  using G = WpGeom<NOPS>;
  constexpr int T = NW * 64;
  // NF > 2 (round 4): layouts with 3..8 integer channels -- their sections were decoded into dense columns in front of this
  // kernel (stage1_decode_sections_w.h); a point's fields are read from `many` when the point is stored
  constexpr bool MANY = NF > 2;
  constexpr uint32_t NFA = MANY ? 1 : (NF ? NF : 1);  // array extents (NF == 0: nothing is ever folded)
  constexpr uint32_t ROWS = G::kRows;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  unsigned long long* trec = reinterpret_cast<unsigned long long*>(smem + L::kTrecOff);
  unsigned long long* vrec = reinterpret_cast<unsigned long long*>(smem + L::kVrecOff);
  uint32_t* pal = reinterpret_cast<uint32_t*>(smem + L::kPalOff);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + L::kMiscOff);  // [0] irregular, [1] palette index out of range,
                                                                     // [2] end of the regular stream, [3] a wait gave up,
                                                                     // [40..) fp_setup
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CLDN_WP_PROF
  const unsigned long long wp_t0 = __builtin_readcyclecounter();
#endif
  const DecChunk dc = chunks[c];
  if (!dc.valid) {
    if (tid == 0) sec_done[c] = 0u;
    return;
  }
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  const uint32_t step = plan.point_step;
  uint8_t* base = out + (size_t)dc.first_point * step;
  const uint32_t target = n * NOPS;

  if (tid == 0) {
    misc[0] = 0u;
    misc[1] = 0u;
    misc[2] = 0xffffffffu;
    misc[3] = 0u;
    misc[40] = 0xffffffffu;  // fp_setup: payload offset behind the regular stream
    misc[42] = 0xffffffffu;  // fp_setup: entries of the Palette section found by its size
    misc[64] = 0u;           // fp_setup: sections folded in
  }
  for (uint32_t i = tid; i < kWpRing * (1u + NOPS) * 2u; i += T) reinterpret_cast<uint32_t*>(trec)[i] = 0u;  // tags: none
  __syncthreads();

  bool from_cols = false;
  // (PASS 1 of a SPLIT launch only sums: where the sections begin and what is folded into the stores is PASS 2's business)
  const uint32_t reg_size = PASS == 1 ? 0xffffffffu
                                      : fp_setup<NOPS, NF, T>(plan, src, src_size, n, uses_v5, c, reg_end_pre, sec_cols, misc, pal, &from_cols);
  const uint32_t n_fold = PASS == 1 ? 0u : misc[64];

  // folded Palette sections / columns: parameters in (uniform) registers
  uint32_t fs_off[NFA], fs_bpv[NFA], fs_count[NFA], fs_bits[NFA], fs_ioff[NFA];
#pragma unroll
  for (uint32_t a = 0; a < NFA; ++a) {
    const FpSection sct = reinterpret_cast<const FpSection*>(misc + 72)[a < n_fold ? a : 0u];
    fs_off[a] = sct.field_off;
    fs_bpv[a] = SM != 0 ? 2u : sct.bpv;  // (store modes: the one field is a 16-bit one)
    fs_count[a] = sct.count;
    fs_bits[a] = a < n_fold ? sct.bits : 0u;
    fs_ioff[a] = a < n_fold ? sct.index_off : 0u;
  }
  // which store forms the layout allows (uniform)
  // SM != 0: the launcher has checked the layout (stage1_launch_decode: three floats back to back at a 4-byte aligned offset,
  // at most one integer field, of 2 bytes at an even offset; SM == 2: 16-byte points x y z + the field at 12, an aligned cloud and
  // CLDN_HIP_FILL_ZERO) -- the forms below are then compile-time facts instead of seven uniform flags and their branches
  bool contig = ((step | plan.ops[0].offset) & 3u) == 0u;
#pragma unroll
  for (int o = 1; o < NOPS; ++o) contig = contig && plan.ops[o].offset == plan.ops[0].offset + 4u * (uint32_t)o;
  if (SM != 0) contig = true;
  // the first three floats back to back and 4-byte aligned, the rest of the lanes elsewhere. Only for the many-column merge
  // (Ouster-style points, same box: 1.24 -> 1.11 ms); PCL's padded PointXYZI alone is 3 % slower with the 12-byte store
  const bool lead3 = SM == 0 && MANY && NOPS > 3 && ((step | plan.ops[0].offset) & 3u) == 0u && plan.ops[0].offset != 0xffffffffu &&
                     plan.ops[1].offset == plan.ops[0].offset + 4u && plan.ops[2].offset == plan.ops[0].offset + 8u;
  bool packed = true;  // the floats back to back at any alignment, all of them stored
#pragma unroll
  for (int o = 0; o < NOPS; ++o) packed = packed && plan.ops[o].offset != 0xffffffffu && plan.ops[o].offset == plan.ops[0].offset + 4u * (uint32_t)o;
  if (SM != 0) packed = false;
  float res[NOPS];
  uint32_t foff[NOPS];
#pragma unroll
  for (int o = 0; o < NOPS; ++o) {
    res[o] = plan.ops[o].res_f;
    foff[o] = plan.ops[o].offset;
  }
  const bool one_u16 = SM != 0 ? (n_fold == 1u) : (NOPS == 3 && contig && n_fold == 1u && fs_bpv[0] == 2u && ((fs_off[0] | step) & 1u) == 0u);  // XYZ + one 16-bit field
  // fill_zero (CLDN_HIP_FILL_ZERO: the bytes no field covers may be written as 0): the two common padded layouts leave
  // as whole 16-byte stores
  const bool full16 = SM == 2 ? one_u16 : (SM == 1 ? false : (fill_zero != 0u && one_u16 && step == 16u && foff[0] == 0u && fs_off[0] == 12u && ((uintptr_t)base & 15u) == 0u));
  const bool full32 = SM == 0 && fill_zero != 0u && NOPS == 3 && contig && n_fold == 1u && fs_bpv[0] == 4u && step == 32u && foff[0] == 0u &&
                      fs_off[0] == 16u && ((uintptr_t)base & 15u) == 0u;

  // ---------------------------------------------------------------------------------------------------------
  // pieces. v = payload offset + a0 (a0 = misalignment of the payload): piece p owns v in [p, p + 1) * 1008
  // ---------------------------------------------------------------------------------------------------------
  const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
  const uint8_t* src_al = src - a0;
  const uint32_t vend = a0 + src_size;
  const uint32_t n_pieces = src_size ? (vend + kWpPiece - 1u) / kWpPiece : 0u;
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  const uint32_t vals_lds = smem_lds + L::kValsOff + wave * (kWpSlots * 4u);  // LDS byte address of the wave's token slots
  const uint32_t* vals = reinterpret_cast<const uint32_t*>(smem + L::kValsOff + wave * (kWpSlots * 4u));
  const uint32_t dummy_lds = smem_lds + L::kDummyOff + lane * 4u;

  // an aligned unit that holds at least one payload byte lies in a mapped page; anything else is not touched (0xff:
  // bytes that continue a token and end none). Lane l holds unit 62 p + l - 1 (piece 0, lane 0: v0 wraps -- not touched)
  auto load_unit = [&](uint32_t v0, uint32_t(&u)[4]) __attribute__((always_inline)) {
    const bool ok = v0 < vend;
    const uint4 w = *reinterpret_cast<const uint4*>(src_al + (ok ? v0 : 0u));
    u[0] = ok ? w.x : 0xffffffffu;
    u[1] = ok ? w.y : 0xffffffffu;
    u[2] = ok ? w.z : 0xffffffffu;
    u[3] = ok ? w.w : 0xffffffffu;
  };

  uint32_t b[4];   // my unit
  constexpr bool SPLIT = PASS != 0;
  const uint32_t p_step = SPLIT ? gridDim.y * (uint32_t)NW : (uint32_t)NW;
  uint32_t p = SPLIT ? blockIdx.y * (uint32_t)NW + wave : wave;
  if (SPLIT && n_pieces > sp.maxp) {  // (uniform; a payload beyond what the plan allows: the serial decoder raises the errors)
    if (PASS == 2 && blockIdx.y == 0u && tid == 0) {
      reg_end[c] = kDecRedo;
      sec_done[c] = 0u;
    }
    return;
  }
  const size_t sp_base = SPLIT ? (size_t)c * sp.maxp : 0u;
  if (n_pieces) load_unit(min(p, n_pieces) * kWpPiece + lane * 16u - 16u, b);  // (no payload: nothing may be read)
  __builtin_amdgcn_s_waitcnt(0);  // (everything asked for so far is waited for HERE: a wait at the loop's top would also wait for every piece's stores)
  bool gave_up = false;
#ifdef CLDN_WP_PROF
  unsigned long long wp_acc[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long wp_tl = __builtin_readcyclecounter();
  const unsigned long long wp_tstart = wp_tl;
  uint32_t wp_np = 0;
#define WP_T(i) { const unsigned long long t_ = __builtin_readcyclecounter(); wp_acc[i] += t_ - wp_tl; wp_tl = t_; }
#else
#define WP_T(i)
#endif
  __builtin_amdgcn_s_setprio(1);  // (waves that poll a record step down to 0)
  for (; p < n_pieces; p += p_step) {
    // ---- token ends of my unit; the ends that are tokens of the payload (ev); their numbers
    const uint32_t v0 = p * kWpPiece + lane * 16u - 16u;
    if (p == 0u) {  // uniform. What lies in front of the payload: token ends, value bits 0
      if (lane == 0u) b[0] = b[1] = b[2] = b[3] = 0u;
      if (lane == 1u) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) b[k] = a0 >= 4u * k + 4u ? 0u : (a0 > 4u * k ? b[k] & (0xffffffffu << (8u * (a0 - 4u * k))) : b[k]);
      }
    }
    const uint32_t eraw = wp_ends16(b);
    uint32_t ev = lane != 0u ? eraw : 0u;  // (lane 0's tokens are the piece's in front)
    if (p == 0u || p * kWpPiece + 1008u > vend) {  // uniform: the payload's first / last piece
      const uint32_t lo = p == 0u && lane == 1u ? a0 : 0u;
      const uint32_t hi = lane != 0u && vend > v0 ? min(vend - v0, 16u) : 0u;
      ev &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    const uint32_t evc = lane < 63u ? ev : 0u;  // (the halo's tokens are the next piece's: written, not counted or checked)
    const uint32_t cl = (uint32_t)__builtin_popcount(evc);
    const uint32_t incl = wave_inclusive_scan(cl);
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t tb = incl - cl;  // (lane 63: cnt -- the halo's tokens follow the piece's)
    WP_T(0)
    // ---- phase V: every token's value -> its slot
    const uint32_t e20 = (wp_from_lane_below(eraw) >> 12) | (eraw << 4);
    uint32_t flaws = wp_flaws16(b, e20, evc);  // bit j: byte j of my unit gives the chunk back (see the head of the file)
    {
      uint32_t pk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pk[k] = wp_pack7(b[k]);
      load_unit(min(p + p_step, n_pieces) * kWpPiece + lane * 16u - 16u, b);  // my next piece's bytes are requested now (b is free)
      wp_scatter16(pk, e20, ev, vals_lds + tb * 4u, dummy_lds);
    }
    WP_T(1)
    // ---- chain 1: token ends in front of the piece
    uint32_t T0 = 0u;
    if constexpr (SPLIT) {
      T0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sp.t0[sp_base + p]);
    } else if (p != 0u) {
      // (the hop from the record's arrival to this piece's own record is what every piece behind waits for: it runs at the
      // highest priority -- among eight waves of equal priority on a SIMD its few instructions took 2 k cycles)
      WP_BOOST(3);
      const unsigned long long* r = trec + ((p - 1u) & (kWpRing - 1u));
      unsigned long long x = wp_rec_load(r);
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) != p) {  // not there yet: poll at low priority
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t spins = 1u;; ++spins) {
          __builtin_amdgcn_s_sleep(kWpSleep);
          x = wp_rec_load(r);
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) == p) break;
          if ((spins & 63u) == 0u && (spins >= kWpSpinLimit || __hip_atomic_load(&misc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u)) {
            gave_up = true;
            break;
          }
        }
        WP_BOOST_BACK();
      }
      T0 = (uint32_t)x;
      T0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)T0);
    }
    if (gave_up) break;  // uniform
    if (!SPLIT && lane == 0u) wp_rec_store(trec + (p & (kWpRing - 1u)), ((unsigned long long)(p + 1u) << 32) | (T0 + cnt));
    if (!SPLIT) WP_BOOST(1);
    if (T0 >= target) break;  // the regular stream ended in front of this piece (sections; uniform)
    if (T0 + cnt >= target) {  // uniform: the end that closes the last point lies in this piece
      const uint32_t want = target - T0;  // its 1-based rank among the piece's ends
      uint32_t keep = tb + cl < want ? 0xffffu : 0u;  // my unit's bytes up to the regular stream's last one
      if (tb < want && want <= tb + cl) {
        uint32_t m = evc;
        for (uint32_t k = tb + 1u; k < want; ++k) m &= m - 1u;
        const uint32_t last = (uint32_t)__builtin_ctz(m);
        misc[2] = v0 + last + 1u - a0;
        keep = (2u << last) - 1u;
      }
      flaws &= keep;  // (what lies behind it are sections, not tokens)
    }
    WP_T(2)
    // ---- the points this piece owns: those whose first token ends in it. Point j: slots k0 + NOPS * j + o
    const uint32_t q_first = (T0 + (uint32_t)NOPS - 1u) / (uint32_t)NOPS;  // <= n (T0 < target)
    const uint32_t k0 = q_first * (uint32_t)NOPS - T0;
    const uint32_t npts = cnt > k0 ? min((cnt - k0 + (uint32_t)NOPS - 1u) / (uint32_t)NOPS, n - q_first) : 0u;
    wp_wave_sync();
    // ---- phase B, first half: tokens -> values relative to the piece's start. bs = running value behind the rows so far
    int32_t val[ROWS][NOPS];
    uint32_t flagsv = 0u, marksv = 0u;  // bit 4 r + o: value r, o does not take the carry / is a NaN
    int32_t bs[NOPS];
    uint32_t bs_fl = 0u;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) bs[o] = 0;
    const bool irregular = flaws != 0u;
#pragma unroll
    for (uint32_t r = 0; r < ROWS; ++r) {
#pragma unroll
      for (int o = 0; o < NOPS; ++o) val[r][o] = 0;
      if (r * 64u < npts) {  // uniform
        const uint32_t j = r * 64u + lane;
        const bool have = j < npts;
        const uint32_t s0 = have ? k0 + (uint32_t)NOPS * j : 0u;
        int32_t dlt[NOPS];
        bool zero = false;
#pragma unroll
        for (int o = 0; o < NOPS; ++o) {
          const uint32_t u = vals[s0 + (uint32_t)o];
          zero = zero || u == 0u;
          const uint32_t u1 = u - 1u;
          dlt[o] = (int32_t)((u1 >> 1) ^ (0u - (u1 & 1u)));  // u == 0 (the NaN marker) -> 0x80000000, which no token of <= 4 bytes gives
        }
        if (__ballot(have && zero) == 0ull) {  // no marker in this row (the rule for lidar data): DPP prefix sums
#pragma unroll
          for (int o = 0; o < NOPS; ++o) {
            const uint32_t inc = wave_inclusive_scan(have ? (uint32_t)dlt[o] : 0u);
            val[r][o] = (int32_t)((uint32_t)bs[o] + inc);
            bs[o] = (int32_t)((uint32_t)bs[o] + (uint32_t)__builtin_amdgcn_readlane((int)inc, 63));
          }
          flagsv |= bs_fl << (4u * r);
        } else {
          int32_t inc[NOPS];
          uint32_t mk = 0u;
#pragma unroll
          for (int o = 0; o < NOPS; ++o) {
            const bool m = have && dlt[o] == (int32_t)0x80000000;
            mk |= m ? (1u << o) : 0u;
            inc[o] = (m || !have) ? 0 : dlt[o];
          }
          uint32_t fin = mk;
          wp_seg_scan<NOPS>(inc, fin);
          const uint32_t f63 = (uint32_t)__builtin_amdgcn_readlane((int)fin, 63);
#pragma unroll
          for (int o = 0; o < NOPS; ++o) {
            val[r][o] = (fin & (1u << o)) ? inc[o] : (int32_t)((uint32_t)bs[o] + (uint32_t)inc[o]);
            const int32_t l63 = __builtin_amdgcn_readlane(inc[o], 63);
            bs[o] = (f63 & (1u << o)) ? l63 : (int32_t)((uint32_t)bs[o] + (uint32_t)l63);
          }
          flagsv |= (fin | bs_fl) << (4u * r);
          marksv |= mk << (4u * r);
          bs_fl |= f63;
        }
      }
    }
    if (__ballot(irregular) != 0ull && lane == 0u) misc[0] = 1u;
    if constexpr (PASS == 1) {  // the piece's aggregate leaves; k_wp_carry combines them
      if (lane <= (uint32_t)NOPS) {
        int32_t v = (int32_t)bs_fl;
#pragma unroll
        for (int o = 0; o < NOPS; ++o) v = lane == (uint32_t)o ? bs[o] : v;
        sp.agg[(sp_base + p) * (size_t)(NOPS + 1) + lane] = v;
      }
      wp_wave_sync();  // the next piece's bytes and list overwrite this one's
      continue;
    }
    // ---- the points' integer fields are requested now (a column value, or the dword that holds a folded Palette's
    // index), rows and fields side by side: they arrive while the wave waits for its carry and converts
    uint32_t raw[NFA][ROWS];
#pragma unroll
    for (uint32_t a = 0; a < NFA; ++a)
#pragma unroll
      for (uint32_t r = 0; r < ROWS; ++r) raw[a][r] = 0u;
    if (!MANY && n_fold != 0u) {  // uniform
#pragma unroll
      for (uint32_t r = 0; r < ROWS; ++r) {
        if (r * 64u < npts) {  // uniform
          const uint32_t j = r * 64u + lane;
          const uint32_t q = q_first + (j < npts ? j : 0u);
#pragma unroll
          for (uint32_t a = 0; a < NFA; ++a) {
            if (a >= n_fold) break;  // uniform
            if (from_cols) {
              const uint8_t* colp = (a == 0u ? col0 : col1) + (size_t)dc.first_point * fs_bpv[a];
              raw[a][r] = fs_bpv[a] == 2u ? (uint32_t)reinterpret_cast<const uint16_t*>(colp)[q] : reinterpret_cast<const uint32_t*>(colp)[q];
            } else if (fs_bits[a] != 0u) {  // (bits != 0: at least two table entries, the payload has 4 bytes)
              const uint32_t o = fs_ioff[a] + (__umul24(q, fs_bits[a]) >> 3);  // payload offset of the index's first byte
              const uint32_t oc = min(o, src_size - 4u);                       // (the last indexes: the dword that ends with the payload)
              uint32_t w;
              __builtin_memcpy(&w, src + oc, 4);
              raw[a][r] = w;
            }
          }
        }
      }
    }
    WP_T(3)
    // ---- chain 2: the values in front of the piece
    int32_t carry[NOPS];
#pragma unroll
    for (int o = 0; o < NOPS; ++o) carry[o] = 0;
    if constexpr (SPLIT) {
#pragma unroll
      for (int o = 0; o < NOPS; ++o) carry[o] = __builtin_amdgcn_readfirstlane(sp.carry[(sp_base + p) * (size_t)NOPS + (size_t)o]);
    } else if (p != 0u) {
      WP_BOOST(3);
      const unsigned long long* r = vrec + (size_t)((p - 1u) & (kWpRing - 1u)) * NOPS + min(lane, (uint32_t)NOPS - 1u);
      unsigned long long x = wp_rec_load(r);
      if (__ballot((uint32_t)(x >> 32) != p) != 0ull) {  // not there yet: poll at low priority
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t spins = 1u;; ++spins) {
          __builtin_amdgcn_s_sleep(kWpSleep);
          x = wp_rec_load(r);
          if (__ballot((uint32_t)(x >> 32) != p) == 0ull) break;
          if ((spins & 63u) == 0u && (spins >= kWpSpinLimit || __hip_atomic_load(&misc[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u)) {
            gave_up = true;
            break;
          }
        }
        WP_BOOST_BACK();
      }
#pragma unroll
      for (int o = 0; o < NOPS; ++o) carry[o] = __builtin_amdgcn_readlane((int)(uint32_t)x, o);
    }
    if (gave_up) break;  // uniform
    if constexpr (!SPLIT) {
      uint32_t mine = 0u;
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        const uint32_t pre = (bs_fl & (1u << o)) ? (uint32_t)bs[o] : (uint32_t)carry[o] + (uint32_t)bs[o];
        mine = lane == (uint32_t)o ? pre : mine;
      }
#ifdef CLDN_WP_HOPDELAY  // (experiment: how much of the kernel is the chain? every hop of chain 2 made CLDN_WP_HOPDELAY x 64 cycles longer)
      __builtin_amdgcn_s_sleep(CLDN_WP_HOPDELAY);
#endif
      if (lane < (uint32_t)NOPS)
        wp_rec_store(vrec + (size_t)(p & (kWpRing - 1u)) * NOPS + lane, ((unsigned long long)(p + 1u) << 32) | mine);
      WP_BOOST(1);
    }
    // The bytes of my next piece were requested long ago: naming them here puts the wait for them IN FRONT of this piece's
    // stores. Left to the top of the loop it becomes s_waitcnt vmcnt(0) -- the compiler cannot count the stores of a variable
    // number of rows -- and every piece would begin by waiting for the write acknowledgements of the piece before.
    // The same for the dwords that hold the points' Palette indexes / column values: ONE wait here instead of an
    // s_waitcnt vmcnt(0) per row behind the row before's stores (the waitcnt pass loses count across the rows' branches).
    asm volatile("" ::"v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
#pragma unroll
    for (uint32_t a = 0; a < NFA; ++a)
#pragma unroll
      for (uint32_t r = 0; r < ROWS; ++r) asm volatile("" : "+v"(raw[a][r]));
    WP_T(4)
    // ---- phase B, second half: values -> floats -> the points. A lane stores its own point.
    const bool piece_flags = bs_fl != 0u;  // uniform: a marker somewhere in the piece
#pragma unroll
    for (uint32_t r = 0; r < ROWS; ++r) {
      if (r * 64u < npts) {  // uniform
        const uint32_t j = r * 64u + lane;
        const bool have = j < npts;
        const uint32_t q = q_first + (have ? j : 0u);  // point of the chunk (lanes without one: a valid address, unused)
        float f[NOPS];
        if (!piece_flags) {
#pragma unroll
          for (int o = 0; o < NOPS; ++o) f[o] = __fmul_rn((float)(int32_t)((uint32_t)carry[o] + (uint32_t)val[r][o]), res[o]);
        } else {
#pragma unroll
          for (int o = 0; o < NOPS; ++o) {
            const uint32_t bit = 1u << (4u * r + (uint32_t)o);
            const int32_t iv = (flagsv & bit) ? val[r][o] : (int32_t)((uint32_t)carry[o] + (uint32_t)val[r][o]);
            f[o] = (marksv & bit) ? __uint_as_float(0x7fc00000u) : __fmul_rn((float)iv, res[o]);
          }
        }
        uint32_t pv[NFA];
#pragma unroll
        for (uint32_t a = 0; a < NFA; ++a) pv[a] = raw[a][r];
        if (!MANY && !from_cols && n_fold != 0u) {
#pragma unroll
          for (uint32_t a = 0; a < NFA; ++a) {
            if (a >= n_fold) break;  // uniform
            const uint32_t bits = fs_bits[a];
            uint32_t idx = 0u;
            if (bits != 0u) {  // uniform
              const uint32_t bit0 = __umul24(q, bits);              // < 32768 * 10
              const uint32_t o = fs_ioff[a] + (bit0 >> 3);
              const uint32_t oc = min(o, src_size - 4u);
              idx = (raw[a][r] >> ((bit0 & 7u) + 8u * (o - oc))) & ((1u << bits) - 1u);
            }
            if (have && idx >= fs_count[a]) misc[1] = 1u;  // index beyond the palette: the serial decoder redoes the sections and raises the error
            pv[a] = pal[a * kFastPalEntries + (idx & (kFastPalEntries - 1u))];
          }
        }
#ifdef CLDN_WP_ABL
        if (have && (!(CLDN_WP_ABL & 1) || f[0] == 1234567.0f)) {  // (ablation 1: nothing is stored)
#else
        if (have) {
#endif
          uint8_t* pt = base + __umul24(q, step);  // (q < 32768, step <= 1024)
          if (full16) {
            *reinterpret_cast<float4*>(pt) = make_float4(f[0], f[1], f[2], __uint_as_float(pv[0] & 0xffffu));
          } else if (full32) {
            reinterpret_cast<float4*>(pt)[0] = make_float4(f[0], f[1], f[2], 0.0f);
            reinterpret_cast<float4*>(pt)[1] = make_float4(__uint_as_float(pv[0]), 0.0f, 0.0f, 0.0f);
          } else {
            if (contig) {
              FloatVec<NOPS> v;
#pragma unroll
              for (int o = 0; o < NOPS; ++o) v.v[o] = f[o];
              *reinterpret_cast<FloatVec<NOPS>*>(pt + foff[0]) = v;
            } else if (packed) {  // the floats lie back to back at an odd address (18-byte points): ONE unaligned store
              FloatVec<NOPS> v;
#pragma unroll
              for (int o = 0; o < NOPS; ++o) v.v[o] = f[o];
              __builtin_memcpy(pt + foff[0], &v, NOPS * 4);
            } else if (lead3) {  // x, y, z back to back, the fourth float elsewhere (PCL's padded PointXYZI, Ouster-style points): 12 + 4 bytes
              FloatVec<3> v;
              v.v[0] = f[0];
              v.v[1] = f[1];
              v.v[2] = f[2];
              *reinterpret_cast<FloatVec<3>*>(pt + foff[0]) = v;
#pragma unroll
              for (int o = 3; o < NOPS; ++o)
                if (foff[o] != 0xffffffffu) __builtin_memcpy(pt + foff[o], &f[o], 4);
            } else {
#pragma unroll
              for (int o = 0; o < NOPS; ++o)
                if (foff[o] != 0xffffffffu) __builtin_memcpy(pt + foff[o], &f[o], 4);
            }
            if (MANY) {
              // all of the point's column values are requested before the first one is stored (one round trip, not one per
              // field); uniform guards, the plan and the column table are read with scalar loads
              uint32_t cv[8];
#pragma unroll
              for (uint32_t a = 0; a < 8u; ++a) {
                cv[a] = 0u;
                if (a < n_fold) {
                  const uint32_t f_bpv = plan.adaptive[a].bpv;
                  const uint8_t* colp = many.p[a] + (size_t)dc.first_point * f_bpv;
                  cv[a] = f_bpv == 2u ? (uint32_t)reinterpret_cast<const uint16_t*>(colp)[q] : reinterpret_cast<const uint32_t*>(colp)[q];
                }
              }
              // (two 16-bit fields side by side in one aligned dword leave as ONE store: uniform test)
              bool paired = false;
#pragma unroll
              for (uint32_t a = 0; a < 8u; ++a) {
                if (a < n_fold && !paired) {
                  const uint32_t f_off = plan.adaptive[a].offset;
                  if (plan.adaptive[a].bpv == 2u) {
                    const bool pair = a + 1u < n_fold && a + 1u < 8u && plan.adaptive[a + 1u < 8u ? a + 1u : a].bpv == 2u &&
                                      plan.adaptive[a + 1u < 8u ? a + 1u : a].offset == f_off + 2u && ((f_off | step) & 3u) == 0u;
                    if (pair) {
                      const uint32_t w = (cv[a] & 0xffffu) | (cv[a + 1u < 8u ? a + 1u : a] << 16);
                      __builtin_memcpy(pt + f_off, &w, 4);
                      paired = true;
                    } else {
                      const uint16_t h = (uint16_t)cv[a];
                      __builtin_memcpy(pt + f_off, &h, 2);
                    }
                  } else {
                    __builtin_memcpy(pt + f_off, &cv[a], 4);
                  }
                } else {
                  paired = false;
                }
              }
            } else if (one_u16) {
              *reinterpret_cast<uint16_t*>(pt + fs_off[0]) = (uint16_t)pv[0];
            } else if (SM == 0) {  // (store modes: at most the one 16-bit field above -- the launcher has checked the layout)
#pragma unroll
              for (uint32_t a = 0; a < NFA; ++a) {
                if (a >= n_fold) break;  // uniform
                if (fs_bpv[a] == 2u) {
                  const uint16_t h = (uint16_t)pv[a];
                  __builtin_memcpy(pt + fs_off[a], &h, 2);
                } else if (fs_bpv[a] == 4u) {
                  __builtin_memcpy(pt + fs_off[a], &pv[a], 4);
                } else {
                  st_raw(pt + fs_off[a], pv[a], fs_bpv[a]);
                }
              }
            }
          }
        }
      }
    }
    wp_wave_sync();  // the next piece's bytes and list overwrite this one's
    WP_T(5)
#ifdef CLDN_WP_PROF
    ++wp_np;
#endif
  }
#ifdef CLDN_WP_PROF
  if (PASS == 0 && lane == 0u && (c == 0u || c == 700u) && (wave == 0u || wave == 7u || wave == 15u)) {
    const unsigned long long t_ = __builtin_readcyclecounter();
    printf("chunk %u wave %u pieces %u: start %llu total %llu prologue %llu | counts %llu V %llu chain1 %llu B1 %llu chain2 %llu B2 %llu\n", c, wave, wp_np,
           wp_t0, t_ - wp_t0, wp_tstart - wp_t0, wp_acc[0], wp_acc[1], wp_acc[2], wp_acc[3], wp_acc[4], wp_acc[5]);
  }
#endif
  if (gave_up && lane == 0u) {
    misc[3] = 1u;
    misc[0] = 1u;
  }
  __syncthreads();
  if constexpr (PASS == 1) return;  // (what PASS 1 found irregular PASS 2 finds again)
  if constexpr (PASS == 2) {
    // the chunk's verdict is given by the last of its workgroups to get here (flags in global memory: k_wp_counts reset them)
    if (tid == 0) {
      uint32_t* fl = sp.flags + (size_t)c * 4u;
      if (misc[0] != 0u) atomicOr(fl + 0, 1u);
      if (misc[1] != 0u) atomicOr(fl + 1, 1u);
      if (misc[2] != 0xffffffffu) atomicMin(fl + 2, misc[2]);
      __threadfence();
      const uint32_t done = atomicAdd(fl + 3, 1u);
      if (done + 1u == gridDim.y) {
        __threadfence();
        misc[0] = __hip_atomic_load(fl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        misc[1] = __hip_atomic_load(fl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        misc[2] = __hip_atomic_load(fl + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        misc[3] = 2u;  // not the last one
      }
    }
    if (tid != 0 || misc[3] == 2u) return;
  }
  if (tid == 0) {
    const uint32_t pos = misc[2];
    const bool redo = misc[0] != 0u || pos == 0xffffffffu;
    reg_end[c] = redo ? kDecRedo : pos;
    const bool folded = !redo && n_fold != 0u && misc[1] == 0u && pos == reg_size;
    sec_done[c] = folded ? 2u : 0u;
    if (!redo) atomicAdd(&status[kStatFastRegular], 1u);
    if (folded) atomicAdd(&status[kStatFastSections], 1u);
    if (folded && !from_cols) atomicAdd(&status[kStatFoldedByGuess], 1u);
  }
}

// ---- the two light kernels of a SPLIT launch ----------------------------------------------------------------------------
// grid = n_chunks, 1024 threads: the token ends of every piece of the chunk (the same masks as k_decode_points_w's), one
// exclusive scan -> t0; the chunk's flags reset, its aggregates zeroed (pieces behind the regular stream write none)
__global__ __launch_bounds__(1024) void k_wp_counts(const uint8_t* __restrict__ streams, const DecChunk* __restrict__ chunks,
                                                   const WpSplit sp, uint32_t agg_stride) {
  __shared__ uint32_t scan[40];
  __shared__ uint32_t carry_sh;
  const uint32_t c = blockIdx.x, tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DecChunk dc = chunks[c];
  if (tid < 4u) sp.flags[(size_t)c * 4u + tid] = tid == 2u ? 0xffffffffu : 0u;
  if (!dc.valid) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
  const uint8_t* src_al = src - a0;
  const uint32_t vend = a0 + src_size;
  const uint32_t n_pieces = src_size ? (vend + kWpPiece - 1u) / kWpPiece : 0u;
  if (n_pieces > sp.maxp) return;
  uint32_t* t0 = sp.t0 + (size_t)c * sp.maxp;
  for (uint32_t p = wave; p < n_pieces; p += 16u) {
    const uint32_t v0 = p * kWpPiece + lane * 16u;
    const bool ok = v0 < vend && lane < kWpUnits;  // (here lane l < 62 counts unit l of the piece)
    const uint4 w = *reinterpret_cast<const uint4*>(src_al + (ok ? v0 : 0u));
    const uint32_t b[4] = {ok ? w.x : 0xffffffffu, ok ? w.y : 0xffffffffu, ok ? w.z : 0xffffffffu, ok ? w.w : 0xffffffffu};
    uint32_t ends = wp_ends16(b);
    if (p * kWpPiece < a0 || (p + 1u) * kWpPiece > vend) {  // uniform: the payload's first / last piece
      const uint32_t lo = a0 > v0 ? min(a0 - v0, 16u) : 0u;
      const uint32_t hi = vend > v0 ? min(vend - v0, 16u) : 0u;
      ends &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    const uint32_t cnt = wave_sum((uint32_t)__builtin_popcount(ends));
    if (lane == 0u) t0[p] = cnt;
  }
  for (size_t i = tid; i < (size_t)n_pieces * agg_stride; i += 1024u) sp.agg[(size_t)c * sp.maxp * agg_stride + i] = 0;
  if (tid == 0) carry_sh = 0u;
  __threadfence_block();
  __syncthreads();
  for (uint32_t base = 0; base < n_pieces; base += 1024u) {
    const uint32_t p = base + tid;
    const uint32_t mine = p < n_pieces ? t0[p] : 0u;
    uint32_t total;
    const uint32_t excl = block_exclusive_scan<1024>(mine, scan, &total);  // (barriers inside)
    const uint32_t before = carry_sh;
    __syncthreads();
    if (p < n_pieces) t0[p] = before + excl;
    if (tid == 0) carry_sh = before + total;
    __syncthreads();
  }
}

// grid = n_chunks, one wave: carry[piece] = what chain 2 hands a piece = the segmented running combination of the aggregates
// of the pieces in front of it (a set reset flag: the aggregate replaces the running value)
template <int NOPS>
__global__ __launch_bounds__(64) void k_wp_carry(const uint8_t* __restrict__ streams, const DecChunk* __restrict__ chunks, const WpSplit sp) {
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  const DecChunk dc = chunks[c];
  if (!dc.valid) return;
  const uint32_t a0 = (uint32_t)((uintptr_t)(streams + dc.src_off) & 15u);
  const uint32_t vend = a0 + dc.src_size;
  const uint32_t n_pieces = dc.src_size ? (vend + kWpPiece - 1u) / kWpPiece : 0u;
  if (n_pieces > sp.maxp) return;
  const int32_t* agg = sp.agg + (size_t)c * sp.maxp * (NOPS + 1);
  int32_t* carry = sp.carry + (size_t)c * sp.maxp * NOPS;
  int32_t run[NOPS];
#pragma unroll
  for (int o = 0; o < NOPS; ++o) run[o] = 0;
  for (uint32_t base = 0; base < n_pieces; base += 64u) {
    const uint32_t p = base + lane;
    int32_t inc[NOPS];
    uint32_t fin = 0u;
#pragma unroll
    for (int o = 0; o < NOPS; ++o) inc[o] = p < n_pieces ? agg[(size_t)p * (NOPS + 1) + o] : 0;
    if (p < n_pieces) fin = (uint32_t)agg[(size_t)p * (NOPS + 1) + NOPS];
    wp_seg_scan<NOPS>(inc, fin);
    int32_t pub[NOPS];  // the values behind piece p
#pragma unroll
    for (int o = 0; o < NOPS; ++o) pub[o] = (fin & (1u << o)) ? inc[o] : (int32_t)((uint32_t)run[o] + (uint32_t)inc[o]);
#pragma unroll
    for (int o = 0; o < NOPS; ++o) {
      const int32_t prev = __shfl_up(pub[o], 1);
      if (p < n_pieces) carry[(size_t)p * NOPS + o] = lane == 0u ? run[o] : prev;
    }
#pragma unroll
    for (int o = 0; o < NOPS; ++o) run[o] = __builtin_amdgcn_readlane(pub[o], 63);
  }
}

}  // namespace cldn
