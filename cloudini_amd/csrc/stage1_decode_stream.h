// stage1_decode_stream.h -- k_decode_stream_w: the barrier-free decoder of GENERAL regular streams (round 4): any list of
// up to 8 per-point encoders -- FloatN lanes, FieldDecoderFloat_Lossy<float/double>, FieldDecoderInt<T> (varint tokens
// of up to 10 bytes, include/cloudini_lib/field_decoder.hpp:79-106,330-353, encoding_utils.hpp:98-148), and, when the
// token ends come from k_mark_token_ends' bitmap, raw FieldDecoderCopy fields and FieldDecoderFloat_XOR
// (field_decoder.hpp:56-75,108-156). It takes the streams k_decode_points_w does not (that kernel keeps the FloatN-only
// layouts of the BASELINE configs, with 32-bit arithmetic and a point's values held in registers).
//
// Same decomposition as stage1_decode_wave.h: pieces of 1 KiB at 16-byte aligned addresses, one wave per piece, the waves
// of a chunk's workgroup take the pieces round robin; a piece owns the points that begin behind the token ends it holds;
// two chains of tagged LDS records hand over the token count in front of a piece and the running values.
// What differs: a point's tokens are walked op by op (uniform loop over the plan, token lengths from the piece's end
// bits, values as int64), and the piece is decoded TWICE from its LDS copy -- a first walk only adds up every op's
// differences (the piece's aggregate, published at once), the second walk, after the carry of the pieces in front has
// arrived, scans, converts and stores. Nothing per point is kept in registers between the two, so a row of 64 points is
// a loop iteration, whatever the number of ops. Per-op state (running value, aggregate) lives in LANE o of a register.
#pragma once

namespace cldn {

constexpr uint32_t kSwPiece = 1024u;
constexpr uint32_t kSwHalo = 96u;           // a point has at most kSwMaxPointBytes bytes behind its first
constexpr uint32_t kSwMaxPointBytes = 88u;
constexpr uint32_t kSwMaxOps = 8u;
constexpr uint32_t kSwRing = 32u;
constexpr uint32_t kSwListEntries = 1032u;  // points a piece can own (one op of one byte: one per byte) + slack
constexpr uint32_t kSwSpinLimit = 1u << 18;
constexpr uint32_t kSwMaxRounds = 40u;      // MODE 2: window changes inside one piece (a chunk's first stamps settle through several)

template <int NW, int MODE = 0>
struct SwLds {
  static constexpr bool FORM = MODE != 0;
  static constexpr bool GOR = MODE == 2;
  static constexpr uint32_t kRing = GOR ? 16u : kSwRing;                      // chain records (a piece waits for the one in front only)
  static constexpr uint32_t kBytesOff = 0;                                    // per wave: [piece][halo]
  static constexpr uint32_t kEndsOff = kSwPiece + kSwHalo;                    // u16 [64 + 6] + pad: end bits of the units
  static constexpr uint32_t kListOff = kEndsOff + 160u;                       // u16 [kSwListEntries] (FORM: a point has 2 bytes at least)
  static constexpr uint32_t kListEntries = FORM ? kSwPiece / 2u + 8u : kSwListEntries;
  static constexpr uint32_t kJumpOff = (kListOff + kListEntries * 2u + 15u) & ~15u;  // FORM: u16 [kSwPiece]: where the point that starts at a byte ends
  // MODE 1: u16 [kSwMaxPointBytes][8]: a candidate's first point in every 128-byte block. MODE 2: u16 [kSwPiece]: where the
  // FOURTH point from a byte ends (round 6: candidates and the list follow the jumps four points at a time); a piece that is
  // redone in rounds keeps its rounds there instead
  static constexpr uint32_t kCheckOff = kJumpOff + (FORM ? kSwPiece * 2u : 0u);
  static constexpr uint32_t kWaveBytes = kCheckOff + (GOR ? kSwPiece * 2u : (FORM ? kSwMaxPointBytes * 16u : 0u));
  static_assert(kSwMaxRounds * kSwMaxOps * 4u + (kSwMaxRounds + 1u) * 2u <= kSwMaxPointBytes * 16u, "MODE 2 keeps its rounds where the fourth-point table is");
  static constexpr uint32_t kLutOff = (uint32_t)NW * kWaveBytes;              // u16 [kSwMaxOps][256] (MODE 0 only)
  static constexpr uint32_t kTrecOff = kLutOff + (FORM ? 0u : kSwMaxOps * 512u);  // u64 [kRing]
  static constexpr uint32_t kVrecOff = kTrecOff + kRing * 8u;                 // u64 [kRing][kSwMaxOps][2]: {tag, lo}, {tag, hi} of the value behind the piece
  static constexpr uint32_t kGrecOff = kVrecOff + kRing * kSwMaxOps * 16u;    // u64 [kRing][kSwMaxOps]: {tag, Gorilla window of op o behind the piece} (MODE 2)
  static constexpr uint32_t kMiscOff = kGrecOff + kRing * kSwMaxOps * 8u;
  static constexpr uint32_t kTotal = kMiscOff + 256u;
  static_assert(!GOR || 2u * kTotal <= 160u * 1024u, "two workgroups per CU");
};

#define SW_DPP64(X, CTRL, RMASK, BC)                                                                                  \
  ((((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)((X) >> 32), CTRL, RMASK, 0xf, BC)) << 32) |   \
   (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(X), CTRL, RMASK, 0xf, BC))

// inclusive prefix over the 64 lanes with + or ^ (XOR = true)
template <bool XOR>
__device__ __forceinline__ uint64_t sw_scan64(uint64_t x) {
#define SW_STEP(CTRL, RMASK, BC)                      \
  {                                                   \
    const uint64_t o = SW_DPP64(x, CTRL, RMASK, BC);  \
    x = XOR ? (x ^ o) : (x + o);                      \
  }
  SW_STEP(0x111, 0xf, true)
  SW_STEP(0x112, 0xf, true)
  SW_STEP(0x114, 0xf, true)
  SW_STEP(0x118, 0xf, true)
  SW_STEP(0x142, 0xa, false)
  SW_STEP(0x143, 0xc, false)
#undef SW_STEP
  return x;
}

// inclusive segmented prefix sum: lanes whose `f` is set restart the sum with their own value; f becomes the OR of the
// flags up to the lane
__device__ __forceinline__ void sw_seg_scan64(uint64_t& x, uint32_t& f) {
#define SW_STEP(CTRL, RMASK, BC)                                                                      \
  {                                                                                                   \
    const uint64_t o = SW_DPP64(x, CTRL, RMASK, BC);                                                  \
    const uint32_t of = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)f, CTRL, RMASK, 0xf, BC);      \
    x = f ? x : x + o;                                                                                \
    f |= of;                                                                                          \
  }
  SW_STEP(0x111, 0xf, true)
  SW_STEP(0x112, 0xf, true)
  SW_STEP(0x114, 0xf, true)
  SW_STEP(0x118, 0xf, true)
  SW_STEP(0x142, 0xa, false)
  SW_STEP(0x143, 0xc, false)
#undef SW_STEP
}

__device__ __forceinline__ uint64_t sw_lane64(uint64_t x, uint32_t l) {
  return (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), (int)l)) << 32) |
         (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, (int)l);
}

// One token of the stream copy in LDS: `len` bytes from byte `bp`. Varint tokens (raw == false): decodeVarint's value
// (zig-zag undone), *marker = the token is the single byte 0x00 (the NaN marker of the float decoders), *bad = what
// decodeVarint throws on (more than 10 bytes, bits beyond 64, a zero spread over several bytes). Raw tokens: the bytes.
__device__ __forceinline__ uint64_t sw_token(const uint32_t* wbuf, uint32_t bp, uint32_t len, bool raw, bool* marker, bool* bad) {
  const uint32_t di = bp >> 2, sh = (bp & 3u) * 8u;
  const uint32_t d0 = wbuf[di], d1 = wbuf[di + 1u], d2 = wbuf[di + 2u];
  const uint32_t lo = __builtin_amdgcn_alignbit(d1, d0, sh), hi = __builtin_amdgcn_alignbit(d2, d1, sh);
  uint64_t w = (((uint64_t)hi) << 32) | lo;  // the token's first 8 bytes
  if (len < 8u) w &= (1ull << (8u * len)) - 1ull;
  *marker = false;
  if (raw) return w;
  uint64_t x = w & 0x7f7f7f7f7f7f7f7full;
  x = (x & 0x007f007f007f007full) | ((x & 0x7f007f007f007f00ull) >> 1);
  x = (x & 0x00003fff00003fffull) | ((x & 0x3fff00003fff0000ull) >> 2);
  x = (x & 0x000000000fffffffull) | ((x & 0x0fffffff00000000ull) >> 4);
  if (len > 8u) {  // bytes 8 and 9: 7 more bits, then the one bit that is left of 64
    const uint32_t d3 = wbuf[di + 3u];
    const uint32_t top = __builtin_amdgcn_alignbit(d3, d2, sh);
    x |= ((uint64_t)(top & 0x7fu)) << 56;
    if (len > 9u) {
      const uint32_t b9 = (top >> 8) & 0x7fu;
      if (b9 > 1u || len > 10u) *bad = true;
      x |= ((uint64_t)b9) << 63;
    }
  }
  if (x == 0ull) {
    if (len == 1u) *marker = true;
    else *bad = true;
  }
  const uint64_t u1 = x - 1ull;
  return (u1 >> 1) ^ (0ull - (u1 & 1ull));
}

// The same for tokens of at most 4 bytes (the common case: the caller asks the whole wave): 32-bit arithmetic
__device__ __forceinline__ uint64_t sw_token32(const uint32_t* wbuf, uint32_t bp, uint32_t len, bool raw, bool* marker, bool* bad) {
  const uint32_t di = bp >> 2, sh = (bp & 3u) * 8u;
  uint32_t w = __builtin_amdgcn_alignbit(wbuf[di + 1u], wbuf[di], sh);
  if (len < 4u) w &= (1u << (8u * len)) - 1u;
  *marker = false;
  if (raw) return w;
  const uint32_t lo = w & 0x7f7f7f7fu;
  const uint32_t x1 = lo - ((lo & 0x7f007f00u) >> 1);   // 7-bit groups -> 14-bit groups
  const uint32_t x = x1 - __umul24(x1 >> 16, 49152u);   // -> one value (< 2^28)
  if (x == 0u) {
    if (len == 1u) *marker = true;
    else *bad = true;
  }
  const uint32_t u1 = x - 1u;
  return (uint64_t)(int64_t)(int32_t)((u1 >> 1) ^ (0u - (u1 & 1u)));  // (the marker's value is never used)
}

// One token of FieldDecoderFloat_Gorilla<double> (field_decoder.hpp:262-305; bits LSB-first, every token a whole number of
// bytes) at byte `bp` of the LDS copy -> the XOR difference against the value before (the chunk's first value: its raw
// bits), *len = the token's bytes. st = the window in effect (valid << 16 | leading << 8 | trailing). *bad: what the serial
// decoder refuses (a reuse token without a window, a window of more than 64 bits).
__device__ __forceinline__ uint64_t sw_gorilla(const uint32_t* wbuf, uint32_t bp, bool first, uint32_t st, uint32_t* len, bool* bad) {
  const uint32_t di = bp >> 2, sh = (bp & 3u) * 8u;
  const uint32_t d0 = wbuf[di], d1 = wbuf[di + 1u], d2 = wbuf[di + 2u], d3 = wbuf[di + 3u];
  const uint32_t W0 = __builtin_amdgcn_alignbit(d1, d0, sh), W1 = __builtin_amdgcn_alignbit(d2, d1, sh), W2 = __builtin_amdgcn_alignbit(d3, d2, sh);
  if (first) {
    *len = 8u;
    return (((uint64_t)W1) << 32) | W0;
  }
  if ((W0 & 1u) == 0u) {  // '0': the value before, again
    *len = 1u;
    return 0ull;
  }
  uint32_t k, m, tr;
  if ((W0 & 2u) == 0u) {  // '10': the window in effect
    const uint32_t ld = (st >> 8) & 0xffu;
    tr = st & 0xffu;
    if ((st >> 16) == 0u || ld + tr >= 64u) {
      *bad = true;
      *len = 1u;
      return 0ull;
    }
    m = 64u - ld - tr;
    k = 2u;
  } else {  // '11': 5 bits of leading zeros, 6 bits of (meaningful - 1)
    const uint32_t sl = (W0 >> 2) & 31u;
    m = ((W0 >> 7) & 63u) + 1u;
    if (sl + m > 64u) {
      *bad = true;
      *len = 2u;
      return 0ull;
    }
    tr = 64u - sl - m;
    k = 13u;
  }
  *len = (k + m + 7u) >> 3;
  const uint32_t lo = __builtin_amdgcn_alignbit(W1, W0, k), hi = __builtin_amdgcn_alignbit(W2, W1, k);
  uint64_t bits = (((uint64_t)hi) << 32) | lo;
  if (m < 64u) bits &= (1ull << m) - 1ull;
  return bits << tr;
}

// grid = n_chunks, NW * 64 threads, SwLds<NW>::kTotal bytes of LDS. token_ends == NULL: the token ends are the bytes with
// a clear MSB (plans made of varint tokens only); else the bitmap k_mark_token_ends laid out.
// reg_end[c] = where the regular stream ends, kDecRedo when the chunk is irregular (the 64-bit tile kernel / the serial
// decoder take it and raise the errors).
// FORM = true (token_ends == NULL): streams with raw fields between the varints (FieldDecoderCopy / XOR: a raw byte may look
// like anything, so the MSBs do not say where tokens end). What is known is the FORM of a point -- op after op, a varint
// or `size` raw bytes: every lane works out where the points that would start at its 16 bytes end (jump table), the first
// kSwMaxPointBytes lanes follow the jumps from their byte to the piece's end (where the next piece is entered, how many
// points start on the way), and chain 1 hands over {entry offset, points so far} instead of a token count: one lookup per
// piece; the owner of the entry then writes the piece's point starts by following the jumps once more. This replaces
// k_mark_token_ends' bitmap (one workgroup per chunk, tiles in sequence: 3.6 ms per 16 M points) for these layouts.
// MODE 2 (one FieldDecoderFloat_Gorilla<double> op in the point, field_decoder.hpp:158-302): the length of a window-reuse
// token is a state the tokens in front of it left ('11' tokens open a window of `meaningful` bits, '10' tokens reuse it), so
// the jump table can only be built once the state at the piece's entry is known: chain 1 is waited for FIRST (it carries
// {entry, points, window}), then the table is built for that window and one lane follows the jumps; a point whose '11'
// token changes the window ends the round -- the table is rebuilt for the new window from there. The token's value bits
// are XOR differences: the op joins the XOR-coded ones in both walks.
#define SW_PRIO_HOP() __builtin_amdgcn_s_setprio(3)  // (chain hops at the highest wave priority, as in k_decode_points_w)
#ifdef CLDN_SW_TRACE
#define SW_T(k) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tr_[k] += t_ - tl_; tl_ = t_; }
#else
#define SW_T(k)
#endif
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(MODE ? 6 : 8, 8))) void k_decode_stream_w(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                             const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                             uint32_t* __restrict__ reg_end, uint32_t* __restrict__ status,
                                                             const uint32_t* __restrict__ token_ends, uint32_t sect_chunks,
                                                             const DecColumns sect_cols, const uint8_t* __restrict__ col_flags,
                                                             const uint32_t* __restrict__ reg_end_pre, uint8_t* __restrict__ sec_done) {
  constexpr bool FORM = MODE != 0;
  constexpr bool GOR = MODE == 2;
  using L = SwLds<NW, MODE>;
  constexpr int T = NW * 64;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint16_t* lut = reinterpret_cast<uint16_t*>(smem + L::kLutOff);
  unsigned long long* trec = reinterpret_cast<unsigned long long*>(smem + L::kTrecOff);
  unsigned long long* vrec = reinterpret_cast<unsigned long long*>(smem + L::kVrecOff);
  unsigned long long* grec = reinterpret_cast<unsigned long long*>(smem + L::kGrecOff);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + L::kMiscOff);  // [0] irregular, [2] end of the regular stream, [3] a wait gave up
  // sect_chunks != 0 (MODE 0 only): SECTION MODE -- grid (chunks, fields); `chunks` holds the DeltaVarint sections
  // k_section_offsets sized (row = field), the section of field a is a stream of n tokens of the ONE integer op
  // plan.ops[a], and reg_end is the per-chunk counter of finished sections (stage1_decode_sections_w.h)
  const bool sect = MODE == 0 && sect_chunks != 0u;
  const uint32_t ob = sect ? blockIdx.y : 0u;  // first op of the plan this launch row uses
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DecChunk dc = chunks[sect ? (size_t)blockIdx.y * sect_chunks + c : (size_t)c];
  if (!sect && sec_done != nullptr && tid == 0) sec_done[c] = 0u;  // (set again at the end; nothing stale survives an early return)
  if (!dc.valid) return;
  if (sect && dc.valid != 1u) return;  // another mode: k_sections_w
  if (!sect && token_ends != nullptr && reg_end[c] == kDecRedo) return;  // k_mark_token_ends found the stream irregular
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  const uint32_t n_ops = sect ? 1u : plan.n_ops;
  // (section mode with out == NULL: the values go to the field's dense column instead of the points)
  const bool sect_to_cols = sect && out == nullptr;
  const uint32_t step = sect_to_cols ? (uint32_t)plan.ops[ob].size : plan.point_step;
  uint8_t* base = sect_to_cols ? const_cast<uint8_t*>(sect_cols.p[ob]) + (size_t)dc.first_point * step : out + (size_t)dc.first_point * step;
  const uint32_t target = n * n_ops;
  // COLUMN MERGE (regular mode, col_flags != NULL): the chunk's integer fields were decoded into dense columns in front of
  // this launch (sect_cols, flag per chunk) -- every point then leaves complete, no second pass over the cloud. What the
  // columns were read from is checked at the end: the sections must begin where this stream turns out to end.
  const bool use_cols = !sect && col_flags != nullptr && col_flags[c] != 0u && plan.n_adaptive != 0u && plan.n_adaptive <= 8u;  // (uniform)
  const uint32_t n_cols = use_cols ? plan.n_adaptive : 0u;
  if (!sect && (n_ops == 0u || n == 0u)) {  // no per-point encoder (integer fields only): the sections begin at once
    if (tid == 0) {
      reg_end[c] = 0u;
      atomicAdd(&status[kStatFastRegular], 1u);
    }
    return;
  }
  const uint8_t* ebits = (!sect && token_ends != nullptr) ? reinterpret_cast<const uint8_t*>(token_ends + token_ends_word(dc.src_off, c)) : nullptr;
  // every op raw (EncodingOptions::NONE, lossless floats): a point has a fixed size F and the token ends follow from the
  // byte offset alone -- pat[r] = end bits of the 16 bytes that begin r bytes into a point (no bitmap is read)
  uint16_t* pat = reinterpret_cast<uint16_t*>(smem + L::kMiscOff + 64u);  // [kSwMaxPointBytes]
  uint32_t fixed_F = 0u;
  {
    bool all_raw = true;
    uint32_t F = 0u;
    for (uint32_t o = 0; o < n_ops; ++o) {
      const uint32_t k = plan.ops[ob + o].kind;
      all_raw = all_raw && (k == OP_COPY || k == OP_XOR32 || k == OP_XOR64);
      F += plan.ops[ob + o].size;
    }
    if (all_raw && F <= kSwMaxPointBytes) fixed_F = F;
  }
  if (fixed_F != 0u && tid < fixed_F) {
    uint32_t bits = 0u, r = tid;
    for (uint32_t i = 0; i < 16u; ++i) {
      uint32_t acc = 0u;
      bool is_end = false;
      for (uint32_t o = 0; o < n_ops; ++o) {
        acc += plan.ops[ob + o].size;
        is_end = is_end || r + 1u == acc;
      }
      if (is_end) bits |= 1u << i;
      r = r + 1u == fixed_F ? 0u : r + 1u;
    }
    pat[tid] = (uint16_t)bits;
  }
  const uint32_t inv_F = fixed_F > 1u ? (uint32_t)((0x100000000ull + fixed_F - 1u) / fixed_F) : 0u;

  if (tid == 0) {
    misc[0] = 0u;
    misc[2] = 0xffffffffu;
    misc[3] = 0u;
  }
  for (uint32_t i = tid; i < (L::kMiscOff - L::kTrecOff) / 4u; i += T) reinterpret_cast<uint32_t*>(trec)[i] = 0u;
  // which ends of a byte's end mask close points when the first k0 of them do not: entry = mask | (next k0) << 8
  for (uint32_t e = tid; !FORM && e < n_ops * 256u; e += T) {
    uint32_t k = e >> 8, sel = 0u;
    for (uint32_t bit = 0; bit < 8u; ++bit) {
      if ((e >> bit) & 1u) {
        if (k == 0u) {
          sel |= 1u << bit;
          k = n_ops - 1u;
        } else {
          --k;
        }
      }
    }
    lut[e] = (uint16_t)(sel | (k << 8));
  }
  __syncthreads();

  // x / n_ops for x < 2^19 without a division (n_ops <= 8: the error of the rounded-up reciprocal stays below 2^-13)
  const uint32_t inv_ops = n_ops > 1u ? (uint32_t)((0x100000000ull + n_ops - 1u) / n_ops) : 0u;
  auto div_ops = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t { return n_ops > 1u ? __umulhi(x, inv_ops) : x; };
  const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
  const uint8_t* src_al = src - a0;
  const uint32_t vend = a0 + src_size;
  const uint32_t n_pieces = src_size ? (vend + kSwPiece - 1u) / kSwPiece : 0u;
  uint8_t* wmem = smem + wave * L::kWaveBytes;
  uint32_t* wbuf = reinterpret_cast<uint32_t*>(wmem + L::kBytesOff);
  uint16_t* ebuf = reinterpret_cast<uint16_t*>(wmem + L::kEndsOff);
  uint16_t* plist = reinterpret_cast<uint16_t*>(wmem + L::kListOff);
  uint16_t* jt = reinterpret_cast<uint16_t*>(wmem + L::kJumpOff);    // (FORM)
  uint16_t* cpt = reinterpret_cast<uint16_t*>(wmem + L::kCheckOff);  // (MODE 1)
  uint16_t* hop4 = reinterpret_cast<uint16_t*>(wmem + L::kCheckOff); // (MODE 2)
  uint32_t* rstate = reinterpret_cast<uint32_t*>(wmem + L::kCheckOff);        // (MODE 2, instead of the checkpoints) [kSwMaxRounds][kSwMaxOps]: windows of round r
  uint16_t* rstart = reinterpret_cast<uint16_t*>(wmem + L::kCheckOff + kSwMaxRounds * kSwMaxOps * 4u);  // [kSwMaxRounds + 1]: first point of round r

  auto load_unit = [&](uint32_t v0, uint32_t(&u)[4]) __attribute__((always_inline)) {
    const bool ok = v0 < vend;
    const uint4 w = *reinterpret_cast<const uint4*>(src_al + (ok ? v0 : 0u));
    u[0] = ok ? w.x : 0xffffffffu;
    u[1] = ok ? w.y : 0xffffffffu;
    u[2] = ok ? w.z : 0xffffffffu;
    u[3] = ok ? w.w : 0xffffffffu;
  };
  // end bits of the unit at v0: MSBs, or 16 bits of the bitmap (bit p of it = payload byte p); bytes outside the payload end nothing
  auto unit_ends = [&](uint32_t v0, const uint32_t(&u)[4]) __attribute__((always_inline)) -> uint32_t {
    uint32_t ends;
    if (fixed_F != 0u) {
      const uint32_t o = v0 > a0 ? v0 - a0 : 0u;                              // payload offset (the first unit: bytes in front are cut below)
      const uint32_t r = fixed_F > 1u ? o - __umulhi(o, inv_F) * fixed_F : 0u;  // o % F (o < 2^22, F <= 88)
      ends = pat[r];
      if (v0 < a0) ends <<= (a0 - v0);
    } else if (FORM || ebits == nullptr) {
      ends = wp_ends16(u);
    } else {
      const int32_t o = (int32_t)v0 - (int32_t)a0;  // payload offset of the unit's first byte (first unit: may be < 0)
      const uint32_t ob = o > 0 ? (uint32_t)o : 0u;
      uint32_t w = 0u;
      if (ob < src_size) __builtin_memcpy(&w, ebits + (ob >> 3), 4);  // (the bitmap has a word of slack per chunk)
      w >>= (ob & 7u);
      ends = o >= 0 ? (w & 0xffffu) : ((w << (uint32_t)(-o)) & 0xffffu);
    }
    const uint32_t lo = a0 > v0 ? min(a0 - v0, 16u) : 0u;
    const uint32_t hi = vend > v0 ? min(vend - v0, 16u) : 0u;
    return ends & ((1u << hi) - 1u) & ~((1u << lo) - 1u);
  };

  // per-op state: lane o of these registers belongs to op o
  uint32_t kind_l = 0xffu, size_l = 0u;
  if (lane < n_ops) {
    kind_l = plan.ops[ob + lane].kind;
    size_l = plan.ops[ob + lane].size;
  }
  const bool raw_l = kind_l == OP_COPY || kind_l == OP_XOR32 || kind_l == OP_XOR64;
  const unsigned long long raw_ops = __ballot(raw_l);                                   // bit o: op o's token is `size` raw bytes
  const unsigned long long gor_ops = __ballot(kind_l == OP_GORILLA64);                  // (MODE 2: one or more)
  const unsigned long long xor_ops = __ballot(kind_l == OP_XOR32 || kind_l == OP_XOR64 || kind_l == OP_GORILLA64);  // its values combine with ^
  const unsigned long long copy_ops = __ballot(kind_l == OP_COPY);                      // no state at all
  const unsigned long long int_ops = __ballot(kind_l == OP_INT);                        // a marker is an error there
  const unsigned long long narrow_ops = __ballot(kind_l == OP_QF32 || (kind_l == OP_INT && size_l <= 4u));  // 32 bits of the running value are all that is used
  // FORM modes keep the first walk's values in LDS for the second (round 6) when they fit where the jump table was: 4 bytes per
  // value of a narrow op, 8 otherwise. Lane o: the bytes per point of the ops in front of op o.
  uint32_t vpre_l = 0u, vtot = 0u;
  if constexpr (FORM) {
    const uint32_t vs = lane < n_ops ? (((narrow_ops >> lane) & 1ull) ? 4u : 8u) : 0u;
    const uint32_t incl = wave_inclusive_scan(vs);
    vpre_l = incl - vs;
    vtot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
  uint64_t run_l = 0ull;  // lane o: op o's running value (behind the last point handled so far)
  uint32_t st_last = 0u;  // MODE 2, lane o: op o's Gorilla window behind the piece this wave handled last

  uint32_t b[4], bh[4];
  uint32_t p = wave;
  if (n_pieces) {
    load_unit(min(p, n_pieces) * kSwPiece + lane * 16u, b);
    load_unit((min(p, n_pieces) + 1u) * kSwPiece + min(lane, 5u) * 16u, bh);
  }
  bool gave_up = false;
#ifdef CLDN_SW_TRACE
  unsigned long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl_ = 0;
  uint32_t trn_ = 0;
#endif
  __builtin_amdgcn_s_setprio(1);
  for (; p < n_pieces; p += NW) {
#ifdef CLDN_SW_TRACE
    if (p == wave) tl_ = __builtin_amdgcn_s_memtime();
#endif
    SW_T(7)
    // ---- bytes and end bits -> LDS
    *reinterpret_cast<uint4*>(wbuf + lane * 4u) = make_uint4(b[0], b[1], b[2], b[3]);
    if (lane < 6u) *reinterpret_cast<uint4*>(wbuf + kSwPiece / 4u + lane * 4u) = make_uint4(bh[0], bh[1], bh[2], bh[3]);
    const uint32_t v0 = p * kSwPiece + lane * 16u;
    const uint32_t ends = unit_ends(v0, b);
    ebuf[lane] = (uint16_t)ends;
    if (lane < 6u) ebuf[64u + lane] = (uint16_t)unit_ends((p + 1u) * kSwPiece + lane * 16u, bh);
    if (lane == 6u) {
      ebuf[70] = 0u;
      ebuf[71] = 0u;
    }
    const uint32_t cl = (uint32_t)__builtin_popcount(ends);
    const uint32_t incl = wave_inclusive_scan(cl);
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t tb = incl - cl;
    {
      const uint32_t pn = min(p + (uint32_t)NW, n_pieces);
      load_unit(pn * kSwPiece + lane * 16u, b);
      load_unit((pn + 1u) * kSwPiece + min(lane, 5u) * 16u, bh);
    }
    uint32_t q_first = 0u, npts = 0u, n_rounds = 1u, st_first = 0u;  // (MODE 2: rounds of the piece, window of the first)
    bool stop = false;  // the regular stream ended in front of this piece (uniform)
    if constexpr (FORM) {
      // ---- jump table: where the point that would start at each of my 16 bytes ends (0xffff: no point can start there).
      // st (MODE 2) = the Gorilla window in effect: valid << 16 | leading << 8 | trailing; bit 15 of an entry: the point's
      // '11' token opens a DIFFERENT window (the table is not valid behind that point)
      wp_wave_sync();  // (ebuf is complete)
      const uint32_t* eb32j = reinterpret_cast<const uint32_t*>(ebuf);
      uint32_t R[4];  // end bits of the 128 bytes from my first
      {
        const uint32_t wi = lane >> 1, wsft = (lane & 1u) * 16u;
        const uint32_t q0 = eb32j[wi], q1 = eb32j[wi + 1u], q2 = eb32j[wi + 2u], q3 = eb32j[wi + 3u], q4 = eb32j[wi + 4u];
        R[0] = __builtin_amdgcn_alignbit(q1, q0, wsft);
        R[1] = __builtin_amdgcn_alignbit(q2, q1, wsft);
        R[2] = __builtin_amdgcn_alignbit(q3, q2, wsft);
        R[3] = __builtin_amdgcn_alignbit(q4, q3, wsft);
      }
      // the form of the point that would start at byte i of mine: returns its end (relative to my first byte), *ok,
      // *new_st = the window its '11' token opens (0 = none)
      // (st: lane o = the window of op o in effect. owner < 64: the windows the '11' tokens of THAT lane's point open are
      // captured into *st_out, lane o again.)
      auto point_form = [&](uint32_t i, uint32_t st, bool* ok_out, bool* changed, uint32_t owner, uint32_t* st_out) __attribute__((always_inline)) -> uint32_t {
        uint32_t rel = i;
        bool ok = true, chg = false;
        for (uint32_t o = 0; o < n_ops; ++o) {  // uniform
          if (GOR && ((gor_ops >> o) & 1ull)) {
            const uint32_t sto = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)o);
            const uint32_t bp = lane * 16u + rel;  // byte of the piece's LDS copy
            const uint32_t w = __builtin_amdgcn_alignbit(wbuf[(bp >> 2) + 1u], wbuf[bp >> 2], (bp & 3u) * 8u);
            uint32_t len, nst = 0u;
            if (p == 0u && lane * 16u + i == a0) {
              len = 8u;  // the chunk's first value: raw bits
            } else if ((w & 1u) == 0u) {
              len = 1u;
            } else if ((w & 2u) == 0u) {
              const uint32_t m = 64u - ((sto >> 8) & 0xffu) - (sto & 0xffu);
              ok = ok && (sto >> 16) != 0u;  // (no window yet: the serial decoder raises the error)
              len = (2u + m + 7u) >> 3;
            } else {
              const uint32_t sl = (w >> 2) & 31u, m = ((w >> 7) & 63u) + 1u;
              ok = ok && sl + m <= 64u;
              len = (13u + m + 7u) >> 3;
              nst = 0x10000u | (sl << 8) | ((64u - sl - m) & 0xffu);
              chg = chg || nst != sto;
            }
            if (owner < 64u) {
              const uint32_t cv = (uint32_t)__builtin_amdgcn_readlane((int)nst, (int)owner);
              if (lane == o && cv != 0u) *st_out = cv;
            }
            rel += len;
          } else if ((raw_ops >> o) & 1ull) {
            rel += (uint32_t)__builtin_amdgcn_readlane((int)size_l, (int)o);
          } else {
            const uint32_t idx = rel >> 5, sh = rel & 31u;
            const uint32_t elo = idx == 0u ? R[0] : (idx == 1u ? R[1] : (idx == 2u ? R[2] : R[3]));
            const uint32_t ehi = idx == 0u ? R[1] : (idx == 1u ? R[2] : (idx == 2u ? R[3] : 0u));
            const uint32_t e = __builtin_amdgcn_alignbit(ehi, elo, sh) & 0x3ffu;  // a varint has 10 bytes at most
            ok = ok && e != 0u;
            rel += e ? (uint32_t)__builtin_ctz(e) + 1u : 1u;
          }
          rel = min(rel, 120u);
        }
        *ok_out = ok;
        *changed = chg;
        return rel;
      };
      // MODE 2 (round 6): the same table OP-OUTER / BYTE-INNER. The sixteen candidate points of a lane advance together, one
      // op at a time: what kind of token the op writes is decided once per op (not once per byte and op), the sixteen Gorilla
      // look-ups of an op are in flight together, and the run of varints a point begins with costs one pass over the end bits:
      // the k-th end at or behind byte i + 1 is the k-th end behind byte i unless byte i ends a token itself (then it is the
      // next one). Points whose leading varints do not end within 64 bytes of the lane's first byte get no entry (the chunk
      // goes to the serial decoder if such a point is real: tokens of more than 10 bytes are malformed anyway).
      [[maybe_unused]] auto make_jt_gor = [&](uint32_t st, bool with_hop4) __attribute__((always_inline)) {
        // rel of the sixteen candidates: four to a register, one byte each (a point advances by at most 64 bytes in its leading
        // varints and by at most 10 per op behind them: < 256)
        uint32_t relp[4] = {0u, 0u, 0u, 0u};
        auto rel_get = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t { return (relp[i >> 2] >> ((i & 3u) * 8u)) & 0xffu; };
        auto rel_add = [&](uint32_t i, uint32_t d) __attribute__((always_inline)) { relp[i >> 2] += d << ((i & 3u) * 8u); };
        uint32_t okm = 0xffffu, chgm = 0u;
        const uint32_t k0 = (uint32_t)__builtin_ctzll(gor_ops | raw_ops | (1ull << n_ops));  // varints a point begins with (uniform)
        if (k0 != 0u) {
          uint64_t m = (((uint64_t)R[1]) << 32) | R[0];
          for (uint32_t j = 1; j < k0; ++j) m &= m - 1ull;  // uniform trip count
#pragma unroll
          for (uint32_t i = 0; i < 16u; ++i) {
            // the lowest set bit of m: the k0-th token end at or behind byte i
            if (m == 0ull) okm &= ~(1u << i);
            rel_add(i, m ? (uint32_t)__builtin_ctzll(m) + 1u : 64u);
            if ((R[0] >> i) & 1u) m &= m - 1ull;
          }
        } else {
          relp[0] = 0x03020100u;
          relp[1] = 0x07060504u;
          relp[2] = 0x0b0a0908u;
          relp[3] = 0x0f0e0d0cu;
        }
        for (uint32_t o = k0; o < n_ops; ++o) {  // uniform
          if ((gor_ops >> o) & 1ull) {
            const uint32_t sto = (uint32_t)__builtin_amdgcn_readlane((int)st, (int)o);
            const uint32_t m10 = 64u - ((sto >> 8) & 0xffu) - (sto & 0xffu);
            const uint32_t len10 = (2u + m10 + 7u) >> 3;
            const bool have_win = (sto >> 16) != 0u;  // (no window yet: the serial decoder raises the error)
#pragma unroll
            for (uint32_t h = 0; h < 16u; h += 8u) {  // eight look-ups in flight
              uint32_t w[8];
#pragma unroll
              for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t bp = lane * 16u + rel_get(h + i);
                w[i] = __builtin_amdgcn_alignbit(wbuf[(bp >> 2) + 1u], wbuf[bp >> 2], (bp & 3u) * 8u);
              }
#pragma unroll
              for (uint32_t i = 0; i < 8u; ++i) {
                const uint32_t sl = (w[i] >> 2) & 31u, m = ((w[i] >> 7) & 63u) + 1u;
                const uint32_t nst = 0x10000u | (sl << 8) | ((64u - sl - m) & 0xffu);
                const bool t0 = (w[i] & 1u) == 0u, t10 = (w[i] & 3u) == 1u, t11 = (w[i] & 3u) == 3u;
                uint32_t len = t0 ? 1u : (t10 ? len10 : (m + 20u) >> 3);
                if (p == 0u && lane * 16u + h + i == a0) len = 8u;  // the chunk's first value: raw bits
                else {
                  if ((t10 && !have_win) || (t11 && sl + m > 64u)) okm &= ~(1u << (h + i));
                  if (t11 && nst != sto) chgm |= 1u << (h + i);
                }
                rel_add(h + i, len);
              }
            }
          } else if ((raw_ops >> o) & 1ull) {
            const uint32_t sz = (uint32_t)__builtin_amdgcn_readlane((int)size_l, (int)o) * 0x01010101u;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) relp[k] += sz;
          } else {
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
              const uint32_t rl = rel_get(i);
              const uint32_t idx = rl >> 5, sh = rl & 31u;
              const uint32_t elo = idx == 0u ? R[0] : (idx == 1u ? R[1] : (idx == 2u ? R[2] : (idx == 3u ? R[3] : 0u)));
              const uint32_t ehi = idx == 0u ? R[1] : (idx == 1u ? R[2] : (idx == 2u ? R[3] : 0u));
              const uint32_t e = __builtin_amdgcn_alignbit(ehi, elo, sh) & 0x3ffu;  // a varint has 10 bytes at most
              if (e == 0u) okm &= ~(1u << i);
              rel_add(i, e ? (uint32_t)__builtin_ctz(e) + 1u : 1u);
            }
          }
        }
        uint32_t packed[8];
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
          const uint32_t x = lane * 16u + i;        // byte of the piece
          const uint32_t len = rel_get(i) - i;
          const uint32_t end = x + len;             // (v-space, relative to the piece)
          const bool ok = ((okm >> i) & 1u) != 0u && len <= kSwMaxPointBytes && p * kSwPiece + x >= a0 && p * kSwPiece + end <= vend;
          uint32_t jv = ok ? end : 0xffffu;
          if (ok && ((chgm >> i) & 1u)) jv |= 0x8000u;
          if (i & 1u) packed[i >> 1] |= jv << 16;
          else packed[i >> 1] = jv;
        }
        uint4* dst = reinterpret_cast<uint4*>(jt + lane * 16u);
        dst[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        dst[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
        if (!with_hop4) return;  // (uniform)
        // hop4[x] = where the FOURTH point from byte x ends, if the four points x, x1, x2, x3 are well formed, begin inside the
        // piece and none of them changes the window (0xffff otherwise: single steps take over). Three rounds of sixteen
        // independent look-ups; candidates and the list then follow the stream four points at a time.
        wp_wave_sync();
        uint32_t good = 0u;  // bit i: the chain from byte i is intact so far
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
          const uint32_t jv = (packed[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu;
          if (jv < kSwPiece) good |= 1u << i;  // (0xffff and flagged entries are >= 0x8000)
        }
#pragma unroll
        for (uint32_t r = 1; r < 4u; ++r) {
          uint32_t nxt[8];
#pragma unroll
          for (uint32_t i = 0; i < 16u; ++i) {
            const uint32_t cur = (packed[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu;
            const uint32_t e = jt[((good >> i) & 1u) ? cur : 0u];
            if (i & 1u) nxt[i >> 1] |= e << 16;
            else nxt[i >> 1] = e;
          }
#pragma unroll
          for (uint32_t i = 0; i < 16u; ++i) {
            const uint32_t e = (nxt[i >> 1] >> ((i & 1u) * 16u)) & 0xffffu;
            // steps 1, 2: the next point must begin inside the piece; step 3: any well-formed end (it may lie in the next piece)
            const bool ok = r < 3u ? e < kSwPiece : e < 0x8000u;
            if (!ok) good &= ~(1u << i);
          }
#pragma unroll
          for (uint32_t k = 0; k < 8u; ++k) packed[k] = nxt[k];
        }
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
          if (((good >> i) & 1u) == 0u) packed[i >> 1] |= 0xffffu << ((i & 1u) * 16u);
        }
        uint4* hd = reinterpret_cast<uint4*>(hop4 + lane * 16u);
        hd[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        hd[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
      };
      auto make_jt = [&](uint32_t st, bool with_hop4) __attribute__((always_inline)) {
        if constexpr (GOR) {
          make_jt_gor(st, with_hop4);
          return;
        }
        uint32_t packed[8];
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
          bool ok, chg;
          uint32_t unused = 0u;
          const uint32_t rel = point_form(i, st, &ok, &chg, 64u, &unused);
          const uint32_t x = lane * 16u + i;        // byte of the piece
          const uint32_t end = x + (rel - i);       // (v-space, relative to the piece)
          ok = ok && rel - i <= kSwMaxPointBytes && p * kSwPiece + x >= a0 && p * kSwPiece + end <= vend;
          uint32_t jv = ok ? end : 0xffffu;
          if (GOR && ok && chg) jv |= 0x8000u;
          if (i & 1u) packed[i >> 1] |= jv << 16;
          else packed[i >> 1] = jv;
        }
        uint4* dst = reinterpret_cast<uint4*>(jt + lane * 16u);
        dst[0] = make_uint4(packed[0], packed[1], packed[2], packed[3]);
        dst[1] = make_uint4(packed[4], packed[5], packed[6], packed[7]);
      };
      // MODE 2: the table is built for the window this wave saw last (windows change rarely: the guess is right for all but a
      // few pieces); a piece whose guess was wrong, or whose own '11' tokens change the window, is redone in rounds below
      SW_T(0)
      uint32_t st_pred = GOR ? st_last : 0u;
      if constexpr (GOR) {
        // a newer guess than this wave's own last piece: the window behind the NEWEST piece in front of this one that has
        // published (lane l looks at piece p - 1 - l). A wave's first piece has no guess of its own: it waits until piece 0
        // -- where a chunk's windows settle -- has published, instead of building a table for "no window" that is thrown away.
        if (p != 0u) {
          const uint32_t back = min(min(p, (uint32_t)NW), L::kRing - 1u);
          for (uint32_t spins = 0;; ++spins) {
            const uint32_t q = p - 1u - min(lane, back - 1u);  // (a piece in front of mine)
            const unsigned long long x = wp_rec_load(trec + (q & (L::kRing - 1u)));
            const unsigned long long have = __ballot(lane < back && (uint32_t)(x >> 32) == q + 1u);
            if (have != 0ull) {
              const uint32_t q1 = p - 1u - (uint32_t)__builtin_ctzll(have);
              const unsigned long long xg = wp_rec_load(grec + (size_t)(q1 & (L::kRing - 1u)) * kSwMaxOps + min(lane, n_ops - 1u));
              if (__ballot(lane < n_ops && (uint32_t)(xg >> 32) != q1 + 1u) == 0ull) st_pred = lane < n_ops ? (uint32_t)xg : 0u;
              break;
            }
            if (p >= (uint32_t)NW || spins >= 4096u) break;  // (only a wave's first piece waits; bounded: the guess is a guess)
            __builtin_amdgcn_s_sleep(kWpSleep);
          }
        }
      }
      make_jt(st_pred, true);
      wp_wave_sync();
      SW_T(1)
      // ---- the first lanes follow the jumps from their byte: where the next piece is entered, and after how many points.
      // On the way a candidate leaves a checkpoint in every 128-byte block it passes: its first point there and how many
      // came before -- the owner of the true entry then lists its points eight blocks side by side.
      const uint32_t maxpt = min(plan.max_regular_bytes, kSwMaxPointBytes);  // candidates: entry offsets [0, maxpt)
      uint32_t ex[2] = {0xffffu, 0xffffu}, ec[2] = {0u, 0u};
      if constexpr (GOR) {
        // (round 6) four points per step through hop4, single steps where it has no entry (the piece's last points, a point
        // that changes the window: 0xfffe, a malformed one: 0xffff); no checkpoints -- the list follows hop4 as well
#pragma unroll
        for (uint32_t set = 0; set < 2u; ++set) {
          if (set == 1u && maxpt <= 64u) break;  // (uniform)
          uint32_t x = set * 64u + lane < maxpt ? set * 64u + lane : 0xffffu, cn = 0u;
          while (__ballot(x < kSwPiece) != 0ull) {
            const bool act = x < kSwPiece;
            const uint32_t xa = act ? x : 0u;
            const uint32_t h = hop4[xa], e = jt[xa];
            const uint32_t single = e == 0xffffu ? 0xffffu : ((e & 0x8000u) ? 0xfffeu : e);
            const uint32_t nx = h != 0xffffu ? h : single;
            cn += act ? (h != 0xffffu ? 4u : 1u) : 0u;
            x = act ? nx : x;
          }
          ex[set] = x;
          ec[set] = cn;
        }
      } else {
        uint32_t x0 = lane < maxpt ? lane : 0xffffu, x1 = 64u + lane < maxpt ? 64u + lane : 0xffffu;
        uint32_t c0 = 0u, c1 = 0u, pb0 = 0xffu, pb1 = 0xffu;
        for (uint32_t b8 = 0; b8 < 8u; ++b8) {
          if (lane < maxpt) cpt[lane * 8u + b8] = 0xffffu;
          if (64u + lane < maxpt) cpt[(64u + lane) * 8u + b8] = 0xffffu;
        }
        if (maxpt <= 64u) {  // (uniform) points of at most 64 bytes: one candidate per lane, no second set to carry along
          while (__ballot(x0 < kSwPiece) != 0ull) {
            if (x0 < kSwPiece) {
              const uint32_t bk = x0 >> 7;
              if (bk != pb0) cpt[lane * 8u + bk] = (uint16_t)((x0 & 127u) | (c0 << 7));
              pb0 = bk;
              x0 = jt[x0];
              if (GOR && x0 != 0xffffu && (x0 & 0x8000u)) x0 = 0xfffeu;  // the window changes behind this point: rounds
              ++c0;
            }
          }
        } else
        while (__ballot(x0 < kSwPiece || x1 < kSwPiece) != 0ull) {  // (the two rounds of candidates side by side)
          if (x0 < kSwPiece) {
            const uint32_t bk = x0 >> 7;
            if (bk != pb0) cpt[lane * 8u + bk] = (uint16_t)((x0 & 127u) | (c0 << 7));
            pb0 = bk;
            x0 = jt[x0];
            if (GOR && x0 != 0xffffu && (x0 & 0x8000u)) x0 = 0xfffeu;  // the window changes behind this point: rounds
            ++c0;
          }
          if (x1 < kSwPiece) {
            const uint32_t bk = x1 >> 7;
            if (bk != pb1) cpt[(64u + lane) * 8u + bk] = (uint16_t)((x1 & 127u) | (c1 << 7));
            pb1 = bk;
            x1 = jt[x1];
            if (GOR && x1 != 0xffffu && (x1 & 0x8000u)) x1 = 0xfffeu;
            ++c1;
          }
        }
        ex[0] = x0;
        ec[0] = c0;
        ex[1] = x1;
        ec[1] = c1;
      }
      SW_T(2)
      // ---- chain 1: {entry offset, points in front} of the piece (MODE 2: and the window at its entry)
      uint32_t entry = a0, pts0 = 0u, st = 0u;  // (st: lane o = window of op o)
      SW_PRIO_HOP();  // (round 6: from a record's arrival to this piece's own record at the highest priority, as in k_decode_points_w)
      if (p != 0u) {
        const unsigned long long* r = trec + ((p - 1u) & (L::kRing - 1u));
        const unsigned long long* rg = grec + (size_t)((p - 1u) & (L::kRing - 1u)) * kSwMaxOps + min(lane, n_ops - 1u);
        unsigned long long x = wp_rec_load(r), xg = GOR ? wp_rec_load(rg) : 0ull;
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) != p || (GOR && __ballot((uint32_t)(xg >> 32) != p) != 0ull)) {
          __builtin_amdgcn_s_setprio(0);
          for (uint32_t spins = 1u;; ++spins) {
            __builtin_amdgcn_s_sleep(kWpSleep);
            x = wp_rec_load(r);
            if (GOR) xg = wp_rec_load(rg);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) == p && (!GOR || __ballot((uint32_t)(xg >> 32) != p) == 0ull)) break;
            if ((spins & 63u) == 0u && (spins >= kSwSpinLimit || *(volatile uint32_t*)&misc[3] != 0u)) {
              gave_up = true;
              break;
            }
          }
          SW_PRIO_HOP();
        }
        const uint32_t rv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
        entry = rv & 0xffu;
        pts0 = rv >> 8;
        if (GOR) st = lane < n_ops ? (uint32_t)xg : 0u;
      }
      SW_T(3)
      if (gave_up) break;
      // (entry 0xff: the piece in front could not be left through a well-formed point -- malformed, or only the bytes
      // behind the regular stream's end; either way nothing starts here)
      const bool dead = entry == 0xffu || entry >= kSwMaxPointBytes;
      const uint32_t el = dead ? 0u : entry;
      const uint32_t my_exit = (uint32_t)__builtin_amdgcn_readlane((int)(el < 64u ? ex[0] : ex[1]), (int)(el & 63u));
      const uint32_t my_cnt = (uint32_t)__builtin_amdgcn_readlane((int)(el < 64u ? ec[0] : ec[1]), (int)(el & 63u));
      const bool rounds_needed = GOR && !dead && pts0 < n && (__ballot(st != st_pred) != 0ull || my_exit == 0xfffeu);
      if (!rounds_needed) {
        const uint32_t out_entry = (dead || my_exit >= 0xfffeu) ? 0xffu : my_exit - kSwPiece;
        const uint32_t pts1 = dead ? pts0 : min(pts0 + my_cnt, n);
        if (lane == 0u) wp_rec_store(trec + (p & (L::kRing - 1u)), ((unsigned long long)(p + 1u) << 32) | (pts1 << 8) | out_entry);
        if (GOR && lane < n_ops) wp_rec_store(grec + (size_t)(p & (L::kRing - 1u)) * kSwMaxOps + lane, ((unsigned long long)(p + 1u) << 32) | st);
        __builtin_amdgcn_s_setprio(1);
        st_last = st;
        if (pts0 >= n) stop = true;
        else if (dead) {
          if (lane == 0u) misc[0] = 1u;  // points are missing and the stream cannot be followed: the serial decoder raises the error
          stop = true;
        } else {
          q_first = pts0;
          npts = min(my_cnt, n - pts0);
          st_first = st;
          // ---- the points' first bytes, in order
          wp_wave_sync();
          bool broken = false;
          if constexpr (GOR) {
            // (round 6) lanes 0..3 take the points 4k + lane: lane r makes r single steps from the entry, then all four follow
            // hop4; where it has no entry the rest is listed by single steps from the lowest point a lane stopped at
            uint32_t x = entry, j = lane;
            bool go = lane < 4u;
#pragma unroll
            for (uint32_t s4 = 0; s4 < 3u; ++s4) {
              const uint32_t e = jt[go && x < kSwPiece ? x : 0u];
              if (lane > s4) x = (x < kSwPiece && e < 0x8000u) ? e : 0xffffu;  // (a flagged or malformed point, or one in the next piece: the tail walk sorts it out)
            }
            uint32_t stop_j = 0xffffu, stop_x = 0u;  // where this lane's hops ended: point stop_j begins at stop_x
            if (go && (x >= kSwPiece || j >= npts)) {
              go = false;  // (nothing of mine in this piece; a lane in front of me stops at its point and the tail walk continues)
            }
            while (__ballot(go) != 0ull) {
              if (go) {
                plist[j] = (uint16_t)x;
                const uint32_t h = hop4[x];
                if (h != 0xffffu && h < kSwPiece && j + 4u < npts) {
                  x = h;
                  j += 4u;
                } else {
                  stop_j = j;
                  stop_x = x;
                  go = false;
                }
              }
            }
            // the lowest stop: from there on single steps (the points of the lanes that went further are listed again, same values)
            uint32_t sj = stop_j;
            sj = min(sj, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffu, (int)sj, 0x111, 0xf, 0xf, false));  // row_shr:1
            sj = min(sj, (uint32_t)__builtin_amdgcn_update_dpp((int)0xffffu, (int)sj, 0x112, 0xf, 0xf, false));  // row_shr:2
            const uint32_t tj = (uint32_t)__builtin_amdgcn_readlane((int)sj, 3);  // min over lanes 0..3
            if (tj != 0xffffu) {
              const unsigned long long owner = __ballot(lane < 4u && stop_j == tj);
              uint32_t wx = (uint32_t)__builtin_amdgcn_readlane((int)stop_x, (int)__builtin_ctzll(owner)), wj = tj;
              if (lane == 0u) {
                while (wj < npts) {
                  plist[wj] = (uint16_t)wx;
                  const uint32_t e = jt[wx];
                  if (e == 0xffffu) {
                    broken = true;  // (the candidates' walk counted the points up to here only)
                    break;
                  }
                  ++wj;
                  wx = e & 0x7fffu;
                  if (wx >= kSwPiece) break;
                }
                if (wj < npts) broken = true;
              }
            } else if (lane == 0u) {
              broken = true;  // (npts != 0: some lane has listed a point)
            }
            // behind the chunk's last point: the sections
            if (lane == 0u && !broken && pts0 + npts == n) {
              const uint32_t e = jt[plist[npts - 1u]];
              if (e == 0xffffu) broken = true;
              else misc[2] = p * kSwPiece + (e & 0x7fffu) - a0;
            }
          } else if (lane < 8u) {
            const uint32_t ck = cpt[entry * 8u + lane];
            if (ck != 0xffffu) {
              uint32_t x = lane * 128u + (ck & 127u), j = ck >> 7;
              while (x < (lane + 1u) * 128u && j < npts) {
                plist[j] = (uint16_t)x;
                const uint32_t nx = jt[x] & 0x7fffu;
                if (nx == 0x7fffu) {
                  broken = true;  // (the candidates' walk counted the points up to here only)
                  break;
                }
                if (pts0 + j + 1u == n) misc[2] = p * kSwPiece + nx - a0;  // behind the chunk's last point: the sections
                x = nx;
                ++j;
              }
            }
          }
          if (__ballot(broken) != 0ull && lane == 0u) misc[0] = 1u;
        }
      } else {
        // ---- MODE 2, rounds: table for the current window, one lane follows the jumps and lists the points, until the piece
        // is left, the chunk's points are complete, or a point changes the window
        uint32_t out_entry = 0xffu, pts1 = pts0;
        const uint32_t limit = min(n - pts0, L::kListEntries - 8u);
        st_first = st;
        uint32_t xcur = entry, j = 0u, rounds = 0u;
        bool broken = false;
        for (;;) {
          if (rounds != 0u || __ballot(st != st_pred) != 0ull) {
            wp_wave_sync();
            make_jt(st, false);
          }
          wp_wave_sync();
          if (lane == 0u) rstart[rounds] = (uint16_t)j;
          if (lane < kSwMaxOps) rstate[rounds * kSwMaxOps + lane] = st;
          uint32_t wx = xcur, wj = j, flag = 0u;
          if (lane == 0u) {
            while (wx < kSwPiece && wj < limit) {
              plist[wj] = (uint16_t)wx;
              const uint32_t e = jt[wx];
              if (e == 0xffffu) {
                flag = 2u;
                break;
              }
              ++wj;
              if (pts0 + wj == n) misc[2] = p * kSwPiece + (e & 0x7ffu) - a0;  // behind the chunk's last point: the sections
              wx = e & 0x7ffu;
              if (e & 0x8000u) {
                flag = 1u;
                break;
              }
            }
          }
          wx = (uint32_t)__builtin_amdgcn_readfirstlane((int)wx);
          wj = (uint32_t)__builtin_amdgcn_readfirstlane((int)wj);
          flag = (uint32_t)__builtin_amdgcn_readfirstlane((int)flag);
          ++rounds;
          if (flag == 2u) {
            broken = true;
            break;
          }
          if (flag == 1u) {
            // the windows the last listed point's '11' tokens opened: its owner lane works the point's form out once more
            const uint32_t xf = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)plist[wj - 1u]);
            uint32_t st_new = st;
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
              if (i == (xf & 15u)) {  // uniform
                bool ok, chg;
                point_form(i, st, &ok, &chg, xf >> 4, &st_new);
              }
            }
            st = st_new;
          }
          xcur = wx;
          j = wj;
          if (flag == 0u || wx >= kSwPiece || wj >= limit) break;
          if (rounds >= kSwMaxRounds) {  // the window changes all the time: the serial decoder takes the chunk
            broken = true;
            break;
          }
        }
        if (broken || (j < n - pts0 && xcur < kSwPiece)) {
          // (a list that ran out of room before the piece's end cannot happen: a point has two bytes at least)
          if (lane == 0u) misc[0] = 1u;
          stop = true;
        } else {
          q_first = pts0;
          npts = j;
          n_rounds = rounds;
          pts1 = pts0 + j;
          out_entry = xcur >= kSwPiece ? xcur - kSwPiece : 0xffu;
        }
        if (lane == 0u) wp_rec_store(trec + (p & (L::kRing - 1u)), ((unsigned long long)(p + 1u) << 32) | (pts1 << 8) | out_entry);
        if (lane < n_ops) wp_rec_store(grec + (size_t)(p & (L::kRing - 1u)) * kSwMaxOps + lane, ((unsigned long long)(p + 1u) << 32) | st);
        __builtin_amdgcn_s_setprio(1);
        st_last = st;
      }
    } else {
      // ---- chain 1: token ends in front of the piece
      uint32_t T0 = 0u;
      if (p != 0u) {
        const unsigned long long* r = trec + ((p - 1u) & (L::kRing - 1u));
        unsigned long long x = wp_rec_load(r);
        if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) != p) {
          __builtin_amdgcn_s_setprio(0);
          for (uint32_t spins = 1u;; ++spins) {
            __builtin_amdgcn_s_sleep(kWpSleep);
            x = wp_rec_load(r);
            if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) == p) break;
            if ((spins & 63u) == 0u && (spins >= kSwSpinLimit || *(volatile uint32_t*)&misc[3] != 0u)) {
              gave_up = true;
              break;
            }
          }
          __builtin_amdgcn_s_setprio(1);
        }
        T0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
      }
      if (gave_up) break;
      if (lane == 0u) wp_rec_store(trec + (p & (L::kRing - 1u)), ((unsigned long long)(p + 1u) << 32) | (T0 + cnt));
      if (T0 >= target) stop = true;
      if (!stop) {
      if (T0 + cnt >= target) {
        const uint32_t want = target - T0;
        if (tb < want && want <= tb + cl) {
          uint32_t m = ends;
          for (uint32_t k = tb + 1u; k < want; ++k) m &= m - 1u;
          misc[2] = v0 + (uint32_t)__builtin_ctz(m) + 1u - a0;
        }
      }
      // ---- where the points this piece owns begin
      const uint32_t A0 = div_ops(T0);
      const uint32_t r0 = T0 - A0 * n_ops;
      const uint32_t extra = p == 0u ? 1u : 0u;
      q_first = A0 + 1u - extra;
      npts = min(div_ops(r0 + cnt) + extra, n - q_first);
      {
        const uint32_t x = r0 + tb;
        const uint32_t a = div_ops(x);
        const uint32_t k0 = n_ops - 1u - (x - a * n_ops);
        const uint32_t e0 = lut[k0 * 256u + (ends & 0xffu)];
        const uint32_t e1 = lut[(e0 >> 8) * 256u + (ends >> 8)];
        uint32_t sel = (e0 & 0xffu) | ((e1 & 0xffu) << 8);
        uint32_t j = a + extra;
        while (sel) {
          plist[j] = (uint16_t)(lane * 16u + (uint32_t)__builtin_ctz(sel) + 1u);
          ++j;
          sel &= sel - 1u;
        }
        if (extra && lane == 0u) plist[0] = (uint16_t)a0;
      }
      }
    }
    if (stop) break;
    wp_wave_sync();

    // A point's tokens, op by op. fn(o, value, marker) is called for every op (uniform o); returns false when the point
    // is irregular. Token lengths: distance to the next end bit (varints), or the field's size (raw ops).
    auto walk = [&](uint32_t j, uint32_t byte0, bool have, auto&& fn) __attribute__((always_inline)) -> bool {
      // the end bits of the 96 bytes behind the point's first
      const uint32_t* eb32 = reinterpret_cast<const uint32_t*>(ebuf);
      const uint32_t ei = byte0 >> 5, es = byte0 & 31u;
      const uint32_t q0 = eb32[ei], q1 = eb32[ei + 1u], q2 = eb32[ei + 2u], q3 = eb32[ei + 3u];
      uint32_t E[3] = {__builtin_amdgcn_alignbit(q1, q0, es), __builtin_amdgcn_alignbit(q2, q1, es), __builtin_amdgcn_alignbit(q3, q2, es)};
      uint32_t pos = 0u;
      bool bad = false;
      for (uint32_t o = 0; o < n_ops; ++o) {  // uniform
        if (GOR && ((gor_ops >> o) & 1ull)) {
          uint32_t stp = (uint32_t)__builtin_amdgcn_readlane((int)st_first, (int)o);  // the window in effect: that of the point's round
          if (n_rounds > 1u) {      // uniform
            for (uint32_t r = 1; r < n_rounds; ++r)
              if (j >= (uint32_t)rstart[r]) stp = rstate[r * kSwMaxOps + o];
          }
          uint32_t len = 1u;
          const uint64_t v = sw_gorilla(wbuf, byte0 + pos, q_first + j == 0u, stp, &len, &bad);
          fn(o, have ? v : 0ull, false);
          pos = min(pos + len, kSwMaxPointBytes - 1u);
          continue;
        }
        const bool raw = (raw_ops >> o) & 1ull;
        const uint32_t size = (uint32_t)__builtin_amdgcn_readlane((int)size_l, (int)o);
        // end bits from `pos` on (pos < 88)
        const uint32_t wi = pos >> 5, ws = pos & 31u;
        const uint32_t elo = wi == 0u ? E[0] : (wi == 1u ? E[1] : E[2]);
        const uint32_t ehi = wi == 0u ? E[1] : (wi == 1u ? E[2] : 0u);
        const uint32_t e = __builtin_amdgcn_alignbit(ehi, elo, ws);
        uint32_t len;
        if (raw) {
          len = size;
          if (ebits != nullptr && ((e >> (size - 1u)) & 1u) == 0u) bad = true;  // the bitmap ends the field somewhere else
        } else {
          len = e ? (uint32_t)__builtin_ctz(e) + 1u : 11u;
          if (len > 10u) {
            bad = true;
            len = 10u;
          }
        }
        bool marker = false;
        const uint64_t v = __ballot(len > 4u) == 0ull ? sw_token32(wbuf, byte0 + pos, len, raw, &marker, &bad)
                                                      : sw_token(wbuf, byte0 + pos, len, raw, &marker, &bad);
        if (marker && ((int_ops >> o) & 1ull)) bad = true;  // "decodeVarint: unexpected NaN marker"
        fn(o, have ? v : 0ull, have && marker);
        pos = min(pos + len, kSwMaxPointBytes - 1u);
      }
      return bad;
    };

    SW_T(4)
    // ---- first walk: every op's aggregate over the piece (lane o: op o)
    uint64_t agg_l = 0ull;
    uint32_t aggf_l = 0u;  // lane o: op o was reset inside the piece (a marker)
    bool irregular = false;
    // values kept for the second walk: op o's from byte npad * vpre(o) of the table's place on, the rows' marker masks behind them
    const uint32_t npad = (npts + 1u) & ~1u;
    const uint32_t n_rows = (npts + 63u) >> 6;
    uint8_t* vbase = wmem + L::kJumpOff;
    const uint32_t marks_off = (npad * vtot + 7u) & ~7u;
    const bool keep = FORM && marks_off + n_rows * n_ops * 8u <= (n_rounds == 1u ? L::kWaveBytes - L::kJumpOff : L::kCheckOff - L::kJumpOff);  // (uniform)
    for (uint32_t r = 0; r * 64u < npts; ++r) {  // uniform
      const uint32_t j = r * 64u + lane;
      const bool have = j < npts;
      const uint32_t byte0 = have ? (uint32_t)plist[j] : 0u;
      const bool bad = walk(j, byte0, have, [&](uint32_t o, uint64_t v, bool mk) __attribute__((always_inline)) {
        if (keep) {
          const uint32_t vo = npad * (uint32_t)__builtin_amdgcn_readlane((int)vpre_l, (int)o);
          if (have) {
            if ((narrow_ops >> o) & 1ull) *reinterpret_cast<uint32_t*>(vbase + vo + j * 4u) = (uint32_t)v;
            else *reinterpret_cast<uint64_t*>(vbase + vo + j * 8u) = v;
          }
          const unsigned long long mks = __ballot(mk);
          if (lane == 0u) *reinterpret_cast<unsigned long long*>(vbase + marks_off + (r * n_ops + o) * 8u) = mks;
        }
        if ((copy_ops >> o) & 1ull) return;
        uint64_t tot;
        bool reset = false;
        if ((narrow_ops >> o) & 1ull) {  // int32 wrap-around arithmetic (FloatN lanes), fields of at most 4 bytes
          const unsigned long long mks = __ballot(mk);
          uint32_t v32 = (uint32_t)v;
          if (mks != 0ull) {  // the values behind the row's last marker
            const uint32_t last = 63u - (uint32_t)__builtin_clzll(mks);
            v32 = lane > last ? v32 : 0u;
            reset = true;
          }
          tot = wave_sum(v32);
        } else if ((xor_ops >> o) & 1ull) {
          tot = sw_lane64(sw_scan64<true>(v), 63u);
        } else {
          const unsigned long long mks = __ballot(mk);
          if (mks == 0ull) {
            tot = sw_lane64(sw_scan64<false>(v), 63u);
          } else {  // the values behind the row's last marker
            const uint32_t last = 63u - (uint32_t)__builtin_clzll(mks);
            tot = sw_lane64(sw_scan64<false>(lane > last ? v : 0ull), 63u);
            reset = true;
          }
        }
        if (lane == o) {
          if ((xor_ops >> o) & 1ull) agg_l ^= tot;
          else agg_l = reset ? tot : agg_l + tot;
          aggf_l |= reset ? 1u : 0u;
        }
      });
      irregular = irregular || (have && bad);
    }
    if (__ballot(irregular) != 0ull && lane == 0u) misc[0] = 1u;
    SW_T(5)
    // ---- chain 2: the running values in front of the piece (lane o: {tag, lo}, {tag | reset, hi} of op o)
    SW_PRIO_HOP();
    if (p != 0u) {
      const unsigned long long* r = vrec + ((size_t)((p - 1u) & (L::kRing - 1u)) * kSwMaxOps + min(lane, n_ops - 1u)) * 2u;
      unsigned long long xl = wp_rec_load(r), xh = wp_rec_load(r + 1);
      if (__ballot((uint32_t)(xl >> 32) != p || (uint32_t)(xh >> 32) != p) != 0ull) {
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t spins = 1u;; ++spins) {
          __builtin_amdgcn_s_sleep(kWpSleep);
          xl = wp_rec_load(r);
          xh = wp_rec_load(r + 1);
          if (__ballot((uint32_t)(xl >> 32) != p || (uint32_t)(xh >> 32) != p) == 0ull) break;
          if ((spins & 63u) == 0u && (spins >= kSwSpinLimit || *(volatile uint32_t*)&misc[3] != 0u)) {
            gave_up = true;
            break;
          }
        }
        SW_PRIO_HOP();
      }
      run_l = (xh << 32) | (xl & 0xffffffffull);
    } else {
      run_l = 0ull;
    }
    if (gave_up) break;
    {
      const uint64_t incl_l = ((xor_ops >> lane) & 1ull) ? (run_l ^ agg_l) : (aggf_l ? agg_l : run_l + agg_l);
      if (lane < n_ops) {
        unsigned long long* w = vrec + ((size_t)(p & (L::kRing - 1u)) * kSwMaxOps + lane) * 2u;
        wp_rec_store(w, ((unsigned long long)(p + 1u) << 32) | (incl_l & 0xffffffffull));
        wp_rec_store(w + 1, ((unsigned long long)(p + 1u) << 32) | (incl_l >> 32));
      }
    }
    __builtin_amdgcn_s_setprio(1);
    SW_T(6)
    // ---- second walk: values, converted and stored; a lane stores its own point
    for (uint32_t r = 0; r * 64u < npts; ++r) {  // uniform
      const uint32_t j = r * 64u + lane;
      const bool have = j < npts;
      const uint32_t byte0 = have ? (uint32_t)plist[j] : 0u;
      uint8_t* pt = base + __umul24(q_first + (have ? j : 0u), step);  // (q < 2^16, step <= 1024)
      auto emit = [&](uint32_t o, uint64_t v, bool mk) __attribute__((always_inline)) {
        const DevOp& op = plan.ops[ob + o];
        const uint32_t off = op.offset;
        const uint32_t kind = op.kind;
        if (kind == OP_COPY) {
          if (have && off != 0xffffffffu) st_raw(pt + off, v, op.size);
          return;
        }
        const uint64_t before = sw_lane64(run_l, o);  // (uniform)
        uint64_t cur;
        if (kind == OP_XOR32 || kind == OP_XOR64 || kind == OP_GORILLA64) {
          const uint64_t inc = sw_scan64<true>(v);
          cur = before ^ inc;
          const uint64_t after = before ^ sw_lane64(inc, 63u);
          if (lane == o) run_l = after;
          if (have && off != 0xffffffffu) st_raw(pt + off, cur, op.size);
          return;
        }
        const unsigned long long mks = __ballot(mk);
        if ((narrow_ops >> o) & 1ull) {
          const uint32_t b32 = (uint32_t)before;
          if (mks == 0ull) {
            const uint32_t inc = wave_inclusive_scan((uint32_t)v);
            cur = b32 + inc;
            const uint32_t after = b32 + (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            if (lane == o) run_l = after;
          } else {
            int32_t inc[1] = {mk ? 0 : (int32_t)(uint32_t)v};
            uint32_t f = mk ? 1u : 0u;
            wp_seg_scan<1>(inc, f);
            cur = f ? (uint32_t)inc[0] : b32 + (uint32_t)inc[0];
            const uint32_t f63 = (uint32_t)__builtin_amdgcn_readlane((int)f, 63);
            const uint32_t i63 = (uint32_t)__builtin_amdgcn_readlane(inc[0], 63);
            if (lane == o) run_l = f63 ? i63 : b32 + i63;
          }
        } else if (mks == 0ull) {
          const uint64_t inc = sw_scan64<false>(v);
          cur = before + inc;
          const uint64_t after = before + sw_lane64(inc, 63u);
          if (lane == o) run_l = after;
        } else {
          uint64_t inc = mk ? 0ull : v;
          uint32_t f = mk ? 1u : 0u;
          sw_seg_scan64(inc, f);
          cur = f ? inc : before + inc;
          const uint32_t f63 = (uint32_t)__builtin_amdgcn_readlane((int)f, 63);
          const uint64_t i63 = sw_lane64(inc, 63u);
          if (lane == o) run_l = f63 ? i63 : before + i63;
        }
        if (have && off != 0xffffffffu) {
          if (kind == OP_QF32) {
            st_raw(pt + off, mk ? 0x7fc00000u : __float_as_uint(__fmul_rn((float)(int32_t)cur, op.res_f)), 4);
          } else if (kind == OP_LOSSY_F32) {
            st_raw(pt + off, mk ? 0x7fc00000u : __float_as_uint(__fmul_rn((float)(long long)cur, op.res_f)), 4);
          } else if (kind == OP_LOSSY_F64) {
            st_raw(pt + off, mk ? 0x7ff8000000000000ull : (uint64_t)__double_as_longlong(__dmul_rn((double)(long long)cur, op.res_d)), 8);
          } else {
            st_raw(pt + off, cur, op.size);
          }
        }
      };
      if (keep) {
        for (uint32_t o = 0; o < n_ops; ++o) {  // uniform
          const uint32_t vo = npad * (uint32_t)__builtin_amdgcn_readlane((int)vpre_l, (int)o);
          uint64_t v = 0ull;
          if (have) {
            if ((narrow_ops >> o) & 1ull) v = (uint64_t)(int64_t)(int32_t)*reinterpret_cast<const uint32_t*>(vbase + vo + j * 4u);
            else v = *reinterpret_cast<const uint64_t*>(vbase + vo + j * 8u);
          }
          const unsigned long long mks = *reinterpret_cast<const unsigned long long*>(vbase + marks_off + (r * n_ops + o) * 8u);
          emit(o, v, ((mks >> lane) & 1ull) != 0ull);
        }
      } else {
        walk(j, byte0, have, emit);
      }
      if (n_cols != 0u) {  // (uniform) the point's column values: all requested before the first one is stored
        const uint32_t q = q_first + (have ? j : 0u);
        uint32_t cv[8];
#pragma unroll
        for (uint32_t a = 0; a < 8u; ++a) {
          cv[a] = 0u;
          if (a < n_cols) {
            const uint32_t f_bpv = plan.adaptive[a].bpv;
            const uint8_t* colp = sect_cols.p[a] + (size_t)dc.first_point * f_bpv;
            cv[a] = f_bpv == 2u ? (uint32_t)reinterpret_cast<const uint16_t*>(colp)[q] : reinterpret_cast<const uint32_t*>(colp)[q];
          }
        }
#pragma unroll
        for (uint32_t a = 0; a < 8u; ++a) {
          if (a < n_cols && have) st_raw(pt + plan.adaptive[a].offset, cv[a], plan.adaptive[a].bpv);
        }
      }
    }
    wp_wave_sync();
#ifdef CLDN_SW_TRACE
    ++trn_;
#endif
  }
#ifdef CLDN_SW_TRACE
  if (MODE == 2 && lane == 0u && (c == 0u || c == 300u) && (wave == 0u || wave == 5u || wave == 11u))
    printf("chunk %u wave %u pieces %u: A %llu jt %llu chase %llu wait1 %llu list %llu walk1 %llu wait2 %llu walk2 %llu\n", c, wave, trn_, tr_[0], tr_[1], tr_[2],
           tr_[3], tr_[4], tr_[5], tr_[6], tr_[7]);
#endif
  if (gave_up && lane == 0u) {
    misc[3] = 1u;
    misc[0] = 1u;
  }
  __syncthreads();
  if (tid == 0) {
    const uint32_t pos = misc[2];
    const bool redo = misc[0] != 0u || pos == 0xffffffffu;
    if (sect) {
      if (!redo && pos == src_size) atomicAdd(reg_end + c, 1u);  // the section's n tokens end exactly with it
    } else {
      reg_end[c] = redo ? kDecRedo : pos;
      if (!redo) atomicAdd(&status[kStatFastRegular], 1u);
      if (sec_done != nullptr) {  // like k_decode_points_w: 2 = the sections went out with the points
        const bool merged = !redo && use_cols && reg_end_pre != nullptr && reg_end_pre[c] == pos;
        sec_done[c] = merged ? 2u : 0u;
        if (merged) atomicAdd(&status[kStatFastSections], 1u);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_decode_fixed (end of round 4): regular streams of fixed-size tokens only -- FieldDecoderFloat_XOR<float / double>
// (EncodingOptions::LOSSLESS, include/cloudini_lib/field_decoder.hpp: the value is the running XOR of the stream's words) and
// FieldDecoderCopy. Point i of a chunk lies at byte i * P of the stream (P = the sum of the field sizes), so nothing has to be
// found: one workgroup per chunk walks it in tiles of 1024 points, a thread loads its point's fields, the running XOR of every
// field is one DPP scan per wave + one exchange of the waves' totals per tile (double-buffered: one barrier), and the point leaves
// with one store per field. The stream kernel decoded these streams with its token machinery at 1.35 TB/s.
// At most kFxMaxOps fields; anything else (or a payload shorter than n * P) leaves the chunk to the kernels behind (kDecRedo).
// ---------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kFxMaxOps = 8;
constexpr uint32_t kFxThreads = 1024;

__global__ __launch_bounds__(kFxThreads) void k_decode_fixed(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                             const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                             uint32_t* __restrict__ reg_end, uint32_t* __restrict__ status,
                                                             uint32_t point_bytes) {
  __shared__ uint64_t wtot[2][kFxThreads / 64u][kFxMaxOps];
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DecChunk dc = chunks[c];
  if (!dc.valid) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t n = dc.n_points;
  const uint32_t n_ops = plan.n_ops;
  const uint64_t need = (uint64_t)n * point_bytes;
  const bool ok = n_ops <= kFxMaxOps && need <= dc.src_size && (plan.n_adaptive != 0u || need == dc.src_size);  // (uniform)
  if (!ok) {
    if (tid == 0) reg_end[c] = kDecRedo;
    return;
  }
  const uint32_t step = plan.point_step;
  uint8_t* base = out + (size_t)dc.first_point * step;
  // four 32-bit XOR fields back to back in a 16-byte point (lossless XYZI): one 16-byte load and store per point, 32-bit scans,
  // the waves' totals scanned by 16 lanes instead of read by everybody
  const bool quad = step == 16u && point_bytes == 16u && n_ops == 4u && plan.ops[0].kind == OP_XOR32 && plan.ops[1].kind == OP_XOR32 &&
                    plan.ops[2].kind == OP_XOR32 && plan.ops[3].kind == OP_XOR32 && plan.ops[0].offset == 0u && plan.ops[1].offset == 4u &&
                    plan.ops[2].offset == 8u && plan.ops[3].offset == 12u;  // (uniform)
  if (quad) {
    uint32_t* wt32 = reinterpret_cast<uint32_t*>(&wtot[0][0][0]);  // [2][16 waves][4]
    auto xscan = [](uint32_t x) __attribute__((always_inline)) {
#define FX_STEP(CTRL, RMASK, BC) x ^= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, RMASK, 0xf, BC);
      FX_STEP(0x111, 0xf, true)
      FX_STEP(0x112, 0xf, true)
      FX_STEP(0x114, 0xf, true)
      FX_STEP(0x118, 0xf, true)
      FX_STEP(0x142, 0xa, false)
      FX_STEP(0x143, 0xc, false)
#undef FX_STEP
      return x;
    };
    uint32_t r[4] = {0u, 0u, 0u, 0u};
    for (uint32_t t0 = 0, tile = 0; t0 < n; t0 += kFxThreads, ++tile) {
      const uint32_t i = t0 + tid;
      const bool have = i < n;
      uint4 q4 = make_uint4(0u, 0u, 0u, 0u);
      __builtin_memcpy(&q4, src + (size_t)(have ? i : 0u) * 16u, 16);
      uint32_t in[4] = {have ? q4.x : 0u, have ? q4.y : 0u, have ? q4.z : 0u, have ? q4.w : 0u};
      uint32_t inc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) inc[k] = xscan(in[k]);
      uint32_t* wt = wt32 + (tile & 1u) * 64u;
      if (lane == 63u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) wt[wave * 4u + k] = inc[k];
      }
      __syncthreads();
      uint32_t before[4], all[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t tw = lane < 16u ? wt[lane * 4u + k] : 0u;  // lane w: wave w's total
        const uint32_t sc = xscan(tw);                             // inclusive over the waves
        all[k] = (uint32_t)__builtin_amdgcn_readlane((int)sc, 15);
        const uint32_t ex = sc ^ tw;                               // exclusive
        before[k] = (uint32_t)__builtin_amdgcn_readlane((int)ex, (int)wave);
      }
      if (have) {
        const uint4 o = make_uint4(r[0] ^ before[0] ^ inc[0], r[1] ^ before[1] ^ inc[1], r[2] ^ before[2] ^ inc[2], r[3] ^ before[3] ^ inc[3]);
        __builtin_memcpy(base + (size_t)i * 16u, &o, 16);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) r[k] ^= all[k];
    }
    if (tid == 0) {
      reg_end[c] = (uint32_t)need;
      atomicAdd(&status[kStatFastRegular], 1u);
    }
    return;
  }
  uint64_t run[kFxMaxOps];  // running XOR of every field in front of the tile (uniform)
#pragma unroll
  for (uint32_t k = 0; k < kFxMaxOps; ++k) run[k] = 0ull;
  for (uint32_t t0 = 0, tile = 0; t0 < n; t0 += kFxThreads, ++tile) {
    const uint32_t i = t0 + tid;
    const bool have = i < n;
    const uint8_t* q = src + (size_t)(have ? i : 0u) * point_bytes;
    uint64_t v[kFxMaxOps], incl[kFxMaxOps];
    uint32_t at = 0u;
#pragma unroll
    for (uint32_t k = 0; k < kFxMaxOps; ++k) {
      v[k] = 0ull;
      if (k < n_ops) {  // (uniform)
        const uint32_t size = plan.ops[k].size;
        uint64_t x = 0ull;
        if (size == 4u) {
          uint32_t w;
          __builtin_memcpy(&w, q + at, 4);
          x = w;
        } else if (size == 8u) {
          __builtin_memcpy(&x, q + at, 8);
        } else {
          for (uint32_t b = 0; b < size; ++b) x |= (uint64_t)q[at + b] << (8u * b);
        }
        v[k] = have ? x : 0ull;
        at += size;
      }
    }
    const uint32_t buf = tile & 1u;
#pragma unroll
    for (uint32_t k = 0; k < kFxMaxOps; ++k) {
      incl[k] = 0ull;
      if (k < n_ops && plan.ops[k].kind != OP_COPY) {
        incl[k] = sw_scan64<true>(v[k]);
        if (lane == 63u) wtot[buf][wave][k] = incl[k];
      }
    }
    __syncthreads();  // (the other buffer is read by nobody any more: the previous tile's readers passed the barrier before it)
#pragma unroll
    for (uint32_t k = 0; k < kFxMaxOps; ++k) {
      if (k < n_ops) {
        const uint32_t off = plan.ops[k].offset;
        const uint32_t size = plan.ops[k].size;
        uint64_t val = v[k];
        if (plan.ops[k].kind != OP_COPY) {
          uint64_t before = run[k], all = 0ull;
#pragma unroll
          for (uint32_t w = 0; w < kFxThreads / 64u; ++w) {
            const uint64_t x = wtot[buf][w][k];
            before ^= w < wave ? x : 0ull;
            all ^= x;
          }
          val = before ^ incl[k];
          run[k] ^= all;
        }
        if (have && off != 0xffffffffu) st_raw(base + (size_t)i * step + off, val, size);
      }
    }
  }
  if (tid == 0) {
    reg_end[c] = (uint32_t)need;
    atomicAdd(&status[kStatFastRegular], 1u);
  }
}

}  // namespace cldn
