// stage1_decode_dv.h -- k_section_dv_w (round 6): the lone DeltaVarint section of a chunk -> a dense column, with the point
// decoder's machinery (stage1_decode_wave.h) instead of k_sections_cols_fast's slices (a serial token walk per thread and three
// block scans per 4 KiB). decodeV5AdaptiveIntSection, mode 0 (src/v5_codec.cpp:764-879): [0x00] then one varint per point, the
// difference to the value before it (int64 arithmetic in the reference; a field of at most 4 bytes keeps only the low 32 bits,
// so wrapping 32-bit sums are exact).
//
// One workgroup of 16 waves per chunk; the section is cut into pieces of 62 units, one wave per piece, round robin, no barrier:
//   phase V   token values byte-parallel into LDS slots (wp_scatter16) -- here a token IS a "point" with one lane
//   chain 1   tokens in front of the piece = index of its first value in the column
//   rows      64 tokens per row, one per lane: zigzag, one DPP prefix sum
//   chain 2   the running value in front of the piece
//   stores    col[T0 + j] = carry + sum: 64 consecutive values per store instruction
// Tokens of 5 bytes (differences of 2^27 and more: the first value of an rgba field, say) take a side path: the lanes that hold
// one compute the difference's low 32 bits from the bytes themselves, overwrite the token's slot with it and mark the slot in a
// bit mask of the piece; the rows read the mask only in pieces that have such a token.
// Handed back to the kernels behind it (sec_cols[c] stays 0; they decode the chunk or raise the errors): a token of more than 5
// bytes, a 0x00 byte where a token ends (the marker byte is no integer token: decodeVarint rejects it; padded forms are legal
// but the encoder never writes them), token counts that do not match the points, a last byte that ends no token.
// Where the section begins comes from k_locate_sections (reg_end_pre; the section's mode byte in slices_done[c] >> 24).
#pragma once

namespace cldn {

constexpr uint32_t kDvWaves = 16u;
constexpr uint32_t kDvRows = (kWpPiece + 63u) / 64u;  // a piece of one-byte tokens: 992 of them

struct DvwLds {
  static constexpr uint32_t kValsOff = 0u;
  static constexpr uint32_t kDummyOff = kDvWaves * kWpSlots * 4u;
  static constexpr uint32_t kTrecOff = kDummyOff + 256u;          // u64 [kWpRing]: tokens up to and including a piece
  static constexpr uint32_t kVrecOff = kTrecOff + kWpRing * 8u;   // u64 [kWpRing]: running value behind a piece
  static constexpr uint32_t kMaskOff = kVrecOff + kWpRing * 8u;   // u32 [kDvWaves][32]: slots that hold a 5-byte token's difference
  static constexpr uint32_t kMiscOff = kMaskOff + kDvWaves * 128u;  // [0] irregular, [1] a wait gave up, [2] tokens of the section
  static constexpr uint32_t kTotal = kMiscOff + 64u;
};

__global__ __launch_bounds__(kDvWaves * 64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_section_dv_w(
    const DevPlan plan, const uint8_t* __restrict__ streams, const DecChunk* __restrict__ chunks, uint8_t* __restrict__ col0,
    const uint32_t* __restrict__ reg_end_pre, uint8_t* __restrict__ sec_cols, const uint32_t* __restrict__ slices_done,
    uint32_t* __restrict__ status) {
  using L = DvwLds;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  unsigned long long* trec = reinterpret_cast<unsigned long long*>(smem + L::kTrecOff);
  unsigned long long* vrec = reinterpret_cast<unsigned long long*>(smem + L::kVrecOff);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + L::kMiscOff);
  uint32_t* lmask_all = reinterpret_cast<uint32_t*>(smem + L::kMaskOff);
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((slices_done[c] >> 24) != 0u) return;  // (uniform) no DeltaVarint section: the kernels behind take the chunk
  const DecChunk dc = chunks[c];
  if (!dc.valid || plan.n_adaptive != 1u || plan.adaptive[0].bpv > 4u) return;
  const uint32_t off = reg_end_pre[c];
  const uint32_t n = dc.n_points;
  if (off == 0xffffffffu || off >= dc.src_size || n == 0u) return;
  const uint8_t* src = streams + dc.src_off + off + 1u;  // behind the mode byte
  const uint32_t src_size = dc.src_size - off - 1u;
  if (src_size < n || (src_size != 0u && (src[src_size - 1u] & 0x80u) != 0u)) return;  // fewer bytes than tokens; the last byte ends none
  const uint32_t bpv = plan.adaptive[0].bpv;
  uint8_t* col = col0 + (size_t)dc.first_point * bpv;

  if (tid < 3u) misc[tid] = 0u;
  for (uint32_t i = tid; i < kWpRing * 4u; i += kDvWaves * 64u) reinterpret_cast<uint32_t*>(trec)[i] = 0u;  // tags: none
  __syncthreads();

  const uint32_t a0 = (uint32_t)((uintptr_t)src & 15u);
  const uint8_t* src_al = src - a0;
  const uint32_t vend = a0 + src_size;
  const uint32_t n_pieces = (vend + kWpPiece - 1u) / kWpPiece;
  const uint32_t smem_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  const uint32_t vals_lds = smem_lds + L::kValsOff + wave * (kWpSlots * 4u);
  const uint32_t* vals = reinterpret_cast<const uint32_t*>(smem + L::kValsOff + wave * (kWpSlots * 4u));
  const uint32_t dummy_lds = smem_lds + L::kDummyOff + lane * 4u;
  uint32_t* lmask = lmask_all + wave * 32u;
  uint32_t* vals_w = reinterpret_cast<uint32_t*>(smem + L::kValsOff + wave * (kWpSlots * 4u));

  // (the bytes in front of the section belong to the payload -- the mode byte, the regular stream -- and the unit behind its
  // end may lie behind the payload: a unit is read only where it holds a byte of the section)
  auto load_unit = [&](uint32_t v0, uint32_t(&u)[4]) __attribute__((always_inline)) {
    const bool ok = v0 < vend;
    const uint4 w = *reinterpret_cast<const uint4*>(src_al + (ok ? v0 : 0u));
    u[0] = ok ? w.x : 0xffffffffu;
    u[1] = ok ? w.y : 0xffffffffu;
    u[2] = ok ? w.z : 0xffffffffu;
    u[3] = ok ? w.w : 0xffffffffu;
  };
  uint32_t b[4];
  uint32_t p = wave;
  load_unit(min(p, n_pieces) * kWpPiece + lane * 16u - 16u, b);
  __builtin_amdgcn_s_waitcnt(0);
  bool gave_up = false;
  bool irregular = false;
  __builtin_amdgcn_s_setprio(1);
  for (; p < n_pieces; p += kDvWaves) {
    const uint32_t v0 = p * kWpPiece + lane * 16u - 16u;
    if (p == 0u) {  // uniform. What lies in front of the section: token ends, value bits 0
      if (lane == 0u) b[0] = b[1] = b[2] = b[3] = 0u;
      if (lane == 1u) {
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) b[k] = a0 >= 4u * k + 4u ? 0u : (a0 > 4u * k ? b[k] & (0xffffffffu << (8u * (a0 - 4u * k))) : b[k]);
      }
    }
    const uint32_t eraw = wp_ends16(b);
    uint32_t ev = (lane != 0u && lane != 63u) ? eraw : 0u;  // lanes 1..62 own the piece's units (no halo: a token is a whole "point")
    if (p == 0u || p * kWpPiece + 1008u > vend) {  // uniform: the section's first / last piece
      const uint32_t lo = p == 0u && lane == 1u ? a0 : 0u;
      const uint32_t hi = lane != 0u && vend > v0 ? min(vend - v0, 16u) : 0u;
      ev &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    }
    const uint32_t cl = (uint32_t)__builtin_popcount(ev);
    const uint32_t incl = wave_inclusive_scan(cl);
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t tb = incl - cl;
    const uint32_t below = wp_from_lane_below(eraw);
    const uint32_t e20 = (below >> 12) | (eraw << 4);
    // tokens of 5 bytes and of more: an end behind four / five bytes that continue (e24: bit 8 + j = byte j, bits 0..7 the bytes in front)
    uint32_t long5;
    {
      const uint32_t e24 = (below >> 8) | (eraw << 8);
      const uint32_t c24 = ~e24;
      const uint32_t run2 = c24 & (c24 << 1);
      const uint32_t run4 = run2 & (run2 << 2);
      const uint32_t five_up = e24 & (run4 << 1);             // bit i: an end with bytes i - 4 .. i - 1 continuing
      const uint32_t six_up = five_up & (c24 << 5);           // ... and byte i - 5 too
      long5 = ((five_up & ~six_up) >> 8) & ev;
      const uint32_t zero_end = (~wp_nonzero16(b) & 0xffffu) << 8;
      irregular = irregular || ((((six_up | (zero_end & (c24 << 1))) >> 8) & ev) != 0u);
    }
    const bool has_long = __ballot(long5 != 0u) != 0ull;  // uniform
    {
      uint32_t pk[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) pk[k] = wp_pack7(b[k]);
      wp_scatter16(pk, e20, ev, vals_lds + tb * 4u, dummy_lds);
    }
    if (has_long) {  // (rare) the 5-byte tokens' differences from the bytes themselves: low 32 bits, over the token's slot
      if (lane < 32u) lmask[lane] = 0u;
      wp_wave_sync();
      const uint32_t front = wp_from_lane_below(b[3]);  // bytes -4 .. -1
      for (uint32_t m = long5; m; m &= m - 1u) {
        const uint32_t i = (uint32_t)__builtin_ctz(m);  // the token's bytes: i - 4 .. i
        // window of 8 bytes that begins at byte i - 4 (i.e. at byte i of the 20-byte run front, b[0..3])
        const uint32_t q = i >> 2, sh = (i & 3u) * 8u;
        const uint32_t w0 = q == 0u ? front : (q == 1u ? b[0] : (q == 2u ? b[1] : b[2]));
        const uint32_t w1 = q == 0u ? b[0] : (q == 1u ? b[1] : (q == 2u ? b[2] : b[3]));
        const uint32_t w2 = q == 0u ? b[1] : (q == 1u ? b[2] : (q == 2u ? b[3] : 0u));
        const uint32_t lo4 = sh ? ((w0 >> sh) | (w1 << (32u - sh))) : w0;             // bytes 0..3 of the token
        const uint32_t b4 = (sh ? ((w1 >> sh) | (w2 << (32u - sh))) : w1) & 0x7fu;  // byte 4 (it ends the token)
        const uint64_t u = (uint64_t)wp_pack7(lo4) | ((uint64_t)b4 << 28);
        const uint64_t u1 = u - 1ull;
        const uint32_t d = (uint32_t)((u1 >> 1) ^ (0ull - (u1 & 1ull)));
        const uint32_t slot = tb + (uint32_t)__builtin_popcount(ev & ((1u << i) - 1u));
        vals_w[slot] = d;
        atomicOr(&lmask[slot >> 5], 1u << (slot & 31u));
      }
    }
    load_unit(min(p + kDvWaves, n_pieces) * kWpPiece + lane * 16u - 16u, b);  // my next piece's bytes
    // ---- chain 1: tokens in front of the piece
    uint32_t T0 = 0u;
    if (p != 0u) {
      const unsigned long long* r = trec + ((p - 1u) & (kWpRing - 1u));
      unsigned long long x = wp_rec_load(r);
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) != p) {
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t spins = 1u;; ++spins) {
          __builtin_amdgcn_s_sleep(kWpSleep);
          x = wp_rec_load(r);
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) == p) break;
          if ((spins & 63u) == 0u && (spins >= kWpSpinLimit || __hip_atomic_load(&misc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u)) {
            gave_up = true;
            break;
          }
        }
        __builtin_amdgcn_s_setprio(1);
      }
      T0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    }
    if (gave_up) break;  // uniform
    if (lane == 0u) wp_rec_store(trec + (p & (kWpRing - 1u)), ((unsigned long long)(p + 1u) << 32) | (T0 + cnt));
    if (T0 + cnt > n) {  // (uniform) more tokens than points: nobody needs to wait for this piece's value any more
      irregular = true;
      if (lane == 0u) __hip_atomic_store(&misc[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      break;
    }
    if (p + 1u == n_pieces && lane == 0u) misc[2] = T0 + cnt;
    wp_wave_sync();
    // ---- rows of 64 tokens: values relative to the piece's start
    uint32_t val[kDvRows];
    uint32_t bs = 0u;
#pragma unroll
    for (uint32_t r = 0; r < kDvRows; ++r) {
      val[r] = 0u;
      if (r * 64u < cnt) {  // uniform
        const uint32_t j = r * 64u + lane;
        const bool have = j < cnt;
        const uint32_t u = vals[have ? j : 0u];
        const uint32_t u1 = u - 1u;
        uint32_t d = (u1 >> 1) ^ (0u - (u1 & 1u));
        bool is_long = false;
        if (has_long) {  // uniform
          is_long = ((lmask[j >> 5] >> (j & 31u)) & 1u) != 0u;
          d = is_long ? u : d;
        }
        irregular = irregular || (have && !is_long && u == 0u);
        const uint32_t inc = wave_inclusive_scan(have ? d : 0u);
        val[r] = bs + inc;
        bs += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      }
    }
    // ---- chain 2: the value in front of the piece
    uint32_t carry = 0u;
    if (p != 0u) {
      const unsigned long long* r = vrec + ((p - 1u) & (kWpRing - 1u));
      unsigned long long x = wp_rec_load(r);
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) != p) {
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t spins = 1u;; ++spins) {
          __builtin_amdgcn_s_sleep(kWpSleep);
          x = wp_rec_load(r);
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) == p) break;
          if ((spins & 63u) == 0u && (spins >= kWpSpinLimit || __hip_atomic_load(&misc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u)) {
            gave_up = true;
            break;
          }
        }
        __builtin_amdgcn_s_setprio(1);
      }
      carry = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
    }
    if (gave_up) break;  // uniform
    if (lane == 0u) wp_rec_store(vrec + (p & (kWpRing - 1u)), ((unsigned long long)(p + 1u) << 32) | (carry + bs));
    asm volatile("" ::"v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));  // (the wait for my next piece's bytes in front of the stores)
    // ---- stores: 64 consecutive values per instruction
#pragma unroll
    for (uint32_t r = 0; r < kDvRows; ++r) {
      if (r * 64u < cnt) {  // uniform
        const uint32_t j = r * 64u + lane;
        if (j < cnt) {
          const uint32_t v = carry + val[r];
          if (bpv == 2u) reinterpret_cast<uint16_t*>(col)[T0 + j] = (uint16_t)v;
          else reinterpret_cast<uint32_t*>(col)[T0 + j] = v;
        }
      }
    }
    wp_wave_sync();  // the next piece's values overwrite this one's slots
  }
  if (gave_up && lane == 0u) __hip_atomic_store(&misc[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (__ballot(irregular) != 0ull && lane == 0u) misc[0] = 1u;
  __syncthreads();
  if (tid == 0u && misc[0] == 0u && __hip_atomic_load(&misc[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u && misc[2] == n) {
    sec_cols[c] = 1u;  // every point has its value: the chunk is done
    atomicAdd(&status[kStatDvChunks], 1u);
  }
}

}  // namespace cldn
