// stage1_decode_fast.h -- k_decode_points: stage-1 decode of chunks whose regular stream is one fused FloatN encoder
// (3 or 4 int32-delta varint tokens per point; FieldDecoderFloatN_Lossy::decode, src/field_decoder.cpp:43-86), with
// the chunk's Palette sections (decodeV5AdaptiveIntSection, src/v5_codec.cpp:764-879, mode 1) folded into the same
// pass. Included by stage1_kernels.hip behind stage1_decode.h.
//
// k_decode_varint walks the 16 byte positions of every thread with the full token logic under a predicate (233
// lane-instructions per token); here the two jobs are separated:
//   phase A  byte-parallel and cheap: every thread flags the token ends in its 16 bytes, one block scan numbers them,
//            and the byte behind every NOPS-th end -- where a point starts -- goes into an LDS list in point order;
//   phase B  point-parallel, every lane busy: a thread takes 3 consecutive points; for each it reads the 4 * (NOPS + 1)
//            bytes behind the point's start from the LDS copy of the tile and walks the NOPS tokens in registers (a
//            token's length is the position of its first byte with a clear MSB; the window is shifted by it), sums
//            the deltas, a segmented block scan (NaN markers reset a lane) turns them into values, and the values leave
//            through an LDS transposition so that a store instruction covers consecutive points.
// A tile starts exactly at a point boundary (the next tile begins behind the last token it consumed), so no point is
// ever cut and nothing but the per-lane running values is carried between tiles.
//
// Sections: where the regular stream ends is not written anywhere. With one adaptive field the place is guessed from
// the end of the payload (a Palette section's size is a function of its entry count: all counts are tried, the guess is
// checked against where the tiles really end); with two, a counting pre-pass over the token-end flags finds it. If every section of the chunk is a Palette with at most kFastPalEntries entries (intensity, ring,
// reflectivity ... of real lidars), their tables go to LDS and every point is completed when it is written: x, y, z
// and its integer fields together, so that each output line is written once by one workgroup (the separate section
// kernel re-dirtied every 64-byte line: WRITE_SIZE was 1.97x the output). Other sections are left to
// k_decode_sections / k_decode_general exactly as before, and so is every chunk this kernel finds irregular (tokens of
// more than 5 bytes, a stream that ends early, an index beyond its palette): reg_end[c] = kDecRedo hands it to
// k_decode_varint, which keeps the reference's error reporting.
#pragma once

namespace cldn {

constexpr uint32_t kFpPPT = 3;                          // points per thread and tile
constexpr uint32_t kFpThreads = 512;                    // 8 waves; four workgroups share a CU
constexpr uint32_t kFpTileBytes = kFpThreads * 16u;     // 8 KiB of stream per tile
constexpr uint32_t kFpTilePoints = kFpThreads * kFpPPT; // at most 1536 points leave per tile
constexpr uint32_t kFastPalEntries = 1024;
constexpr uint32_t kFastPalFields = 2;

// NF = Palette sections the instantiation can fold (0, 1, 2 = min(adaptive fields of the plan, kFastPalFields)): the
// staging area and the palette tables are sized for it, so that XYZ(I) clouds with one integer field keep four
// workgroups per CU (the 4-lane layout with room for two folded fields needs 46 KB and gets three)
template <int NOPS, int NF = 2>
struct FpLds {
  static constexpr uint32_t kTileOff = 0;                                   // [16 zero bytes][tile][32 pad]
  static constexpr uint32_t kPosOff = 16u + kFpTileBytes + 32u;             // u16 [kFpTilePoints + 1 + 7]: where point q starts
  static constexpr uint32_t kPosEntries = 1u + kFpTilePoints + 7u;
  static constexpr uint32_t kWorkEnd = (kPosOff + kPosEntries * 2u + 15u) & ~15u;
  static constexpr uint32_t kStageBytes = kFpTilePoints * (NOPS + NF) * 4u;  // decoded points (floats + folded
                                                                                         // fields), overlays tile + list
  static constexpr uint32_t kScanOff = (kWorkEnd > kStageBytes ? kWorkEnd : kStageBytes);
  static constexpr uint32_t kWaveRec = NOPS * 4u + 8u;                      // per wave: int[NOPS] + flags; [16], [17] = carry
  static constexpr uint32_t kPalOff = (kScanOff + 18u * kWaveRec + 15u) & ~15u;     // (read and written in turns)
  static constexpr uint32_t kMiscOff = kPalOff + (uint32_t)NF * kFastPalEntries * 4u;
  static constexpr uint32_t kTotal = kMiscOff + 512u;
};

// 16 payload bytes at payload offset `o` (any alignment of the stream); bytes behind the payload read as 0xff
// (continuation bytes: no token ends there)
__device__ __forceinline__ void fp_load16(const uint8_t* __restrict__ src, uint32_t src_size, uint32_t o, uint32_t (&b)[4]) {
  if (o + 16u <= src_size) {
    const uint8_t* q = src + o;
    const uint32_t mis = (uint32_t)((uintptr_t)q & 3u);
    const uint32_t* a = reinterpret_cast<const uint32_t*>(q - mis);
    // 5 aligned dwords cover the 16 bytes; the fifth is only read when the 16 bytes really reach into it
    const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3];
    const uint32_t d4 = mis ? a[4] : 0u;
    b[0] = __builtin_amdgcn_alignbyte(d1, d0, mis);
    b[1] = __builtin_amdgcn_alignbyte(d2, d1, mis);
    b[2] = __builtin_amdgcn_alignbyte(d3, d2, mis);
    b[3] = __builtin_amdgcn_alignbyte(d4, d3, mis);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t w = 0xffffffffu;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const uint32_t i = o + 4u * k + bb;
        if (i < src_size) w = (w & ~(0xffu << (8 * bb))) | ((uint32_t)src[i] << (8 * bb));
      }
      b[k] = w;
    }
  }
}

// the same through ONE unaligned 16-byte load (gfx950 performs it natively): for streaming passes that only count
__device__ __forceinline__ void fp_load16u(const uint8_t* __restrict__ src, uint32_t src_size, uint32_t o, uint32_t (&b)[4]) {
  if (o + 16u <= src_size) {
    uint4 w;
    __builtin_memcpy(&w, src + o, 16);
    b[0] = w.x; b[1] = w.y; b[2] = w.z; b[3] = w.w;
  } else {
    fp_load16(src, src_size, o, b);
  }
}

// bit j = byte j of the 16 ends a token (MSB clear)
__device__ __forceinline__ uint32_t fp_ends16(const uint32_t (&b)[4]) {
  uint32_t ends = 0u;
#pragma unroll
  for (int k = 0; k < 4; ++k) ends |= ((((~b[k] & 0x80808080u) >> 7) * 0x00204081u) >> 21 & 0xfu) << (4 * k);
  return ends;
}

// Where a lone Palette section would begin, found from the END of the payload: a Palette section of U entries has
// 3 + U * bpv + ceil(bits(U) * n / 8) bytes (src/v5_codec.cpp:298-306), so every U has exactly one place where its header
// [1][U lo][U hi] would have to be, and the threads try all U <= kFastPalEntries. Three bytes of a token stream that merely
// look like a header are no rarity -- 2 to 6 % of the chunks of a lidar batch have such a place -- and every chunk that
// follows a wrong guess ends in the serial section decoder (150 us for the whole launch). So the smallest candidate is
// checked by the workgroup together (one independent load or two per thread, one round trip): the byte in front of it
// ends a token, the first 8 entries are distinct, the first 32 indexes are below U. A real section always passes (the
// encoder's entries are distinct, its indexes in range); a look-alike passes about once in 10^4. A candidate that fails
// is skipped and the next one tried; after four of them the caller counts tokens instead.
// All threads of the workgroup call it (barriers inside); sh = two LDS words. Returns U, or 0xffffffff.
template <int T>
__device__ __forceinline__ uint32_t pal_guess_from_end(const uint8_t* __restrict__ src, uint32_t src_size, uint32_t n, uint32_t bpv,
                                                       uint32_t* sh) {
  const uint32_t tid = threadIdx.x;
  uint32_t floor = 0u;
  for (uint32_t round = 0; round < 4u; ++round) {
    if (tid == 0u) {
      sh[0] = 0xffffffffu;
      sh[1] = 0u;
    }
    __syncthreads();
    for (uint32_t U = tid + 1u; U <= kFastPalEntries; U += (uint32_t)T) {
      const uint64_t S = 3ull + (uint64_t)U * bpv + ((uint64_t)palette_bits(U) * n + 7u) / 8u;
      if (U > floor && S <= src_size) {
        const uint8_t* h = src + (src_size - (uint32_t)S);
        if (h[0] == 1u && ((uint32_t)h[1] | ((uint32_t)h[2] << 8)) == U) atomicMin(&sh[0], U);
      }
    }
    __syncthreads();
    const uint32_t U = sh[0];
    if (U == 0xffffffffu) return U;
    const uint32_t bits = palette_bits(U);
    const uint32_t off = src_size - (uint32_t)(3ull + (uint64_t)U * bpv + ((uint64_t)bits * n + 7u) / 8u);
    const uint8_t* tab = src + off + 3u;
    bool bad = false;
    if (tid == 0u) {
      bad = off != 0u && (src[off - 1u] & 0x80u) != 0u;  // the token in front of the section would not be over
    } else if (tid <= 32u) {
      const uint32_t k = tid - 1u;  // index k of the section
      if (bits != 0u && k < n) {
        const uint8_t* idx = tab + (size_t)U * bpv;
        const uint32_t avail = src_size - (off + 3u + U * bpv);
        const uint32_t bit0 = k * bits, by = bit0 >> 3;
        uint32_t w = 0u;
        for (uint32_t b = 0; b < 3u; ++b)  // bits <= 10: three bytes hold an index
          if (by + b < avail) w |= (uint32_t)idx[by + b] << (8u * b);
        bad = ((w >> (bit0 & 7u)) & ((1u << bits) - 1u)) >= U;
      }
    } else if (tid < 41u) {
      const uint32_t i = tid - 33u;  // entry i against the entries in front of it
      if (i >= 1u && i < U) {
        uint32_t v = 0u;
        for (uint32_t b = 0; b < bpv; ++b) v |= (uint32_t)tab[i * bpv + b] << (8u * b);
        for (uint32_t j = 0; j < i; ++j) {
          uint32_t w = 0u;
          for (uint32_t b = 0; b < bpv; ++b) w |= (uint32_t)tab[j * bpv + b] << (8u * b);
          bad = bad || w == v;
        }
      }
    }
    if (bad) sh[1] = 1u;
    __syncthreads();
    const bool rejected = sh[1] != 0u;
    __syncthreads();  // (sh is written again)
    if (!rejected) return U;
    floor = U;
  }
  return 0xffffffffu;
}

struct DecColumns {   // dense columns of up to 8 adaptive fields (value i of the batch at p[a] + i * bpv)
  const uint8_t* p[8];
};

struct FpSection {   // a Palette section folded into the point pass
  uint32_t field_off;  // offset of the field inside the point
  uint32_t bpv;        // 2 or 4
  uint32_t count;      // palette entries
  uint32_t bits;       // bits per index
  uint32_t index_off;  // payload offset of the packed indexes
};

// ---------------------------------------------------------------------------------------------------------------
// k_locate_sections: where a chunk's sections begin = behind token number n_points * n_ops of its payload. The token
// ends (bytes with a clear MSB) are counted by NW waves (4, or 16 when a batch has few chunks), each over its share of the payload, 64 bytes per lane and
// step; the wave that holds the last token walks its quarter again and finds the byte. A light kernel (no LDS to speak
// of, eight workgroups per CU): the count is a plain streaming read of the regular stream.
// reg_end_pre[c] = the offset, 0xffffffff = not found / not looked for (one-field plans whose section looks like a small
// Palette from the end of the payload are left to k_decode_points' own guess). grid = n_chunks, NW * 64 threads.
// ---------------------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_locate_sections(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                         const DecChunk* __restrict__ chunks, uint32_t n_ops,
                                                         uint32_t* __restrict__ reg_end_pre, uint8_t* __restrict__ sec_cols,
                                                         uint32_t* __restrict__ slices_done, uint32_t keep_guess, uint32_t try_dv,
                                                         uint32_t* __restrict__ status) {
  __shared__ uint32_t wcnt[NW];
  __shared__ uint32_t found, pal_sh[2];
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DecChunk dc = chunks[c];
  if (tid == 0) {
    reg_end_pre[c] = 0xffffffffu;
    sec_cols[c] = 0u;
    slices_done[c] = 0xffu << 24;
    found = 0xffffffffu;
  }
  if (!dc.valid || plan.n_adaptive == 0u || plan.n_adaptive > 8u) return;  // (more than kFastPalFields: the columns of stage1_decode_sections_w.h)
  for (uint32_t a = 0; a < plan.n_adaptive; ++a)
    if (plan.adaptive[a].bpv > 4u) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  const uint32_t target = n * n_ops;
  __syncthreads();
  if (plan.n_adaptive == 1u) {
    {
      const uint32_t U = pal_guess_from_end<NW * 64>(src, src_size, n, plan.adaptive[0].bpv, pal_sh);  // uniform
      if (U != 0xffffffffu) {
        // k_decode_points_w makes this guess again itself; the stream kernel's column flow (keep_guess) wants the place
        if (keep_guess && tid == 0) {
          reg_end_pre[c] = src_size - (3u + U * plan.adaptive[0].bpv + (palette_bits(U) * n + 7u) / 8u);
          slices_done[c] = 1u << 24;
        }
        return;
      }
    }
    // A lone DeltaRle section (ring / line index of a lidar) is also found from the end: [3][u32 runs][runs x (varint
    // difference, varint length)] -- every position p of the payload's last bytes with src[p] == 3, a run count r in
    // 1..n behind it, a token end in front of it and exactly 2 r token ends between p + 5 and the payload's end is a
    // candidate; one candidate = the section (a regular token stream does not hold a 3 followed by a count whose upper
    // bytes are 0: zeros are NaN markers). Spares the count over the whole payload (C4: 274 MB per batch); like the
    // Palette guess it is verified by where k_decode_points' tiles end.
    constexpr uint32_t T = NW * 64u, W = T * 16u;
    __shared__ __attribute__((aligned(16))) uint32_t win[W / 4u + 4u];
    __shared__ uint32_t end_mask[T], end_pre[T], scan_tmp[32], cand[2];
    const uint32_t wbase = src_size > W ? src_size - W : 0u;  // payload offset of the window
    uint32_t b[4];
    fp_load16u(src, src_size, wbase + tid * 16u, b);           // (bytes behind the payload read 0xff: no ends)
    *reinterpret_cast<uint4*>(win + tid * 4u) = make_uint4(b[0], b[1], b[2], b[3]);
    if (tid < 4u) win[W / 4u + tid] = 0xffffffffu;
    if (tid == 0u) cand[0] = 0xffffffffu, cand[1] = 0u;
    const uint32_t ends = fp_ends16(b);
    uint32_t total_ends;
    const uint32_t pre = block_exclusive_scan<(int)T>((uint32_t)__builtin_popcount(ends), scan_tmp, &total_ends);  // barrier inside
    end_mask[tid] = ends;
    end_pre[tid] = pre;
    __syncthreads();
    const bool closed = src_size != 0u && (src[src_size - 1u] & 0x80u) == 0u;  // the payload ends with a token end
    if (closed) {
      const uint8_t* wb = reinterpret_cast<const uint8_t*>(win);
      for (uint32_t j = 0; j < 16u; ++j) {
        const uint32_t x = tid * 16u + j;  // window offset
        const uint32_t pos = wbase + x;
        if (pos + 5u > src_size || ((b[j >> 2] >> (8u * (j & 3u))) & 0xffu) != 3u) continue;
        const bool end_in_front = pos == 0u || (x != 0u ? (wb[x - 1u] & 0x80u) == 0u : (src[pos - 1u] & 0x80u) == 0u);
        if (!end_in_front) continue;
        const uint32_t r = (uint32_t)wb[x + 1u] | ((uint32_t)wb[x + 2u] << 8) | ((uint32_t)wb[x + 3u] << 16) | ((uint32_t)wb[x + 4u] << 24);
        if (r == 0u || r > n) continue;
        const uint32_t y = x + 5u;  // ends in [y, end of the window) -- y may be the window's end
        const uint32_t before_y = y >= W ? total_ends : end_pre[y >> 4] + (uint32_t)__builtin_popcount(end_mask[y >> 4] & ((1u << (y & 15u)) - 1u));
        if (total_ends - before_y != 2u * r) continue;
        atomicMin(&cand[0], pos);
        atomicAdd(&cand[1], 1u);
      }
    }
    __syncthreads();
    if (cand[1] == 1u) {  // uniform
      if (tid == 0) {
        reg_end_pre[c] = cand[0];
        slices_done[c] = 3u << 24;
      }
      return;
    }
    // A lone DeltaVarint section (round 6; rgba of a depth camera) is found from the end too: [0][n varints] closes the
    // payload, so its mode byte is the (n + 1)-th byte with a clear MSB counted from the payload's end -- if that byte is
    // a 0 behind at least n * n_ops bytes of regular stream, that is the guess (C3: the section is a quarter of the payload,
    // the count from the front reads the other three). Like the guesses above it is verified by where the point kernel's
    // tokens end; a miss costs the tiles read here and goes on to the count below.
    if (closed && n_ops != 0u && try_dv != 0u) {
      __syncthreads();  // (cand is read above)
      if (tid == 0u) cand[0] = 0xffffffffu;
      constexpr uint32_t TILE = T * 64u;  // bytes per step: four units per thread, thread 0 the LAST 64 bytes of the tile
      uint32_t seen = 0u, hi = src_size;  // ends behind the tile; the tile is [hi - TILE, hi)   (uniform)
      while (hi >= TILE + 16u) {          // (a tile that would touch the payload's first bytes: no guess)
        const uint32_t mine = hi - (tid + 1u) * 64u;  // my 64 bytes
        uint32_t e4[4], cnt4 = 0u;
#pragma unroll
        for (uint32_t u = 0; u < 4u; ++u) {  // unit u = bytes [mine + 48 - 16 u, + 16): u = 0 is the last one
          uint4 w;
          __builtin_memcpy(&w, src + mine + 48u - 16u * u, 16);
          const uint32_t bb[4] = {w.x, w.y, w.z, w.w};
          e4[u] = fp_ends16(bb);
          cnt4 += (uint32_t)__builtin_popcount(e4[u]);
        }
        uint32_t in_tile;
        const uint32_t behind = block_exclusive_scan<(int)T>(cnt4, scan_tmp, &in_tile);  // ends of the tile behind my bytes (barriers inside)
        if (seen + in_tile >= n + 1u) {  // uniform: the wanted end lies in this tile
          const uint32_t want = n + 1u - seen;  // its rank counted from the tile's end, 1-based
          if (behind < want && want <= behind + cnt4) {
            uint32_t k = want - behind;
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
              const uint32_t cu = (uint32_t)__builtin_popcount(e4[u]);
              if (k != 0u && k <= cu) {
                uint32_t m = e4[u];
                for (; k > 1u; --k) m &= ~(0x80000000u >> __builtin_clz(m));  // drop the ends behind the wanted one
                cand[0] = mine + 48u - 16u * u + (31u - (uint32_t)__builtin_clz(m));
                k = 0u;
              } else if (k != 0u) {
                k -= cu;
              }
            }
          }
          break;
        }
        seen += in_tile;
        hi -= TILE;
      }
      __syncthreads();
      const uint32_t at = cand[0];
      if (at != 0xffffffffu && at >= n * n_ops && src[at] == 0u) {  // uniform
        if (tid == 0) {
          reg_end_pre[c] = at;
          slices_done[c] = 0u;  // (mode byte 0 in the top byte, no slice done)
          atomicAdd(&status[kStatDvMode], 1u);
        }
        return;
      }
    }
  }
  const uint32_t part = (((src_size + 15u) / 16u + (NW - 1u)) / NW) * 16u;  // bytes per wave, multiple of 16
  const uint32_t w0 = min(src_size, wave * part), w1 = min(src_size, w0 + part);
  uint32_t cnt = 0u;
  uint32_t step_cnt = 0u;  // lane s: my count after step s (steps 0..63; the walk below starts at the last such step)
  uint32_t it = 0u;
  // Steps of 4 KiB, four 16-byte units per lane. The loads carry no branch (a unit that begins behind my part reads my part's
  // first unit and counts nothing; the payload's last, partial unit is the one load that checks its bytes) and the NEXT step's
  // units are requested before this step's are counted: the loop used to wait for memory once per step (C3: 51 us for 102 MB).
  auto step_loads = [&](uint32_t o0, uint32_t (&b)[4][4]) __attribute__((always_inline)) {
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u) {
      const uint32_t o = o0 + u * 1024u + lane * 16u;
      const bool whole = o + 16u <= w1;      // (w1 <= src_size)
      uint4 w;
      __builtin_memcpy(&w, src + (whole ? o : w0), 16);  // w0 + 16 <= w1 whenever a whole unit exists; else the tail path below
      b[u][0] = whole ? w.x : 0xffffffffu;
      b[u][1] = whole ? w.y : 0xffffffffu;
      b[u][2] = whole ? w.z : 0xffffffffu;
      b[u][3] = whole ? w.w : 0xffffffffu;
    }
  };
  const bool any_whole = w0 + 16u <= w1;  // (uniform) my part holds at least one whole unit
  uint32_t nb[4][4];
  if (any_whole) step_loads(w0, nb);
  for (uint32_t o0 = w0; o0 < w1; o0 += 4096u, ++it) {
    uint32_t b[4][4];
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) b[u][k] = any_whole ? nb[u][k] : 0xffffffffu;
    if (any_whole && o0 + 4096u < w1) step_loads(o0 + 4096u, nb);  // (uniform)
    // the part's last, partial unit (only the payload's end is not a multiple of 16): its lane reads it byte-safe
    if (o0 + 4096u >= w1 && (w1 & 15u) != 0u) {  // (uniform) last step of a part that ends inside a unit
      const uint32_t ot = w0 + ((w1 - w0) & ~15u);  // the partial unit's offset
      const uint32_t rel = ot - o0;
      if (lane == ((rel & 1023u) >> 4)) {
        uint32_t t[4];
        fp_load16(src, src_size, ot, t);
        const uint32_t u = rel >> 10;
#pragma unroll
        for (uint32_t uu = 0; uu < 4u; ++uu)
#pragma unroll
          for (int k = 0; k < 4; ++k) b[uu][k] = uu == u ? t[k] : b[uu][k];
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < 4u; ++u)
#pragma unroll
      for (int k = 0; k < 4; ++k) cnt += (uint32_t)__builtin_popcount(~b[u][k] & 0x80808080u);
    const uint32_t upto = wave_sum(cnt);  // tokens of my part that end in steps 0..it
    if (lane == it) step_cnt = upto;
  }
  const uint32_t wsum = wave_sum(cnt);
  if (lane == 0u) wcnt[wave] = wsum;
  __syncthreads();
  uint32_t before = 0u;
  for (uint32_t w = 0; w < wave; ++w) before += wcnt[w];
  if (target != 0u && before < target && target <= before + wsum) {  // the last token ends in my part (one wave)
    // the 4 KiB step that holds it: the first one whose running count reaches the target (steps beyond 63: walk from 63)
    const uint32_t n_steps = min(it, 64u);
    const uint64_t short_of = __ballot(lane < n_steps && before + step_cnt < target);
    const uint32_t s0 = __builtin_amdgcn_readfirstlane((uint32_t)__builtin_popcountll(short_of));  // steps 0..s0-1 end before the target
    uint32_t seen = before + (s0 ? (uint32_t)__builtin_amdgcn_readlane((int)step_cnt, (int)(s0 - 1u)) : 0u);
    for (uint32_t o0 = w0 + s0 * 4096u; o0 < w1; o0 += 1024u) {
      const uint32_t o = o0 + lane * 16u;
      uint32_t b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      if (o < w1) fp_load16u(src, src_size, o, b);
      const uint32_t ends = fp_ends16(b);
      const uint32_t cl = (uint32_t)__builtin_popcount(ends);
      const uint32_t incl = wave_inclusive_scan(cl);
      const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
      if (seen + row >= target) {
        const uint32_t my_first = seen + incl - cl;  // tokens before mine
        if (my_first < target && target <= my_first + cl) {
          uint32_t m = ends;
          for (uint32_t k = my_first + 1u; k < target; ++k) m &= m - 1u;  // drop the ends before the wanted one
          found = o + (uint32_t)__builtin_ctz(m) + 1u;
        }
        break;
      }
      seen += row;
    }
  }
  if (target == 0u && tid == 0) found = 0u;
  __syncthreads();
  if (tid == 0) {
    reg_end_pre[c] = found;
    // the section's mode byte in the top byte: the workgroups of k_sections_cols_fast that have nothing to do leave on it
    slices_done[c] = (found < src_size ? (uint32_t)src[found] : 0xffu) << 24;
    if (found < src_size && src[found] == 0u) atomicAdd(&status[kStatDvMode], 1u);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_sections_cols_fast: the common case of k_decode_sections_cols below -- ONE adaptive field of 2 or 4 bytes whose section
// is DeltaVarint, Rle or DeltaRle (rgba of a depth camera, ring / timestamp of a lidar) -- with a quarter of the threads
// and a third of the LDS, so that every chunk of a batch is in flight at once. Output: the dense column col0.
//   DeltaVarint  slices of 4 KiB at fixed offsets, spread over the chunk's workgroups; a token belongs to the slice
//                that holds its last byte. Token ends (bytes with a clear MSB) -> a block scan numbers them; a thread
//                decodes the tokens that end in its 16 bytes (1..5 bytes: differences of 16/32-bit values have at
//                most 33 bits), a second scan (wrapping 32-bit arithmetic: only the low bytes of a value are stored)
//                and the (count, sum) records of the slices in front turn differences into values and their indexes.
//   DeltaRle     a section of at most 4 KiB / kScfMaxRuns runs: tokens numbered as above, pairs -> run table in LDS via two
//                block scans; a thread then fills 8 consecutive values (one search, then a walk along the table).
//   Rle          (few runs: a constant field) one lane parses the table, same fill.
// Anything else (a marker byte, a longer token, more runs, a size that does not add up) leaves sec_cols[c] = 0: the
// general kernel behind it, and in the end the serial decoder, take the chunk and raise the errors.
// k_locate_sections clears sec_cols[c] and leaves the section's mode byte in slices_done[c]. grid = n_chunks * parts
// (parts: 1..kScfMaxParts, more when the batch has few chunks), 256 threads.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kScfThreads = 256;
constexpr uint32_t kScfTileBytes = kScfThreads * 16u;
constexpr uint32_t kScfMaxRuns = 1024;  // (three tables of this many entries share the value buffer)
constexpr uint32_t kScfMaxParts = 16;   // workgroups per chunk (DeltaVarint slices round robin; runs: the first one): at most
constexpr uint32_t kScfMaxSlices = 48;  // 32768 tokens of 5 bytes are 40 slices
constexpr uint32_t kScfSpinLimit = 1u << 22;

__global__ __launch_bounds__(kScfThreads) void k_sections_cols_fast(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                    const DecChunk* __restrict__ chunks, uint8_t* __restrict__ col0,
                                                                    const uint32_t* __restrict__ reg_end_pre,
                                                                    uint8_t* __restrict__ sec_cols, uint32_t* __restrict__ slices_done,
                                                                    unsigned long long* __restrict__ slice_rec, uint32_t epoch, uint32_t parts) {
  constexpr int T = (int)kScfThreads;
  __shared__ __attribute__((aligned(16))) uint32_t tile[kScfTileBytes / 4u + 8u];  // 8 bytes of history, the slice's bytes, slack for 8-byte windows
  __shared__ uint16_t end_pos[kScfTileBytes + 8u];                                  // byte index of every token end, in order
  __shared__ uint32_t vals[kScfTileBytes];                                          // values of the tile / run table
  __shared__ uint32_t scan[40];
  __shared__ uint32_t flags[4];  // [0] irregular, [1] runs parsed, [2] end offset of the runs
  const uint32_t c = blockIdx.x / parts;
  const uint32_t part = blockIdx.x % parts;
  const uint32_t tid = threadIdx.x;
  {  // (the mode byte k_locate_sections left: a Palette section, or no section found, is nothing to share)
    const uint32_t mb = slices_done[c] >> 24;
    if (part != 0u && mb != 0u && mb != 2u && mb != 3u) return;
  }
  if (sec_cols[c]) return;  // k_section_dv_w (round 6) has decoded the chunk's section
  const DecChunk dc = chunks[c];
  if (!dc.valid || plan.n_adaptive != 1u || plan.adaptive[0].bpv > 4u) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  uint32_t off = reg_end_pre[c];
  if (off == 0xffffffffu || off >= src_size || n == 0u) return;
  const uint32_t bpv = plan.adaptive[0].bpv;
  uint8_t* col = col0 + (size_t)dc.first_point * bpv;
  const uint32_t mode = src[off];
  ++off;
  if (tid == 0) flags[0] = 0u;
  __syncthreads();

  if (mode == 0u) {
    // DeltaVarint: the section is cut into slices of 4 KiB at FIXED offsets; a token belongs to the slice its last byte
    // is in. The `parts` workgroups of a chunk take the slices round robin; how many tokens, and which sum of
    // differences, lie in front of a slice is published per slice (records tagged with the launch's epoch) and read
    // by the slices behind it -- the workgroups in front were dispatched earlier (lower block index).
    const uint32_t sec_bytes = src_size - off;
    const uint32_t n_slices = (sec_bytes + kScfTileBytes - 1u) / kScfTileBytes;
    if (n_slices == 0u || n_slices > kScfMaxSlices) return;
    unsigned long long* rec = slice_rec + (size_t)c * kScfMaxSlices * 2u;
    uint32_t pre_cnt = 0u, pre_sum = 0u, have = 0u;  // tokens / sum of the slices [0, have)  (uniform)
    for (uint32_t s = part; s < n_slices; s += parts) {
      const uint32_t base = off + s * kScfTileBytes;  // >= 1: the mode byte (0, "a token end") is in front of slice 0
      uint32_t b[4];
      fp_load16u(src, src_size, base + tid * 16u, b);  // bytes behind the payload read 0xff: no token ends
      *reinterpret_cast<uint4*>(tile + 2u + tid * 4u) = make_uint4(b[0], b[1], b[2], b[3]);
      if (tid == 0u) {  // the 8 bytes in front of the slice (a token has 5 at most); what is in front of the payload counts as ends
        uint32_t h[2] = {0u, 0u};
        if (base >= 8u) {
          __builtin_memcpy(h, src + base - 8u, 8);
        } else {
          for (uint32_t j = 8u - base; j < 8u; ++j) h[j >> 2] |= (uint32_t)src[base - 8u + j] << (8u * (j & 3u));
        }
        tile[0] = h[0];
        tile[1] = h[1];
      }
      if (tid < 4u) tile[2u + kScfTileBytes / 4u + tid] = 0xffffffffu;
      const uint32_t ends = fp_ends16(b);
      const uint32_t my_cnt = (uint32_t)__builtin_popcount(ends);
      uint32_t n_tile;
      const uint32_t tb = block_exclusive_scan<T>(my_cnt, scan, &n_tile);  // barrier inside: the tile is complete
      // my tokens end in my 16 bytes; the first one begins behind the last end among the 8 bytes in front of them
      uint32_t sum = 0u;
      bool bad = false;
      if (ends) {
        const uint32_t h0 = tile[tid * 4u], h1 = tile[tid * 4u + 1u];
        const uint32_t hist = ((((~h0 & 0x80808080u) >> 7) * 0x00204081u) >> 21 & 0xfu) | ((((~h1 & 0x80808080u) >> 7) * 0x00204081u) >> 21 & 0xfu) << 4;
        uint32_t start = hist ? 32u - (uint32_t)__builtin_clz(hist) : 0u;  // window index (0 = 8 bytes in front of mine)
        uint32_t k = tb;
        for (uint32_t m = ends; m; m &= m - 1u) {
          const uint32_t endw = 8u + (uint32_t)__builtin_ctz(m);
          const uint32_t tl = endw - start + 1u;
          const uint32_t bo = tid * 16u + start;  // byte offset inside `tile` (which begins with the 8 bytes of history)
          const uint32_t di = bo >> 2, sh = (bo & 3u) * 8u;
          const uint32_t w0 = tile[di], w1 = tile[di + 1u], w2 = tile[di + 2u];
          const uint32_t lo = sh ? ((w0 >> sh) | (w1 << (32u - sh))) : w0;             // bytes 0..3 of the token
          const uint32_t b4 = (sh ? ((w1 >> sh) | (w2 << (32u - sh))) : w1) & 0xffu;  // byte 4
          const uint32_t g = (lo & 0x7fu) | (((lo >> 8) & 0x7fu) << 7) | (((lo >> 16) & 0x7fu) << 14) | (((lo >> 24) & 0x7fu) << 21);
          // the groups above the token's length are other tokens' bytes: mask by length; u = zigzag(d) + 1 has 35 bits at most
          const uint32_t keep = tl >= 4u ? 0x0fffffffu : ((1u << (7u * tl)) - 1u);
          const uint64_t u = (uint64_t)(g & keep) | (tl == 5u ? ((uint64_t)(b4 & 0x7fu) << 28) : 0ull);
          bad = bad || tl > 5u || u == 0ull;  // the marker byte is no integer token (decodeVarint rejects it)
          const uint64_t u1 = u - 1ull;
          const uint32_t dv = (uint32_t)((u1 >> 1) ^ (0ull - (u1 & 1ull)));  // low 32 bits of the difference
          vals[k++] = dv;
          sum += dv;
          start = endw + 1u;
        }
      }
      if (bad) flags[0] = 1u;
      uint32_t tile_sum;
      const uint32_t before = block_exclusive_scan<T>(sum, scan + 20, &tile_sum);  // barrier inside (after every write of flags[0])
      const bool irregular = flags[0] != 0u;
      if (tid == 0u) {  // the slices behind wait for this
        const unsigned long long tag = (unsigned long long)epoch << 32;
        __hip_atomic_store(rec + s * 2u + 1u, tag | tile_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(rec + s * 2u, tag | (irregular ? 0xffffffffu : n_tile), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (irregular) return;  // uniform; the chunk goes to the general kernel
      // the slices between my previous one and this one
      if (tid < 64u) {
        uint32_t cnt_i = 0u, sum_i = 0u;
        bool fail = false;
        if (tid < s - have) {
          const unsigned long long* r = rec + (have + tid) * 2u;
          // (relaxed accesses: the records carry their data, each word its own tag -- a release / acquire pair at agent
          // scope writes back and invalidates the L2, which multiplied the kernel's time by four)
          unsigned long long x = 0ull, y = 0ull;
          uint32_t spins = 0u;
          for (;; ++spins) {
            x = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            y = __hip_atomic_load(r + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((uint32_t)(x >> 32) == epoch && (uint32_t)(y >> 32) == epoch) || spins >= kScfSpinLimit) break;
            __builtin_amdgcn_s_sleep(2);
          }
          fail = (uint32_t)(x >> 32) != epoch || (uint32_t)(y >> 32) != epoch || (uint32_t)x == 0xffffffffu;
          cnt_i = (uint32_t)x;
          sum_i = (uint32_t)y;
        }
        const uint32_t cs = wave_sum(fail ? 0u : cnt_i), ss = wave_sum(sum_i);
        const bool any_fail = __ballot(fail) != 0ull;
        if (tid == 0u) {
          flags[1] = cs;
          flags[2] = ss;
          if (any_fail) flags[0] = 1u;
        }
      }
      __syncthreads();
      if (flags[0]) return;  // a slice in front is irregular (or never came: the general kernel decodes the chunk)
      pre_cnt += flags[1];
      pre_sum += flags[2];
      if (pre_cnt + n_tile > n) return;  // more tokens than points
      {
        uint32_t v = pre_sum + before;
        for (uint32_t k = tb; k < tb + my_cnt; ++k) {
          v += vals[k];
          vals[k] = v;
        }
      }
      __syncthreads();
      if (bpv == 2u) {
        uint16_t* o = reinterpret_cast<uint16_t*>(col) + pre_cnt;
        for (uint32_t i = tid; i < n_tile; i += kScfThreads) o[i] = (uint16_t)vals[i];
      } else {
        uint32_t* o = reinterpret_cast<uint32_t*>(col) + pre_cnt;
        for (uint32_t i = tid; i < n_tile; i += kScfThreads) o[i] = vals[i];
      }
      pre_cnt += n_tile;
      pre_sum += tile_sum;
      have = s + 1u;
      // the last slice closes the section: every point has its token and the last byte ends one
      const bool ok = s + 1u < n_slices || (pre_cnt == n && (src[src_size - 1u] & 0x80u) == 0u);
      if (ok && tid == 0u && (atomicAdd(slices_done + c, 1u) & 0xffffffu) + 1u == n_slices) sec_cols[c] = 1u;
      __syncthreads();  // (tile, vals and flags are written again)
    }
    return;
  }

  if (mode == 2u || mode == 3u) {
    // (round 6) every workgroup of the chunk parses the run table -- a section of at most 4 KiB -- and fills its share of the
    // values: one lidar cloud per call has 4 chunks on 256 CUs, and the fill of 32768 values by 256 threads was the longest
    // kernel of the call (26 us)
    if (src_size - off < 4u) return;
    const uint32_t runs = (uint32_t)src[off] | ((uint32_t)src[off + 1u] << 8) | ((uint32_t)src[off + 2u] << 16) | ((uint32_t)src[off + 3u] << 24);
    off += 4u;
    if (runs == 0u || runs > kScfMaxRuns || runs > n) return;
    // run table: start index, value before the run (DeltaRle) or the value (Rle), difference per element
    uint32_t* r_start = vals;                  // [runs + 1]
    uint32_t* r_base = vals + kScfMaxRuns + 8u;  // [runs]
    uint32_t* r_diff = r_base + kScfMaxRuns;     // [runs]
    // the records go to LDS first (one lane parsing them from global memory would wait for every byte); sections that do
    // not fit the tile buffer are left to the general kernel
    const uint32_t sec_bytes = src_size - off;
    if (sec_bytes > kScfTileBytes) return;
    for (uint32_t i = tid; i < (sec_bytes + 15u) / 16u; i += kScfThreads) {
      uint32_t w[4];
      fp_load16u(src, src_size, off + i * 16u, w);
      *reinterpret_cast<uint4*>(tile + i * 4u) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    if (mode == 3u) {
      // DeltaRle: every token is a varint -- (difference, run length) pairs. Token ends are numbered by a block scan,
      // a thread takes 16 consecutive tokens = 8 runs, and two more scans turn (length, difference * length) into the
      // run's first index and the value in front of it.
      uint32_t b[4];
      {
        const uint4 q = *reinterpret_cast<const uint4*>(tile + tid * 4u);
        b[0] = q.x; b[1] = q.y; b[2] = q.z; b[3] = q.w;
      }
      uint32_t ends = fp_ends16(b);
      {
        const uint32_t first = tid * 16u;  // bytes behind the section are no tokens
        ends = first >= sec_bytes ? 0u : (sec_bytes - first >= 16u ? ends : ends & ((1u << (sec_bytes - first)) - 1u));
      }
      uint32_t n_tok;
      const uint32_t tb = block_exclusive_scan<T>((uint32_t)__builtin_popcount(ends), scan, &n_tok);
      {
        uint32_t k = tb;
        for (uint32_t m = ends; m; m &= m - 1u) end_pos[k++] = (uint16_t)(tid * 16u + (uint32_t)__builtin_ctz(m));
      }
      __syncthreads();
      // (uniform) the tokens are exactly the pairs and the last one closes the section
      if (n_tok != runs * 2u || (uint32_t)end_pos[n_tok - 1u] + 1u != sec_bytes) return;
      uint32_t dif[8], len[8];
      uint32_t s_len = 0u, s_val = 0u;
      bool bad = false;
#pragma unroll
      for (uint32_t j = 0; j < 16u; ++j) {
        const uint32_t k = tid * 16u + j;
        uint64_t u = 0ull;
        if (k < n_tok) {
          const uint32_t start = k ? (uint32_t)end_pos[k - 1u] + 1u : 0u;
          const uint32_t tl = (uint32_t)end_pos[k] - start + 1u;
          const uint32_t di = start >> 2, sh = (start & 3u) * 8u;
          const uint32_t w0 = tile[di], w1 = tile[di + 1u], w2 = tile[di + 2u];
          const uint32_t lo = sh ? ((w0 >> sh) | (w1 << (32u - sh))) : w0;
          const uint32_t b4 = (sh ? ((w1 >> sh) | (w2 << (32u - sh))) : w1) & 0xffu;
          const uint32_t g = (lo & 0x7fu) | (((lo >> 8) & 0x7fu) << 7) | (((lo >> 16) & 0x7fu) << 14) | (((lo >> 24) & 0x7fu) << 21);
          const uint32_t keep = tl >= 4u ? 0x0fffffffu : ((1u << (7u * tl)) - 1u);
          u = (uint64_t)(g & keep) | (tl == 5u ? ((uint64_t)(b4 & 0x7fu) << 28) : 0ull);
          bad = bad || tl > 5u;
        }
        if ((j & 1u) == 0u) {  // the difference: zigzag + 1 (0 is the marker byte, no integer)
          bad = bad || (k < n_tok && u == 0ull);
          const uint64_t u1 = u - 1ull;
          dif[j >> 1] = k < n_tok ? (uint32_t)((u1 >> 1) ^ (0ull - (u1 & 1ull))) : 0u;
        } else {               // the run length: 1 .. n
          bad = bad || (k < n_tok && (u == 0ull || u > (uint64_t)n));
          len[j >> 1] = k < n_tok ? (uint32_t)u : 0u;
          s_len += len[j >> 1];
          s_val += dif[j >> 1] * len[j >> 1];
        }
      }
      if (bad) flags[0] = 1u;
      uint32_t tot_len, tot_val;
      uint32_t idx = block_exclusive_scan<T>(s_len, scan, &tot_len);
      uint32_t prev = block_exclusive_scan<T>(s_val, scan + 20, &tot_val);
      (void)tot_val;
      if (flags[0] || tot_len != n) return;  // uniform (the scans' barriers are behind every write of flags[0])
#pragma unroll
      for (uint32_t j = 0; j < 8u; ++j) {
        const uint32_t r = tid * 8u + j;
        if (r < runs) {
          r_start[r] = idx;
          r_base[r] = prev;
          r_diff[r] = dif[j];
        }
        idx += len[j];
        prev += dif[j] * len[j];
      }
      if (tid == 0) r_start[runs] = n;
    } else {
      // Rle: raw values between the run lengths, so that token ends cannot be told from value bytes -- a constant
      // field is one run; more than a few runs go to the general kernel
      if (runs > 32u) return;
      if (tid == 0) {
        const uint8_t* p = reinterpret_cast<const uint8_t*>(tile);
        const uint8_t* end = p + sec_bytes;
        uint32_t idx = 0u;
        bool bad = false;
        for (uint32_t r = 0; r < runs && !bad; ++r) {
          if ((uint32_t)(end - p) < bpv) { bad = true; break; }
          uint32_t base_v = 0u;
          for (uint32_t k = 0; k < bpv; ++k) base_v |= (uint32_t)p[k] << (8u * k);
          p += bpv;
          uint32_t len = 0u, sh = 0u;  // uvarint(run_len)
          for (;;) {
            if (p >= end || sh > 21u) { bad = true; break; }
            const uint8_t byte = *p++;
            len |= (uint32_t)(byte & 0x7fu) << sh;
            if (!(byte & 0x80u)) break;
            sh += 7u;
          }
          if (bad || len == 0u || len > n - idx) { bad = true; break; }
          r_start[r] = idx;
          r_base[r] = base_v;
          r_diff[r] = 0u;
          idx += len;
        }
        if (!bad && (idx != n || p != end)) bad = true;
        r_start[runs] = n;
        flags[0] = bad ? 1u : 0u;
      }
    }
    __syncthreads();
    if (flags[0]) return;
    // fill: a thread owns 8 consecutive values -- one search for the first, a walk along the table for the rest, and
    // one 16/32-byte store
    for (uint32_t i0 = (part * kScfThreads + tid) * 8u; i0 < n; i0 += parts * kScfThreads * 8u) {
      uint32_t lo = 0u, hi = runs;  // last r with r_start[r] <= i0
      while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (r_start[mid] <= i0) lo = mid;
        else hi = mid;
      }
      uint32_t nxt = r_start[lo + 1u], st = r_start[lo], bs = r_base[lo], df = r_diff[lo];
      uint32_t v[8];
#pragma unroll
      for (uint32_t j = 0; j < 8u; ++j) {
        const uint32_t i = i0 + j;
        if (i >= nxt && i < n) {  // (runs are never empty: one step)
          ++lo;
          st = nxt; nxt = r_start[lo + 1u]; bs = r_base[lo]; df = r_diff[lo];
        }
        v[j] = bs + df * (i - st + 1u);
      }
      if (i0 + 8u <= n) {
        if (bpv == 2u) {
          const uint4 q = make_uint4((v[0] & 0xffffu) | (v[1] << 16), (v[2] & 0xffffu) | (v[3] << 16),
                                     (v[4] & 0xffffu) | (v[5] << 16), (v[6] & 0xffffu) | (v[7] << 16));
          __builtin_memcpy(col + (size_t)i0 * 2u, &q, 16);
        } else {
          const uint4 q0 = make_uint4(v[0], v[1], v[2], v[3]), q1 = make_uint4(v[4], v[5], v[6], v[7]);
          __builtin_memcpy(col + (size_t)i0 * 4u, &q0, 16);
          __builtin_memcpy(col + (size_t)i0 * 4u + 16u, &q1, 16);
        }
      } else {
        for (uint32_t j = 0; j < 8u && i0 + j < n; ++j) {
          if (bpv == 2u) { const uint16_t h = (uint16_t)v[j]; __builtin_memcpy(col + (size_t)(i0 + j) * 2u, &h, 2); }
          else __builtin_memcpy(col + (size_t)(i0 + j) * 4u, &v[j], 4);
        }
      }
    }
    // the last of the chunk's workgroups to get here marks the column complete (the counter: the low bits of slices_done)
    __syncthreads();
    if (tid == 0 && (atomicAdd(slices_done + c, 1u) & 0xffffffu) + 1u == parts) sec_cols[c] = 1u;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// k_decode_sections_cols: in front of k_decode_points, for chunks whose sections it cannot fold from a palette table
// (DeltaVarint / Rle / DeltaRle sections, large palettes). The sections are decoded into DENSE COLUMNS (value i of
// the chunk at col[a] + (first_point + i) * bpv) with the parallel section decoder (decode_sections_core); the point
// kernel then reads a point's integer fields next to its floats and writes every point ONCE -- round 2 decoded these
// sections behind the point kernel straight into the AoS cloud, which dirtied every output line a second time
// (C3 / C4: 0.26 ms of a 0.57 / 0.74 ms decode).
// Where the sections begin comes from k_locate_sections (reg_end_pre). sec_cols[c] = 1: the columns hold this chunk's
// fields. Chunks k_locate_sections did not look at (a small Palette seen from the end) are left alone.
// grid = n_chunks, kDvThreads threads, DecSecLds::kTotal bytes of LDS.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kDvThreads) void k_decode_sections_cols(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                     const DecChunk* __restrict__ chunks, uint32_t n_ops,
                                                                     uint8_t* __restrict__ col0, uint8_t* __restrict__ col1,
                                                                     const uint32_t* __restrict__ reg_end_pre,
                                                                     uint8_t* __restrict__ sec_cols, uint32_t after_fast) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const DecChunk dc = chunks[c];
  if (after_fast) {
    if (sec_cols[c]) return;  // k_sections_cols_fast took this chunk (it also cleared the flag of all others)
  } else if (tid == 0) {
    sec_cols[c] = 0u;
  }
  if (!dc.valid || plan.n_adaptive == 0u || plan.n_adaptive > kFastPalFields) return;
  for (uint32_t a = 0; a < plan.n_adaptive; ++a)
    if (plan.adaptive[a].bpv > 4u) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  const uint32_t reg_size = reg_end_pre[c];  // k_locate_sections
  if (reg_size == 0xffffffffu || reg_size > src_size) return;  // not looked for / fewer tokens than the points need
  const bool ok = decode_sections_core(plan, src, src_size, reg_size, n,
                                       [&](uint32_t a) {
                                         SecFieldOut fo;
                                         const uint32_t bpv = plan.adaptive[a].bpv;
                                         fo.base = (a == 0u ? col0 : col1) + (size_t)dc.first_point * bpv;
                                         fo.step = bpv;
                                         fo.off = 0u;
                                         return fo;
                                       },
                                       smem);
  if (ok && tid == 0) sec_cols[c] = 1u;
}

// ---------------------------------------------------------------------------------------------------------------
// fp_setup: what both point kernels (k_decode_points, k_decode_points_w) do before their first byte of regular stream:
// find out where the chunk's sections begin if that is cheap (columns from k_decode_sections_cols, or a lone Palette seen
// from the end of the payload, or -- two fields, no columns -- a count over the payload), read the headers of sections
// that can be folded into the point pass and copy their palette tables to LDS. All threads of the workgroup call it
// (barriers inside). misc: [40] pre-pass result, [42..44) guess scratch, [44..60) wave counts, [64] folded sections,
// [68..) table offsets, [72..) FpSection records. Returns reg_size (0xffffffff = unknown), *from_cols_out.
// ---------------------------------------------------------------------------------------------------------------
template <int NOPS, int NF, int T>
__device__ __forceinline__ uint32_t fp_setup(const DevPlan& plan, const uint8_t* __restrict__ src, uint32_t src_size, uint32_t n,
                                             uint32_t uses_v5, uint32_t c, const uint32_t* __restrict__ reg_end_pre,
                                             const uint8_t* __restrict__ sec_cols, uint32_t* misc, uint32_t* pal,
                                             bool* from_cols_out) {
  constexpr uint32_t NW = (uint32_t)T / 64u;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t target = n * NOPS;
  // ---------------------------------------------------------------------------------------------------------
  // pre-pass: the payload offset behind token number `target`. Wave w counts the token ends of its 1/16 of the
  // payload; the wave that holds the last token walks its part again and finds the byte.
  // ---------------------------------------------------------------------------------------------------------
  uint32_t reg_size = 0xffffffffu;
  const bool v5_sections = uses_v5 && plan.n_adaptive != 0u;
  bool located = false;
  // k_decode_sections_cols has decoded this chunk's sections into columns: every point takes its integer fields from there
  const bool from_cols = NF != 0 && v5_sections && sec_cols != nullptr && sec_cols[c] != 0u && plan.n_adaptive <= (uint32_t)NF;
  if (NF > 2 && !from_cols) {  // (uniform) instantiations for many fields only merge columns: nothing is located or folded here
    *from_cols_out = false;
    return 0xffffffffu;
  }
  if (from_cols) {
    reg_size = reg_end_pre[c];
    located = true;
    if (tid == 0) {
      FpSection* sec = reinterpret_cast<FpSection*>(misc + 72);
      for (uint32_t a = 0; a < plan.n_adaptive; ++a) {
        sec[a].field_off = plan.adaptive[a].offset;
        sec[a].bpv = plan.adaptive[a].bpv;
        sec[a].count = 0u;
        sec[a].bits = 0u;
        sec[a].index_off = 0u;
      }
      misc[64] = plan.n_adaptive;
    }
    __syncthreads();
  }
  if (!from_cols && v5_sections && plan.n_adaptive == 1u && plan.adaptive[0].bpv <= 4u) {
    // One section behind the regular stream. If it is a Palette of U entries its size follows from U alone
    // (3 + U * bpv + ceil(bits(U) * n / 8)), so every U has one place where its header would have to be: the threads
    // try them all. A hit that is no header (3 bytes of a token stream that look like one) is found out when the
    // tiles end somewhere else: the chunk's sections are then left to the section kernels, like any chunk without a
    // hit. This spares the counting pass over the whole payload.
    const uint32_t bpv = plan.adaptive[0].bpv;
    const uint32_t U = pal_guess_from_end<T>(src, src_size, n, bpv, misc + 42);
    if (U != 0xffffffffu) reg_size = src_size - (uint32_t)(3ull + (uint64_t)U * bpv + ((uint64_t)palette_bits(U) * n + 7u) / 8u);
    located = true;
  }
  if (v5_sections && !located) {
    const uint32_t part = (((src_size + 15u) / 16u + NW - 1u) / NW) * 16u;  // bytes per wave, multiple of 16
    const uint32_t w0 = wave * part, w1 = min(src_size, w0 + part);
    uint32_t cnt = 0u;
    for (uint32_t o = w0 + lane * 16u; o < w1; o += 1024u) {
      uint32_t b[4];
      fp_load16(src, src_size, o, b);
#pragma unroll
      for (int k = 0; k < 4; ++k) cnt += (uint32_t)__builtin_popcount(~b[k] & 0x80808080u);  // token ends = bytes with a clear MSB
    }
    const uint32_t wsum = wave_sum(cnt);
    if (lane == 0u) misc[44u + wave] = wsum;
    __syncthreads();
    uint32_t before = 0u;
    for (uint32_t w = 0; w < wave; ++w) before += misc[44u + w];
    if (target != 0u && before < target && target <= before + wsum) {  // the last token ends in my part (one wave)
      uint32_t seen = before;
      for (uint32_t o0 = w0; o0 < w1; o0 += 1024u) {
        const uint32_t o = o0 + lane * 16u;
        uint32_t b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (o < w1) fp_load16(src, src_size, o, b);
        const uint32_t ends = fp_ends16(b);
        const uint32_t cl = (uint32_t)__builtin_popcount(ends);
        const uint32_t incl = wave_inclusive_scan(cl);
        const uint32_t row = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (seen + row >= target) {
          const uint32_t my_first = seen + incl - cl;  // tokens before mine
          if (my_first < target && target <= my_first + cl) {
            uint32_t m = ends;
            for (uint32_t k = my_first + 1u; k < target; ++k) m &= m - 1u;  // drop the ends before the wanted one
            misc[40] = o + (uint32_t)__builtin_ctz(m) + 1u;
          }
          break;
        }
        seen += row;
      }
    }
    if (target == 0u && tid == 0) misc[40] = 0u;
    __syncthreads();
    reg_size = misc[40];
  }
  if (v5_sections && !from_cols) {

    // section headers (one thread): every section a small Palette of a 2- or 4-byte field -> fold them in
    if (tid == 0 && reg_size != 0xffffffffu && plan.n_adaptive <= (uint32_t)NF) {
      uint32_t off = reg_size;
      bool ok = true;
      FpSection* sec = reinterpret_cast<FpSection*>(misc + 72);
      for (uint32_t a = 0; a < plan.n_adaptive && ok; ++a) {
        const uint32_t bpv = plan.adaptive[a].bpv;
        if (bpv > 4u || src_size - min(src_size, off) < 3u || src[off] != 1u) { ok = false; break; }
        const uint32_t count = (uint32_t)src[off + 1u] | ((uint32_t)src[off + 2u] << 8);
        off += 3u;
        if (count == 0u || count > kFastPalEntries || (uint64_t)(src_size - off) < (uint64_t)count * bpv) { ok = false; break; }
        const uint32_t bits = palette_bits(count);
        const uint32_t index_bytes = (uint32_t)(((uint64_t)bits * n + 7u) / 8u);
        const uint32_t table_off = off;
        off += count * bpv;
        if (src_size - off < index_bytes) { ok = false; break; }
        sec[a].field_off = plan.adaptive[a].offset;
        sec[a].bpv = bpv;
        sec[a].count = count;
        sec[a].bits = bits;
        sec[a].index_off = off;
        misc[68u + a] = table_off;
        off += index_bytes;
      }
      if (ok && off == src_size) misc[64] = plan.n_adaptive;  // (trailing bytes: the serial decoder raises the error)
    }
    __syncthreads();
    const uint32_t n_fold = misc[64];
    for (uint32_t a = 0; a < n_fold; ++a) {  // palette tables -> LDS
      const FpSection s = reinterpret_cast<const FpSection*>(misc + 72)[a];
      const uint8_t* tp = src + misc[68u + a];
      for (uint32_t k = tid; k < s.count; k += T) {
        uint32_t v = 0u;
        for (uint32_t bb = 0; bb < s.bpv; ++bb) v |= ((uint32_t)tp[(size_t)k * s.bpv + bb]) << (8u * bb);
        pal[a * kFastPalEntries + k] = v;
      }
    }
    __syncthreads();
  }
  *from_cols_out = from_cols;
  return reg_size;
}

template <int NOPS, int NF>
__global__ __launch_bounds__(kFpThreads) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_decode_points(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                              const DecChunk* __restrict__ chunks, uint8_t* __restrict__ out,
                                                              uint32_t* __restrict__ reg_end, uint8_t* __restrict__ sec_done,
                                                              uint32_t uses_v5, uint32_t* __restrict__ status,
                                                              const uint8_t* __restrict__ col0, const uint8_t* __restrict__ col1,
                                                              const uint32_t* __restrict__ reg_end_pre,
                                                              const uint8_t* __restrict__ sec_cols, uint32_t fill_zero) {
  using L = FpLds<NOPS, NF>;
  constexpr uint32_t NFA = NF ? NF : 1;  // array extents (NF == 0: nothing is ever folded)
  constexpr int T = kFpThreads;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* tile = reinterpret_cast<uint32_t*>(smem + L::kTileOff);             // dwords; byte 16 = first tile byte
  uint16_t* pos_list = reinterpret_cast<uint16_t*>(smem + L::kPosOff);          // [0] = end of the token before the tile
  float* stage = reinterpret_cast<float*>(smem);                                // overlays tile and list in phase B
  uint8_t* scanrec = smem + L::kScanOff;
  uint32_t* pal = reinterpret_cast<uint32_t*>(smem + L::kPalOff);
  uint32_t* misc = reinterpret_cast<uint32_t*>(smem + L::kMiscOff);             // [0] irregular, [2..34) scan scratch,
                                                                                // [40..) pre-pass, [64..) sections
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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

  if (tid < 4u) tile[tid] = 0u;  // the 16 bytes in front of a tile: token ends
  if (tid == 0) {
    misc[0] = 0u;
    misc[1] = 0u;            // a folded Palette index was out of range
    misc[40] = 0xffffffffu;  // pre-pass: payload offset behind the regular stream
    misc[42] = 0xffffffffu;  // entries of the Palette section found by its size (smallest hit)
    misc[64] = 0u;           // sections folded in
  }
  if (tid < (uint32_t)(NOPS + 2)) reinterpret_cast<uint32_t*>(scanrec + 16u * L::kWaveRec)[tid] = 0u;  // carry of tile 0: values 0
  __syncthreads();

  bool from_cols = false;
  const uint32_t reg_size = fp_setup<NOPS, NF, T>(plan, src, src_size, n, uses_v5, c, reg_end_pre, sec_cols, misc, pal, &from_cols);
  const uint32_t n_fold = misc[64];

  // folded Palette sections: parameters in (uniform) registers for the read-out
  uint32_t fs_off[NFA], fs_bpv[NFA], fs_count[NFA], fs_bits[NFA];
  const uint8_t* fs_idx[NFA];
#pragma unroll
  for (uint32_t a = 0; a < NFA; ++a) {
    const FpSection sct = reinterpret_cast<const FpSection*>(misc + 72)[a < n_fold ? a : 0u];
    fs_off[a] = sct.field_off;
    fs_bpv[a] = sct.bpv;
    fs_count[a] = sct.count;
    fs_bits[a] = a < n_fold ? sct.bits : 0u;
    fs_idx[a] = src + (a < n_fold ? sct.index_off : 0u);
  }
  // which store forms the layout allows (uniform)
  bool contig = ((step | plan.ops[0].offset) & 3u) == 0u;
#pragma unroll
  for (int o = 1; o < NOPS; ++o) contig = contig && plan.ops[o].offset == plan.ops[0].offset + 4u * (uint32_t)o;
  bool packed = true;  // the floats back to back at any alignment, all of them stored
#pragma unroll
  for (int o = 0; o < NOPS; ++o) packed = packed && plan.ops[o].offset != 0xffffffffu && plan.ops[o].offset == plan.ops[0].offset + 4u * (uint32_t)o;
  float res[NOPS];
  uint32_t foff[NOPS];
#pragma unroll
  for (int o = 0; o < NOPS; ++o) {
    res[o] = plan.ops[o].res_f;
    foff[o] = plan.ops[o].offset;
  }

  // ---------------------------------------------------------------------------------------------------------
  // tiles
  // ---------------------------------------------------------------------------------------------------------
  uint32_t pos = 0u;       // payload offset of the tile (a point boundary)
  uint32_t pts_done = 0u;
  uint32_t par = 0u;       // which carry record this tile reads (it writes the other one)
  bool bad = false;
  uint32_t b[4];           // my 16 bytes of the tile (those of the next tile are fetched while this one is decoded)
  fp_load16(src, src_size, tid * 16u, b);
  while (pts_done < n) {
    if (pos >= src_size) { bad = true; break; }
    // ---- phase A: bytes -> LDS; the byte where every point starts -> list
    *reinterpret_cast<uint4*>(tile + 4u + tid * 4u) = make_uint4(b[0], b[1], b[2], b[3]);
    const uint32_t ends = fp_ends16(b);
    uint32_t n_tile;
    const uint32_t tb = block_exclusive_scan<T>((uint32_t)__builtin_popcount(ends), misc + 2, &n_tile);  // barrier inside
    const uint32_t want_pts = min(kFpTilePoints, n - pts_done);
    {
      // my token ends are tokens tb, tb + 1, ... of the tile; point q starts behind the end of token q * NOPS - 1.
      // Only those ends are listed: the tokens inside a point are found by walking its bytes in phase B.
      const uint32_t tq = tb / (uint32_t)NOPS;
      const uint32_t rank0 = (uint32_t)(NOPS - 1) - (tb - tq * (uint32_t)NOPS);  // ends of mine to skip first
      uint32_t m = ends;
#pragma unroll
      for (uint32_t i = 0; i + 1u < (uint32_t)NOPS; ++i) {
        const uint32_t mm = m & (m - 1u);
        m = (i < rank0) ? mm : m;
      }
      uint32_t q = tq + 1u;
      while (m != 0u && q <= want_pts) {
        pos_list[q] = (uint16_t)(tid * 16u + (uint32_t)__builtin_ctz(m) + 1u);
        ++q;
#pragma unroll
        for (int i = 0; i < NOPS; ++i) m &= m - 1u;
      }
      if (tid == 0) pos_list[0] = (uint16_t)0u;
    }
    __syncthreads();
    const uint32_t npts = min(n_tile / (uint32_t)NOPS, want_pts);  // whole points of this tile
    if (npts == 0u) { bad = true; break; }           // 16 KiB without NOPS token ends: not a FloatN stream
    const uint32_t next_pos = pos + (uint32_t)pos_list[npts];  // behind the last token this tile consumes
    if (pts_done + npts < n) fp_load16(src, src_size, next_pos + tid * 16u, b);  // (`ends` is all this tile still needs of b)

    // ---- phase B: my points [q0, q0 + kFpPPT)
    const uint32_t q0 = tid * kFpPPT;
    int32_t dlt[kFpPPT][NOPS];   // 0x80000000 = the NaN marker (no token of at most 4 bytes decodes to it)
    bool long_tok = false;
    bool any_marker = false;
#pragma unroll
    for (uint32_t i = 0; i < kFpPPT; ++i) {
      // points behind the tile's last one run the same straight-line code on whatever the list holds (masked into the
      // tile); nothing of theirs is used: they come after every real point in scan order
      const bool have = (q0 + i) < npts;
      const uint32_t byte0 = 16u + ((uint32_t)pos_list[q0 + i] & (kFpTileBytes - 1u));  // byte index in the LDS copy
      const uint32_t di = byte0 >> 2, sh = (byte0 & 3u) * 8u;
      uint32_t d[NOPS + 1];
#pragma unroll
      for (int k = 0; k <= NOPS; ++k) d[k] = tile[di + (uint32_t)k];
      uint32_t W[NOPS];  // W[k] = bytes [4k, 4k + 4) behind the current token's start
#pragma unroll
      for (int k = 0; k < NOPS; ++k) W[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        // Tokens of up to 4 bytes (|delta| < 2^27 ticks) are decoded here with 32-bit arithmetic; a longer one
        // (special values, damaged streams) sends the chunk to k_decode_varint.
        const uint32_t w = W[0];
        const uint32_t t = ~w & 0x80808080u;     // the bytes that can end the token
        const uint32_t keep = t ^ (t - 1u);      // everything up to and including the first of them
        const uint32_t wk = w & keep;            // == 0: the marker byte 0x00
        const uint32_t lo = wk & 0x7f7f7f7fu;
        const uint32_t u = (lo & 0x7fu) | (((lo >> 8) & 0x7fu) << 7) | (((lo >> 16) & 0x7fu) << 14) | ((lo >> 24) << 21);
        // no end in 4 bytes; or an overlong zero, which decodeVarint rejects
        long_tok = long_tok || (have && (t == 0u || (lo == 0u && wk != 0u)));
        any_marker = any_marker || wk == 0u;
        const uint32_t u1 = u - 1u;
        dlt[i][o] = (int32_t)((u1 >> 1) ^ (0u - (u1 & 1u)));   // u == 0 -> 0x80000000
        if (o + 1 < NOPS) {
          const uint32_t adv = (uint32_t)__ffs((int)t);  // bits of this token: 8, 16, 24, 32
#pragma unroll
          for (int k = 0; k + 1 < NOPS - o; ++k) W[k] = (uint32_t)(((((uint64_t)W[k + 1]) << 32) | W[k]) >> adv);
        }
      }
    }
    // folded Palette fields of my points: their indexes are consecutive bits, one 8-byte window holds all kFpPPT
    uint32_t pv[NFA][kFpPPT];
    const uint32_t SP = (uint32_t)NOPS + n_fold;  // dwords per staged point
    if (from_cols) {
#pragma unroll
      for (uint32_t a = 0; a < NFA; ++a) {
        if (a >= n_fold) break;  // uniform
        const uint8_t* colp = (a == 0u ? col0 : col1) + ((size_t)dc.first_point + pts_done) * fs_bpv[a];
#pragma unroll
        for (uint32_t i = 0; i < kFpPPT; ++i) {
          const uint32_t q = min(q0 + i, npts - 1u);  // (points behind the tile's last one read a valid slot; unused)
          pv[a][i] = fs_bpv[a] == 2u ? (uint32_t)reinterpret_cast<const uint16_t*>(colp)[q] : reinterpret_cast<const uint32_t*>(colp)[q];
        }
      }
    } else if (n_fold != 0u) {
#pragma unroll
      for (uint32_t a = 0; a < NFA; ++a) {
        if (a >= n_fold) break;  // uniform
        const uint32_t bits = fs_bits[a];
        uint64_t w64 = 0u;
        if (bits != 0u && q0 < npts) {
          const uint32_t bit0 = (pts_done + q0) * bits;                 // < 32768 * 10
          const uint8_t* ib = fs_idx[a] + (bit0 >> 3);
          const uint32_t mis = (uint32_t)((uintptr_t)ib & 3u);
          const uint8_t* al = ib - mis;                                 // index_off >= 3: still inside the payload
          if (al + 8u <= src + src_size) {  // kFpPPT * 10 + 7 bits <= 5 bytes: two aligned dwords cover them
            const uint32_t* iq = reinterpret_cast<const uint32_t*>(al);
            w64 = ((((uint64_t)iq[1]) << 32) | iq[0]) >> (mis * 8u + (bit0 & 7u));
          } else {                          // the section's last bytes
            const uint32_t avail = (uint32_t)(src + src_size - ib);
            for (uint32_t k = 0; k < 5u && k < avail; ++k) w64 |= ((uint64_t)ib[k]) << (8u * k);
            w64 >>= (bit0 & 7u);
          }
        }
        bool beyond = false;
#pragma unroll
        for (uint32_t i = 0; i < kFpPPT; ++i) {
          const uint32_t idx = (uint32_t)(w64 >> (i * bits)) & ((1u << bits) - 1u);  // < kFastPalEntries
          beyond = beyond || ((q0 + i) < npts && idx >= fs_count[a]);
          pv[a][i] = pal[a * kFastPalEntries + idx];
        }
        if (beyond) misc[1] = 1u;  // index beyond the palette: the serial decoder redoes the sections and raises the error
      }
    }
    // local sums per lane with NaN resets, then the segmented scan over the threads
    int32_t acc[NOPS];
    uint32_t fl = 0u;
    const bool wave_marker = __ballot(any_marker) != 0ull;  // (markers of points that do not exist only cost time)
#pragma unroll
    for (int o = 0; o < NOPS; ++o) acc[o] = 0;
    int32_t inc[NOPS];
    uint32_t fin = 0u;
    if (!wave_marker) {  // no marker in this wave (the rule for lidar data): plain sums, DPP prefix sums
#pragma unroll
      for (uint32_t i = 0; i < kFpPPT; ++i) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o) acc[o] = (int32_t)((uint32_t)acc[o] + (uint32_t)dlt[i][o]);
      }
#pragma unroll
      for (int o = 0; o < NOPS; ++o) inc[o] = (int32_t)wave_inclusive_scan((uint32_t)acc[o]);
    } else {
#pragma unroll
      for (uint32_t i = 0; i < kFpPPT; ++i) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o) {
          const bool m = dlt[i][o] == (int32_t)0x80000000;
          acc[o] = m ? 0 : (int32_t)((uint32_t)acc[o] + (uint32_t)dlt[i][o]);
          if (m) fl |= 1u << o;
        }
      }
      fin = fl;
#pragma unroll
      for (int o = 0; o < NOPS; ++o) inc[o] = acc[o];
#pragma unroll
      for (int dl = 1; dl < 64; dl <<= 1) {
        const uint32_t of = (uint32_t)__shfl_up((int)fin, dl);
        int32_t ov[NOPS];
#pragma unroll
        for (int o = 0; o < NOPS; ++o) ov[o] = __shfl_up(inc[o], dl);
        if (lane >= (uint32_t)dl) {
#pragma unroll
          for (int o = 0; o < NOPS; ++o)
            if (!(fin & (1u << o))) inc[o] = (int32_t)((uint32_t)inc[o] + (uint32_t)ov[o]);
          fin |= of;
        }
      }
    }
    if (long_tok) misc[0] = 1u;
    if (lane == 63u) {
      int32_t* rec = reinterpret_cast<int32_t*>(scanrec + wave * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) rec[o] = inc[o];
      *reinterpret_cast<uint32_t*>(scanrec + wave * L::kWaveRec + NOPS * 4u) = fin;
    }
    __syncthreads();  // also: every thread is done with the tile bytes and the list -> the staging area may overlay them
    int32_t in[NOPS];
    {
      const int32_t* crec = reinterpret_cast<const int32_t*>(scanrec + (16u + par) * L::kWaveRec);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) in[o] = crec[o];
      {
        // state behind waves 0..wave-1: lane l < 16 holds wave l's record, a 4-step segmented scan over those 16 lanes
        // (row-local DPP shifts), lane wave-1 then has the combination of all waves before mine
        int32_t rv[NOPS];
        uint32_t rfl = 0u;
        const uint32_t wl = lane & 15u;
        {
          const int32_t* rec = reinterpret_cast<const int32_t*>(scanrec + wl * L::kWaveRec);
#pragma unroll
          for (int o = 0; o < NOPS; ++o) rv[o] = rec[o];
          rfl = *reinterpret_cast<const uint32_t*>(scanrec + wl * L::kWaveRec + NOPS * 4u);
        }
#define FP_SEG_STEP(DL)                                                                                         \
  {                                                                                                            \
    const uint32_t of = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)rfl, 0x110 + DL, 0xf, 0xf, true); /* row_shr */ \
    int32_t ov[NOPS];                                                                                          \
    _Pragma("unroll") for (int o = 0; o < NOPS; ++o) ov[o] = __builtin_amdgcn_update_dpp(0, rv[o], 0x110 + DL, 0xf, 0xf, true); \
    if (wl >= (uint32_t)DL) {                                                                                  \
      _Pragma("unroll") for (int o = 0; o < NOPS; ++o)                                                         \
        if (!(rfl & (1u << o))) rv[o] = (int32_t)((uint32_t)rv[o] + (uint32_t)ov[o]);                          \
      rfl |= of;                                                                                               \
    }                                                                                                          \
  }
        FP_SEG_STEP(1)
        FP_SEG_STEP(2)
        FP_SEG_STEP(4)
        FP_SEG_STEP(8)
#undef FP_SEG_STEP
        if (wave > 0u) {
          const uint32_t pfw = (uint32_t)__builtin_amdgcn_readlane((int)rfl, (int)wave - 1);
#pragma unroll
          for (int o = 0; o < NOPS; ++o) {
            const int32_t pvw = __builtin_amdgcn_readlane(rv[o], (int)wave - 1);
            in[o] = (pfw & (1u << o)) ? pvw : (int32_t)((uint32_t)in[o] + (uint32_t)pvw);
          }
        }
      }
      const uint32_t pf = (uint32_t)__shfl_up((int)fin, 1);
#pragma unroll
      for (int o = 0; o < NOPS; ++o) {
        const int32_t pv = __shfl_up(inc[o], 1);
        if (lane > 0u) in[o] = (pf & (1u << o)) ? pv : (int32_t)((uint32_t)in[o] + (uint32_t)pv);
      }
    }
    if (misc[0]) { bad = true; break; }  // uniform (read behind the barrier)
    // final values -> staging (point-major, NOPS floats per point); the values behind the tile's last point are the
    // next tile's carry (the record this tile did not read)
    int32_t* crec_next = reinterpret_cast<int32_t*>(scanrec + (17u - par) * L::kWaveRec);
    if (!wave_marker) {
#pragma unroll
      for (uint32_t i = 0; i < kFpPPT; ++i) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o) {
          in[o] = (int32_t)((uint32_t)in[o] + (uint32_t)dlt[i][o]);
          stage[(q0 + i) * SP + (uint32_t)o] = __fmul_rn((float)in[o], res[o]);  // points >= npts: harmless slots, never read
        }
#pragma unroll
        for (uint32_t a = 0; a < NFA; ++a)
          if (a < n_fold) stage[(q0 + i) * SP + (uint32_t)NOPS + a] = __uint_as_float(pv[a][i]);
        if (q0 + i + 1u == npts) {
#pragma unroll
          for (int o = 0; o < NOPS; ++o) crec_next[o] = in[o];
        }
      }
    } else {
#pragma unroll
      for (uint32_t i = 0; i < kFpPPT; ++i) {
#pragma unroll
        for (int o = 0; o < NOPS; ++o) {
          const bool m = dlt[i][o] == (int32_t)0x80000000;
          in[o] = m ? 0 : (int32_t)((uint32_t)in[o] + (uint32_t)dlt[i][o]);
          const float f = m ? __uint_as_float(0x7fc00000u) : __fmul_rn((float)in[o], res[o]);
          stage[(q0 + i) * SP + (uint32_t)o] = f;
        }
#pragma unroll
        for (uint32_t a = 0; a < NFA; ++a)
          if (a < n_fold) stage[(q0 + i) * SP + (uint32_t)NOPS + a] = __uint_as_float(pv[a][i]);
        if (q0 + i + 1u == npts) {
#pragma unroll
          for (int o = 0; o < NOPS; ++o) crec_next[o] = in[o];
        }
      }
    }
    par ^= 1u;
    __syncthreads();
    // ---- read-out: consecutive lanes, consecutive points; folded Palette fields complete the point. The layout
    // decisions are uniform: one switch per tile picks a straight-line variant (the common lidar layouts get code
    // without a branch per point).
    auto store_floats_contig = [&](uint8_t* pt, uint32_t q) {
      if (NOPS == 3) {
        FloatVec<3> v;
        v.v[0] = stage[q * SP]; v.v[1] = stage[q * SP + 1u]; v.v[2] = stage[q * SP + 2u];
        *reinterpret_cast<FloatVec<3>*>(pt + foff[0]) = v;
      } else {
        FloatVec<4> v;
#pragma unroll
        for (int o = 0; o < 4; ++o) v.v[o] = stage[q * SP + (uint32_t)o];
        *reinterpret_cast<FloatVec<4>*>(pt + foff[0]) = v;
      }
    };
    const bool one_u16 = NOPS == 3 && contig && n_fold == 1u && fs_bpv[0] == 2u && ((fs_off[0] | step) & 1u) == 0u;  // XYZ + one 16-bit field
    // fill_zero (CLDN_HIP_FILL_ZERO: the bytes no field covers may be written as 0): the two common padded layouts leave
    // as whole 16-byte stores -- XYZ f32 + a 16-bit field in a 16-byte point, XYZ f32 + a 32-bit field at 16 in a
    // 32-byte point
    const bool full16 = fill_zero != 0u && one_u16 && step == 16u && foff[0] == 0u && fs_off[0] == 12u && ((uintptr_t)base & 15u) == 0u;
    const bool full32 = fill_zero != 0u && NOPS == 3 && contig && n_fold == 1u && fs_bpv[0] == 4u && step == 32u && foff[0] == 0u &&
                        fs_off[0] == 16u && ((uintptr_t)base & 15u) == 0u;
    if (full16) {
#pragma unroll
      for (uint32_t r = 0; r < kFpPPT; ++r) {
        const uint32_t q = r * (uint32_t)T + tid;
        if (q < npts) {
          float4 sv = *reinterpret_cast<const float4*>(stage + q * 4u);  // x, y, z, field
          sv.w = __uint_as_float(__float_as_uint(sv.w) & 0xffffu);
          *reinterpret_cast<float4*>(base + (size_t)(pts_done + q) * 16u) = sv;
        }
      }
    } else if (full32) {
#pragma unroll
      for (uint32_t r = 0; r < kFpPPT; ++r) {
        const uint32_t q = r * (uint32_t)T + tid;
        if (q < npts) {
          const float4 sv = *reinterpret_cast<const float4*>(stage + q * 4u);  // x, y, z, field
          float4* pt = reinterpret_cast<float4*>(base + (size_t)(pts_done + q) * 32u);
          pt[0] = make_float4(sv.x, sv.y, sv.z, 0.0f);
          pt[1] = make_float4(sv.w, 0.0f, 0.0f, 0.0f);
        }
      }
    } else if (contig && n_fold == 0u) {
#pragma unroll
      for (uint32_t r = 0; r < kFpPPT; ++r) {
        const uint32_t q = r * (uint32_t)T + tid;
        if (q < npts) store_floats_contig(base + (size_t)(pts_done + q) * step, q);
      }
    } else if (one_u16) {
#pragma unroll
      for (uint32_t r = 0; r < kFpPPT; ++r) {
        const uint32_t q = r * (uint32_t)T + tid;
        if (q < npts) {
          uint8_t* pt = base + (size_t)(pts_done + q) * step;
          const float4 sv = *reinterpret_cast<const float4*>(stage + q * 4u);  // x, y, z, field
          FloatVec<3> v;
          v.v[0] = sv.x; v.v[1] = sv.y; v.v[2] = sv.z;
          *reinterpret_cast<FloatVec<3>*>(pt + foff[0]) = v;
          *reinterpret_cast<uint16_t*>(pt + fs_off[0]) = (uint16_t)__float_as_uint(sv.w);
        }
      }
    } else {
#pragma unroll
      for (uint32_t r = 0; r < kFpPPT; ++r) {
        const uint32_t q = r * (uint32_t)T + tid;
        if (q < npts) {
          uint8_t* pt = base + (size_t)(pts_done + q) * step;
          if (contig) {
            store_floats_contig(pt, q);
          } else if (packed) {  // the floats lie back to back at an odd address (18-byte points): ONE unaligned store
            if (NOPS == 3) {    // (gfx950 takes the misalignment: a memcpy of 12 / 16 bytes is one global_store_dwordx3 / x4)
              FloatVec<3> v;
              v.v[0] = stage[q * SP]; v.v[1] = stage[q * SP + 1u]; v.v[2] = stage[q * SP + 2u];
              __builtin_memcpy(pt + foff[0], &v, 12);
            } else {
              FloatVec<4> v;
#pragma unroll
              for (int o = 0; o < 4; ++o) v.v[o] = stage[q * SP + (uint32_t)o];
              __builtin_memcpy(pt + foff[0], &v, 16);
            }
          } else {
#pragma unroll
            for (int o = 0; o < NOPS; ++o)
              if (foff[o] != 0xffffffffu) {
                const float f = stage[q * SP + (uint32_t)o];
                __builtin_memcpy(pt + foff[o], &f, 4);
              }
          }
#pragma unroll
          for (uint32_t a = 0; a < NFA; ++a) {
            if (a >= n_fold) break;  // uniform
            const uint32_t v = __float_as_uint(stage[q * SP + (uint32_t)NOPS + a]);
            if (fs_bpv[a] == 2u) {
              const uint16_t h = (uint16_t)v;
              __builtin_memcpy(pt + fs_off[a], &h, 2);
            } else if (fs_bpv[a] == 4u) {
              __builtin_memcpy(pt + fs_off[a], &v, 4);
            } else {
              st_raw(pt + fs_off[a], v, fs_bpv[a]);
            }
          }
        }
      }
    }
    pos = next_pos;
    pts_done += npts;
    __syncthreads();  // the staging area is free again
  }
  __syncthreads();
  if (tid == 0) {
    const bool redo = bad || misc[0] != 0u;
    reg_end[c] = redo ? kDecRedo : pos;
    const bool folded = !redo && n_fold != 0u && misc[1] == 0u && pos == reg_size;
    sec_done[c] = folded ? 2u : 0u;
    if (!redo) atomicAdd(&status[kStatFastRegular], 1u);
    if (folded) atomicAdd(&status[kStatFastSections], 1u);
  }
}

}  // namespace cldn
