// stage1_decode_sections_w.h -- the V5 sections of a chunk decoded SIDE BY SIDE (round 4): layouts with several integer
// channels (an Ouster-style point has five) went through k_decode_sections, which decodes a chunk's sections one after the
// other in a workgroup that holds 99 KB of LDS (1.6 of the 2.0 ms of a 16 M-point decode). Here
//   k_section_offsets  one light workgroup per chunk walks the sections WITHOUT decoding them and leaves, per field, a
//                      DecChunk-shaped record {where the section's body begins, its bytes, mode}: a Palette's size follows
//                      from its entry count (src/v5_codec.cpp:298-306), a DeltaVarint section ends behind its n-th token, a
//                      DeltaRle section behind its 2 * runs-th (a parallel count of the bytes with a clear MSB), an Rle
//                      section is parsed by one lane from an LDS copy (few runs, or the chunk stays with the old kernel);
//   k_sections_w       grid (chunks, fields): Palette, DeltaRle and Rle sections straight into the points (AoS);
//   k_decode_stream_w  (stage1_decode_stream.h, section mode, the same grid): a DeltaVarint section is a stream of n
//                      varint tokens of ONE integer op -- exactly what that kernel decodes;
//   k_sections_done    a chunk whose fields all arrived is marked (sec_done), the others are left to k_decode_sections and,
//                      in the end, the serial decoder, which raise the errors.
// (decodeV5AdaptiveIntSection, src/v5_codec.cpp:764-879.)
#pragma once

namespace cldn {

constexpr uint32_t kSoThreads = 256;
constexpr uint32_t kSoMaxFields = 8;
constexpr uint32_t kSoRleStage = 4096;   // bytes of an Rle section one lane parses from LDS
constexpr uint32_t kSwsThreads = 256;
constexpr uint32_t kSwsMaxRuns = 1024;
constexpr uint32_t kSwsMaxPal = 4096;    // palette entries of a section k_sections_w takes (16 KB of LDS)

// payload offset behind the `want`-th byte with a clear MSB at or behind `from` (want >= 1), 0xffffffff if there are fewer.
// All kSoThreads threads call it; sh = 40 LDS words.
__device__ __forceinline__ uint32_t so_nth_end(const uint8_t* __restrict__ src, uint32_t src_size, uint32_t from, uint32_t want, uint32_t* sh) {
  const uint32_t tid = threadIdx.x;
  uint32_t seen = 0u;
  for (uint32_t base = from; base < src_size; base += kSoThreads * 16u) {
    const uint32_t o = base + tid * 16u;
    uint32_t b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    if (o < src_size) fp_load16u(src, src_size, o, b);
    const uint32_t ends = wp_ends16(b);
    const uint32_t cl = (uint32_t)__builtin_popcount(ends);
    uint32_t total;
    const uint32_t before = block_exclusive_scan<(int)kSoThreads>(cl, sh + 2, &total);  // barrier inside
    if (seen + total >= want) {  // uniform
      if (tid == 0) sh[0] = 0xffffffffu;
      __syncthreads();
      const uint32_t my_first = seen + before;  // ends in front of mine
      if (my_first < want && want <= my_first + cl) {
        uint32_t m = ends;
        for (uint32_t k = my_first + 1u; k < want; ++k) m &= m - 1u;
        sh[0] = o + (uint32_t)__builtin_ctz(m) + 1u;
      }
      __syncthreads();
      const uint32_t r = sh[0];
      __syncthreads();
      return r;
    }
    seen += total;
    __syncthreads();  // (the scan's scratch is written again)
  }
  return 0xffffffffu;
}

// grid = n_chunks, kSoThreads threads. reg_end[c] = where the chunk's sections begin (written by the kernel that decoded the
// regular stream). dsec[a * n_chunks + c] = the section of field a: src_off / src_size = its body behind the mode byte,
// valid = mode + 1 (0: not sized). secs_ok[c] = 1 when all sections were sized and end with the payload; done_cnt[c] = 0.
__global__ __launch_bounds__(kSoThreads) void k_section_offsets(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                const DecChunk* __restrict__ chunks, uint32_t n_chunks,
                                                                const uint32_t* __restrict__ reg_end, DecChunk* __restrict__ dsec,
                                                                uint8_t* __restrict__ secs_ok, uint32_t* __restrict__ done_cnt,
                                                                const uint8_t* __restrict__ merged) {
  __shared__ uint32_t sh[48];
  __shared__ __attribute__((aligned(16))) uint8_t stage[kSoRleStage + 16u];
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x;
  const DecChunk dc = chunks[c];
  const uint32_t na = plan.n_adaptive;
  if (tid == 0) {
    secs_ok[c] = 0u;
    done_cnt[c] = 0u;
  }
  for (uint32_t a = tid; a < na; a += kSoThreads) {
    DecChunk z;
    z.src_off = 0;
    z.src_size = 0;
    z.n_points = 0;
    z.first_point = 0;
    z.cloud = 0;
    z.valid = 0;
    dsec[(size_t)a * n_chunks + c] = z;
  }
  if (!dc.valid || na == 0u || na > kSoMaxFields) return;
  if (merged != nullptr && merged[c] == 2u) return;  // the stream kernel wrote the chunk's integer fields with its points
  uint32_t off = reg_end[c];
  if (off == kDecRedo || off > dc.src_size) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  for (uint32_t a = 0; a < na; ++a) {  // uniform
    const uint32_t bpv = plan.adaptive[a].bpv;
    if (bpv > 4u || off >= src_size) return;
    const uint32_t mode = src[off];
    const uint32_t body = off + 1u;
    uint32_t end = 0xffffffffu;
    if (mode == 0u) {
      end = so_nth_end(src, src_size, body, n, sh);
    } else if (mode == 1u) {
      if (src_size - body >= 2u) {
        const uint32_t U = (uint32_t)src[body] | ((uint32_t)src[body + 1u] << 8);
        const uint64_t e = (uint64_t)body + 2ull + (uint64_t)U * bpv + ((uint64_t)palette_bits(U) * n + 7ull) / 8ull;
        if (U != 0u && e <= src_size) end = (uint32_t)e;
      }
    } else if (mode == 2u || mode == 3u) {
      if (src_size - body >= 4u) {
        const uint32_t runs = (uint32_t)src[body] | ((uint32_t)src[body + 1u] << 8) | ((uint32_t)src[body + 2u] << 16) | ((uint32_t)src[body + 3u] << 24);
        if (runs != 0u && runs <= n) {
          if (mode == 3u) {
            end = so_nth_end(src, src_size, body + 4u, 2u * runs, sh);
          } else if (runs <= 256u) {
            // raw values between the run lengths: the token ends cannot be read off the bytes; one lane parses an LDS copy
            const uint32_t nb = min(kSoRleStage, src_size - (body + 4u));
            __syncthreads();
            for (uint32_t i = tid; i < (nb + 15u) / 16u; i += kSoThreads) {
              uint32_t w[4];
              fp_load16u(src, src_size, body + 4u + i * 16u, w);
              *reinterpret_cast<uint4*>(stage + i * 16u) = make_uint4(w[0], w[1], w[2], w[3]);
            }
            __syncthreads();
            if (tid == 0) {
              uint32_t p = 0u;
              bool bad = false;
              for (uint32_t r = 0; r < runs && !bad; ++r) {
                p += bpv;
                for (uint32_t k = 0;; ++k) {
                  if (p >= nb || k >= 10u) { bad = true; break; }
                  if ((stage[p++] & 0x80u) == 0u) break;
                }
              }
              sh[1] = bad ? 0xffffffffu : body + 4u + p;
            }
            __syncthreads();
            end = sh[1];
            __syncthreads();
          }
        }
      }
    }
    if (end == 0xffffffffu || end > src_size || end < body) return;  // uniform: not sized -- the chunk stays with the old kernels
    if (tid == 0) {
      DecChunk s;
      s.src_off = dc.src_off + body;
      s.src_size = end - body;
      s.n_points = n;
      s.first_point = dc.first_point;
      s.cloud = dc.cloud;
      s.valid = mode + 1u;
      dsec[(size_t)a * n_chunks + c] = s;
    }
    off = end;
  }
  if (tid == 0 && off == src_size) secs_ok[c] = 1u;  // (trailing bytes: the serial decoder raises the error)
}

// grid = (n_chunks, n_adaptive), kSwsThreads threads: the Palette (valid == 2), Rle (3) and DeltaRle (4) sections, values
// written into the points. A section it cannot take (more than kSwsMaxPal entries / kSwsMaxRuns runs, anything irregular)
// does not count for its chunk, which then goes through the old kernels.
__device__ __forceinline__ void sections_dv_body(const DevPlan& plan, const uint8_t* __restrict__ streams, const DecChunk& dc, uint32_t c,
                                                 uint32_t a, uint32_t* __restrict__ done_cnt, const DecColumns& cols, uint32_t* tile,
                                                 uint32_t* vals, uint32_t* scan, uint32_t* flags);  // (below)

__global__ __launch_bounds__(kSwsThreads) void k_sections_w(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                            const DecChunk* __restrict__ dsec, uint32_t n_chunks,
                                                            uint8_t* __restrict__ out, uint32_t* __restrict__ done_cnt,
                                                            uint32_t to_cols, const DecColumns cols) {
  constexpr int T = (int)kSwsThreads;
  __shared__ __attribute__((aligned(16))) uint32_t tab[kSwsMaxPal];  // palette entries / run table (3 x kSwsMaxRuns + 16) / staged bytes
  __shared__ __attribute__((aligned(16))) uint16_t end_pos[kSwsMaxRuns * 2u + 16u];
  __shared__ uint32_t scan[48];
  __shared__ uint32_t flags[2];
  const uint32_t c = blockIdx.x, a = blockIdx.y;
  const uint32_t tid = threadIdx.x;
  const DecChunk ds = dsec[(size_t)a * n_chunks + c];
  if (ds.valid == 1u && to_cols >= 2u) {
    // to_cols = 2: the DeltaVarint fields of the grid as well (sections_dv_body, in the arrays of the other modes) -- the
    // workgroups of all modes share ONE launch instead of two half-idle ones in a row
    static_assert(kSwsThreads == kScfThreads && kSwsMaxPal >= kScfTileBytes, "the DeltaVarint body's values fit `tab`");
    static_assert(sizeof(end_pos) >= (kScfTileBytes / 4u + 8u) * 4u, "... and its tile `end_pos`");
    sections_dv_body(plan, streams, ds, c, a, done_cnt, cols, reinterpret_cast<uint32_t*>(end_pos), tab, scan, flags);
    return;
  }
  if (ds.valid < 2u) return;  // not sized, or DeltaVarint (the stream kernel's)
  const uint32_t mode = ds.valid - 1u;
  const uint8_t* src = streams + ds.src_off;
  const uint32_t size = ds.src_size;
  const uint32_t n = ds.n_points;
  const uint32_t bpv = plan.adaptive[a].bpv;
  // to_cols: the values go to the field's dense column (the point kernel behind this one merges them into the points:
  // every point is then written once); else straight into the points
  const uint32_t step = to_cols ? bpv : plan.point_step;
  uint8_t* base = to_cols ? const_cast<uint8_t*>(cols.p[a]) + (size_t)ds.first_point * bpv
                          : out + (size_t)ds.first_point * step + plan.adaptive[a].offset;
  if (tid == 0) flags[0] = 0u;
  __syncthreads();
  auto put = [&](uint32_t i, uint32_t v) __attribute__((always_inline)) {
    uint8_t* pt = base + __umul24(i, step);
    if (bpv == 2u) {
      const uint16_t h = (uint16_t)v;
      __builtin_memcpy(pt, &h, 2);
    } else {
      __builtin_memcpy(pt, &v, 4);
    }
  };
  if (mode == 1u) {
    // ---- Palette: [u16 U][U values][n indexes of bits(U) bits] -> entries to LDS, a thread unpacks runs of 8 indexes
    const uint32_t U = (uint32_t)src[0] | ((uint32_t)src[1] << 8);
    if (U > kSwsMaxPal) return;
    const uint32_t bits = palette_bits(U);
    for (uint32_t k = tid; k < U; k += T) {
      uint32_t v = 0u;
      for (uint32_t b = 0; b < bpv; ++b) v |= (uint32_t)src[2u + k * bpv + b] << (8u * b);
      tab[k] = v;
    }
    __syncthreads();
    const uint8_t* idx = src + 2u + U * bpv;
    const uint32_t idx_bytes = size - (2u + U * bpv);
    bool beyond = false;
    for (uint32_t i0 = tid * 8u; i0 < n; i0 += T * 8u) {
      // 8 indexes = 8 * bits <= 120 bits from bit i0 * bits (a multiple of 8 bits: a byte boundary)
      const uint32_t byte0 = i0 * bits / 8u;  // (i0 is a multiple of 8)
      uint32_t w[4] = {0u, 0u, 0u, 0u};
      if (byte0 + 16u <= idx_bytes) {
        __builtin_memcpy(w, idx + byte0, 16);
      } else {
        for (uint32_t b = 0; b < 16u && byte0 + b < idx_bytes; ++b) w[b >> 2] |= (uint32_t)idx[byte0 + b] << (8u * (b & 3u));
      }
#pragma unroll
      for (uint32_t j = 0; j < 8u; ++j) {
        if (i0 + j < n) {
          uint32_t ix = 0u;
          if (bits) {
            const uint32_t bo = j * bits, wi = bo >> 5, sh = bo & 31u;
            const uint64_t two = (((uint64_t)(wi < 3u ? w[wi + 1u] : 0u)) << 32) | w[wi];
            ix = (uint32_t)(two >> sh) & ((1u << bits) - 1u);
          }
          beyond = beyond || ix >= U;
          put(i0 + j, tab[min(ix, kSwsMaxPal - 1u)]);
        }
      }
    }
    if (beyond) flags[0] = 1u;  // an index beyond the palette: the serial decoder raises the error
    __syncthreads();
    if (tid == 0 && flags[0] == 0u) atomicAdd(done_cnt + c, 1u);
    return;
  }
  // ---- Rle / DeltaRle: [u32 runs][records] -> run table {first index, value before / value, difference} -> fill
  const uint32_t runs = (uint32_t)src[0] | ((uint32_t)src[1] << 8) | ((uint32_t)src[2] << 16) | ((uint32_t)src[3] << 24);
  if (runs == 0u || runs > kSwsMaxRuns || runs > n) return;
  const uint32_t sec_bytes = size - 4u;
  uint32_t* r_start = tab;                      // [runs + 1]
  uint32_t* r_base = tab + kSwsMaxRuns + 8u;    // [runs]
  uint32_t* r_diff = r_base + kSwsMaxRuns;      // [runs]
  if (mode == 3u) {
    // every token a varint, (difference, run length) pairs of at most 5 + 3 bytes: token ends numbered by a block scan over
    // tiles of kSwsThreads * 16 bytes; a run's first index and the value in front of it from two more scans
    if (sec_bytes > 16u * runs || sec_bytes > kSwsThreads * 16u * 4u) return;
    uint32_t n_tok = 0u;
    for (uint32_t t0 = 0; t0 < sec_bytes; t0 += kSwsThreads * 16u) {
      uint32_t b[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
      const uint32_t o = t0 + tid * 16u;
      if (o < sec_bytes) fp_load16u(src + 4u, sec_bytes, o, b);
      const uint32_t ends = wp_ends16(b);
      uint32_t total;
      const uint32_t tb = block_exclusive_scan<T>((uint32_t)__builtin_popcount(ends), scan, &total);
      uint32_t k = n_tok + tb;
      for (uint32_t m = ends; m; m &= m - 1u) {
        if (k < 2u * kSwsMaxRuns) end_pos[k] = (uint16_t)(o + (uint32_t)__builtin_ctz(m));
        ++k;
      }
      n_tok += total;
      __syncthreads();
    }
    if (n_tok != 2u * runs || (uint32_t)end_pos[n_tok - 1u] + 1u != sec_bytes) return;  // uniform
    // a thread takes 8 runs (16 tokens)
    uint32_t dif[8], len[8], s_len = 0u, s_val = 0u;
    bool bad = false;
    const uint8_t* body = src + 4u;
#pragma unroll
    for (uint32_t j = 0; j < 16u; ++j) {
      const uint32_t k = tid * 16u + j;
      uint64_t u = 0ull;
      if (k < n_tok) {
        const uint32_t start = k ? (uint32_t)end_pos[k - 1u] + 1u : 0u;
        const uint32_t tl = (uint32_t)end_pos[k] - start + 1u;
        bad = bad || tl > 5u;
        for (uint32_t b = 0; b < tl && b < 5u; ++b) u |= (uint64_t)(body[start + b] & 0x7fu) << (7u * b);
      }
      if ((j & 1u) == 0u) {
        bad = bad || (k < n_tok && u == 0ull);
        const uint64_t u1 = u - 1ull;
        dif[j >> 1] = k < n_tok ? (uint32_t)((u1 >> 1) ^ (0ull - (u1 & 1ull))) : 0u;
      } else {
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
    if (flags[0] || tot_len != n) return;  // uniform
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
    if (runs > 256u) return;
    if (tid == 0) {
      const uint8_t* p = src + 4u;
      const uint8_t* end = p + sec_bytes;
      uint32_t idx = 0u;
      bool bad = false;
      for (uint32_t r = 0; r < runs && !bad; ++r) {
        if ((uint32_t)(end - p) < bpv) { bad = true; break; }
        uint32_t base_v = 0u;
        for (uint32_t k = 0; k < bpv; ++k) base_v |= (uint32_t)p[k] << (8u * k);
        p += bpv;
        uint32_t len = 0u, sh = 0u;  // readUVarint, src/v5_codec.cpp:176-194
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
  for (uint32_t i0 = tid * 8u; i0 < n; i0 += T * 8u) {
    uint32_t lo = 0u, hi = runs;  // last r with r_start[r] <= i0
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (r_start[mid] <= i0) lo = mid;
      else hi = mid;
    }
    uint32_t nxt = r_start[lo + 1u], st = r_start[lo], bs = r_base[lo], df = r_diff[lo];
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) {
      const uint32_t i = i0 + j;
      if (i < n) {
        if (i >= nxt) {  // (runs are never empty: one step)
          ++lo;
          st = nxt; nxt = r_start[lo + 1u]; bs = r_base[lo]; df = r_diff[lo];
        }
        put(i, mode == 3u ? bs + df * (i - st + 1u) : bs);
      }
    }
  }
  if (tid == 0) atomicAdd(done_cnt + c, 1u);
}

// k_sections_dv_cols: the DeltaVarint sections k_section_offsets sized (dsec record with valid == 1) -> the field's dense
// column, grid (chunks, fields) x 256 threads. The stream kernel's section mode decodes the same sections with its two walks
// and chains (289 us for the two DeltaVarint fields of an Ouster-style batch); a section is ONE integer op, and what
// k_sections_cols_fast does for the plans with one integer field is all it takes: slices of 4 KiB in sequence, a thread
// decodes the tokens that END in its 16 bytes (the first one begins behind the last end among the 8 bytes in front), two block
// scans number the tokens and add up the differences, the running count and sum stay in registers from slice to slice.
// A section that is not exactly n tokens, a token of more than 5 bytes or a marker byte leaves the field to the kernels
// behind (done_cnt is not raised).
// (LDS handed in: k_sections_w runs this body for the DeltaVarint fields of its grid in the arrays it has for the other modes)
__device__ __forceinline__ void sections_dv_body(const DevPlan& plan, const uint8_t* __restrict__ streams, const DecChunk& dc, uint32_t c,
                                                 uint32_t a, uint32_t* __restrict__ done_cnt, const DecColumns& cols,
                                                 uint32_t* tile /* [kScfTileBytes / 4 + 8], 16-byte aligned */,
                                                 uint32_t* vals /* [kScfTileBytes] */, uint32_t* scan /* [40] */, uint32_t* flags) {
  constexpr int T = (int)kScfThreads;
  const uint32_t tid = threadIdx.x;
  const uint8_t* src = streams + dc.src_off;  // the section's body (behind the mode byte)
  const uint32_t src_size = dc.src_size;
  const uint32_t n = dc.n_points;
  const uint32_t bpv = plan.adaptive[a].bpv;
  if (n == 0u || src_size == 0u || bpv > 4u || cols.p[a] == nullptr) return;
  uint8_t* col = const_cast<uint8_t*>(cols.p[a]) + (size_t)dc.first_point * bpv;
  const uint32_t n_slices = (src_size + kScfTileBytes - 1u) / kScfTileBytes;
  if (tid == 0) flags[0] = 0u;
  __syncthreads();
  uint32_t pre_cnt = 0u, pre_sum = 0u;
  for (uint32_t s = 0; s < n_slices; ++s) {
    const uint32_t base = s * kScfTileBytes;
    uint32_t b[4];
    fp_load16u(src, src_size, base + tid * 16u, b);  // bytes behind the section read 0xff: no token ends
    *reinterpret_cast<uint4*>(tile + 2u + tid * 4u) = make_uint4(b[0], b[1], b[2], b[3]);
    if (tid == 0u) {  // the 8 bytes in front of the slice (a token has 5 at most); in front of the body: the mode byte, an end
      uint32_t h[2] = {0u, 0u};
      if (base >= 8u) __builtin_memcpy(h, src + base - 8u, 8);
      tile[0] = h[0];
      tile[1] = h[1];
    }
    if (tid < 4u) tile[2u + kScfTileBytes / 4u + tid] = 0xffffffffu;
    const uint32_t ends = fp_ends16(b);
    const uint32_t my_cnt = (uint32_t)__builtin_popcount(ends);
    uint32_t n_tile;
    const uint32_t tb = block_exclusive_scan<T>(my_cnt, scan, &n_tile);  // barrier inside: the tile is complete
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
        const uint32_t lo = sh ? ((w0 >> sh) | (w1 << (32u - sh))) : w0;
        const uint32_t b4 = (sh ? ((w1 >> sh) | (w2 << (32u - sh))) : w1) & 0xffu;
        const uint32_t g = (lo & 0x7fu) | (((lo >> 8) & 0x7fu) << 7) | (((lo >> 16) & 0x7fu) << 14) | (((lo >> 24) & 0x7fu) << 21);
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
    if (flags[0] != 0u) return;              // (uniform)
    if (pre_cnt + n_tile > n) return;        // more tokens than points
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
    __syncthreads();  // (tile, vals and flags are written again)
  }
  // every point has its token and the section's last byte ends one
  if (tid == 0u && pre_cnt == n && (src[src_size - 1u] & 0x80u) == 0u) atomicAdd(done_cnt + c, 1u);
}

__global__ __launch_bounds__(kScfThreads) void k_sections_dv_cols(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                  const DecChunk* __restrict__ dsec, uint32_t n_chunks,
                                                                  uint32_t* __restrict__ done_cnt, const DecColumns cols) {
  __shared__ __attribute__((aligned(16))) uint32_t tile[kScfTileBytes / 4u + 8u];  // 8 bytes of history, the slice's bytes, slack
  __shared__ uint32_t vals[kScfTileBytes];
  __shared__ uint32_t scan[40];
  __shared__ uint32_t flags[2];
  const uint32_t c = blockIdx.x, a = blockIdx.y;
  const DecChunk dc = dsec[(size_t)a * n_chunks + c];
  if (dc.valid != 1u) return;  // another mode: k_sections_w
  sections_dv_body(plan, streams, dc, c, a, done_cnt, cols, tile, vals, scan, flags);
}

// grid = ceil(n_chunks / 256): sec_done[c] = 1 for the chunks whose sections all arrived
__global__ __launch_bounds__(256) void k_sections_done(uint32_t n_chunks, uint32_t n_adaptive, const uint8_t* __restrict__ secs_ok,
                                                       const uint32_t* __restrict__ done_cnt, uint8_t* __restrict__ sec_done,
                                                       uint32_t* __restrict__ status, uint32_t count_stat) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= n_chunks) return;
  const bool ok = secs_ok[c] != 0u && done_cnt[c] == n_adaptive;
  if (count_stat >= 2u && sec_done[c] == 2u) return;  // (count_stat 2: chunks the stream kernel merged stay marked)
  sec_done[c] = ok ? 1u : 0u;  // (column mode: this is the point kernel's sec_cols flag, which does the counting itself)
  if (ok && count_stat) atomicAdd(&status[kStatFastSections], 1u);
}

}  // namespace cldn
