// lz4_kernels.hip -- stage 2 on the device: an LZ4 *block* per 32768-point chunk.
//
// What it replaces: CompressChunk (src/codec_common.cpp:220-258, LZ4_compress_default at :232-234) fed chunk by chunk by
// WriteStage1Chunk (src/chunk_writer.cpp:27-48). The bytes are NOT those of lz4's own compressor (they need not be:
// any valid block decodes to the same payload); what is guaranteed is that DecompressChunk's LZ4_decompress_safe
// (src/codec_common.cpp:260-299) returns the exact stage-1 payload. The format follows the published block format:
// sequences [token][literal-length bytes][literals][offset u16 LE][match-length bytes], a last sequence of literals
// only, the last 5 bytes literals, no match starting within the last 12 bytes.
//
// The algorithm is deterministic and restated serially in oracle/lz4_model.c (the GPU tests ask for byte equality):
//   k_lz4_match   one wave per 16 KiB sub-range of a chunk's payload: the sub-range is staged in LDS next to a hash table
//                 of its own positions (4096 entries, atomicMax = the most recent position wins), so every byte the
//                 parser touches is an LDS access. Per step the 64 lanes look at 64 consecutive positions, the first
//                 lane with a verified 4-byte match wins, the wave extends it cooperatively (64 bytes per compare),
//                 enters the positions up to the match start and jumps behind the match. Matches never leave their
//                 sub-range; output = the ordered list of (position, length, offset) per sub-range.
//   k_lz4_emit    one workgroup per chunk: sequence sizes from the match lists (a sequence's literals start where the
//                 previous match ended, whichever sub-range that was in), their prefix sum, then headers by lanes and
//                 literals by whole waves, the trailing literals by the whole workgroup.
// The compressed chunks are left in per-chunk slots of their worst-case size; k_finish (stage1_finish.h) frames them as
// [u32 size][block] exactly as it frames stage-1 payloads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cloudini_hip.h"
#include "stage1_device.h"
#include "stage1_launch.h"

namespace cldn {

namespace {

constexpr uint32_t kLzHashMul = 2654435761u;
constexpr uint32_t kLzTableSize = 1u << kLzHashBits;
constexpr uint32_t kLzEmitThreads = 256;

__device__ __forceinline__ uint32_t lz_wave_excl_scan(uint32_t x, uint32_t lane, uint32_t* total) {
  uint32_t incl = x;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
    if (lane >= (uint32_t)d) incl += o;
  }
  *total = (uint32_t)__shfl((int)incl, 63);
  return incl - x;
}

__device__ __forceinline__ uint32_t lz_ext_bytes(uint32_t x) { return x >= 15u ? (x - 15u) / 255u + 1u : 0u; }

// the part of a length above 15 as 255, 255, ..., rest (x >= 15); returns the bytes written
__device__ __forceinline__ uint32_t lz_put_ext(uint8_t* o, uint32_t x) {
  x -= 15u;
  uint32_t k = 0u;
  while (x >= 255u) {
    o[k++] = 255u;
    x -= 255u;
  }
  o[k++] = (uint8_t)x;
  return k;
}

// 4 bytes at any byte offset of an LDS dword array
__device__ __forceinline__ uint32_t lz_lds_u32(const uint32_t* base, uint32_t byte_off) {
  const uint32_t i = byte_off >> 2;
  const uint32_t lo = base[i], hi = base[i + 1u];
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> ((byte_off & 3u) * 8u));
}

// n bytes, any alignment on both sides, by `nthreads` threads (this thread = t): dwords through unaligned accesses
__device__ __forceinline__ void lz_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t n, uint32_t t,
                                        uint32_t nthreads) {
  const uint32_t n4 = n >> 2;
  for (uint32_t i = t; i < n4; i += nthreads) {
    uint32_t w;
    __builtin_memcpy(&w, src + 4u * i, 4);
    __builtin_memcpy(dst + 4u * i, &w, 4);
  }
  const uint32_t done = n4 << 2;
  if (t < (n & 3u)) dst[done + t] = src[done + t];
}

}  // namespace

// grid = n_chunks * subs_per_chunk workgroups of one wave
__global__ __launch_bounds__(64) void k_lz4_match(const uint8_t* __restrict__ stream, const uint64_t* __restrict__ chunk_dst,
                                                  const uint32_t* __restrict__ chunk_payload, uint32_t subs_per_chunk,
                                                  LzMatch* __restrict__ matches, uint32_t* __restrict__ counts) {
  __shared__ uint32_t data[kLzSubBytes / 4u + 4u];  // the sub-range (+ slack for the straddling dword reads)
  __shared__ uint32_t table[kLzTableSize];          // position inside the sub-range + 1; 0 = free
  const uint32_t lane = threadIdx.x;
  const uint32_t c = blockIdx.x / subs_per_chunk, k = blockIdx.x % subs_per_chunk;
  const uint32_t n = chunk_payload[c];
  const uint32_t s = k * kLzSubBytes;
  if (s >= n) return;  // (k_lz4_emit only looks at the sub-ranges the payload has)
  const uint32_t e = n - s < kLzSubBytes ? n : s + kLzSubBytes;
  const uint8_t* in = stream + chunk_dst[c] + 4u;
  LzMatch* out = matches + (size_t)blockIdx.x * kLzMaxMatches;
  uint32_t count = 0u;

  if (n >= 13u && e - s >= 4u) {
    {  // stage [s, e) in LDS: unaligned dword loads, aligned LDS stores; bytes behind e read as 0
      const uint32_t bytes = e - s;
      const uint8_t* src = in + s;
      for (uint32_t i = lane; i < kLzSubBytes / 4u + 4u; i += 64u) {
        uint32_t w = 0u;
        if (4u * i + 4u <= bytes) {
          __builtin_memcpy(&w, src + 4u * i, 4);
        } else if (4u * i < bytes) {
          for (uint32_t b = 0; 4u * i + b < bytes; ++b) w |= (uint32_t)src[4u * i + b] << (8u * b);
        }
        data[i] = w;
      }
      for (uint32_t i = lane; i < kLzTableSize; i += 64u) table[i] = 0u;
    }
    __syncthreads();
    // positions below are relative to s
    const int32_t last_start = min((int32_t)(e - s) - 4, (int32_t)n - 12 - (int32_t)s);  // last position a match may start at
    const uint32_t end_limit = min(e, n - 5u) - s;                                        // where a match ends at the latest
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(data);
    int32_t i = 0;
    while (i <= last_start && count < kLzMaxMatches) {
      const int32_t p = i + (int32_t)lane;
      const bool active = p <= last_start;
      const uint32_t seq = lz_lds_u32(data, active ? (uint32_t)p : 0u);
      const uint32_t h = (seq * kLzHashMul) >> (32u - kLzHashBits);
      const uint32_t cand = active ? table[h] : 0u;
      const bool ok = cand != 0u && lz_lds_u32(data, cand - 1u) == seq;
      const uint64_t m = __ballot(ok);
      const uint32_t f = m ? (uint32_t)__builtin_ctzll(m) : 63u;
      // (the lanes have read the table: the LDS operations of a wave are performed in order)
      if (active && lane <= f) atomicMax(&table[h], (uint32_t)p + 1u);
      if (m == 0ull) {
        i += 64;
        continue;
      }
      const uint32_t pm = (uint32_t)i + f;
      const uint32_t cf = (uint32_t)__shfl((int)cand, (int)f) - 1u;
      uint32_t len = 4u;
      const uint32_t maxlen = end_limit - pm;
      while (len < maxlen) {  // 64 bytes per compare
        const uint32_t q = len + lane;
        const bool differ = q >= maxlen || bytes[pm + q] != bytes[cf + q];
        const uint64_t d = __ballot(differ);
        const uint32_t same = d ? (uint32_t)__builtin_ctzll(d) : 64u;
        len += same;
        if (same < 64u) break;
      }
      if (lane == 0u) {
        LzMatch rec;
        rec.pos = s + pm;
        rec.len = (uint16_t)len;  // <= kLzSubBytes = 16384
        rec.off = (uint16_t)(pm - cf);
        out[count] = rec;
      }
      ++count;
      i = (int32_t)(pm + len);
    }
  }
  if (lane == 0u) counts[blockIdx.x] = count;
}

// grid = n_chunks
__global__ __launch_bounds__(kLzEmitThreads) void k_lz4_emit(const uint8_t* __restrict__ stream, const uint64_t* __restrict__ chunk_dst,
                                                             const uint32_t* __restrict__ chunk_payload, uint32_t subs_per_chunk,
                                                             const LzMatch* __restrict__ matches, const uint32_t* __restrict__ counts,
                                                             uint8_t* __restrict__ out_slots, uint64_t out_stride,
                                                             Seg* __restrict__ out_segs) {
  __shared__ uint32_t cnt[kLzMaxSubs], anchor_in[kLzMaxSubs], sub_size[kLzMaxSubs], out_base[kLzMaxSubs];
  __shared__ uint32_t tail_anchor, tail_at;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr uint32_t NW = kLzEmitThreads / 64u;
  const uint32_t c = blockIdx.x;
  const uint32_t n = chunk_payload[c];
  const uint8_t* in = stream + chunk_dst[c] + 4u;
  uint8_t* out = out_slots + (size_t)c * out_stride;
  uint32_t nsub = (n + kLzSubBytes - 1u) / kLzSubBytes;
  const LzMatch* mbase = matches + (size_t)c * subs_per_chunk * kLzMaxMatches;
  // (a payload with more sub-ranges than the tables hold -- beyond 16 MiB -- leaves as literals only: valid, not smaller)
  const bool use_matches = nsub <= kLzMaxSubs;
  if (!use_matches) nsub = 0u;

  // A: matches per sub-range, and where the last one ends
  for (uint32_t k = tid; k < nsub; k += kLzEmitThreads) {
    const uint32_t m = counts[(size_t)c * subs_per_chunk + k];
    cnt[k] = m;
    uint32_t last_end = 0u;
    if (m) {
      const LzMatch r = mbase[(size_t)k * kLzMaxMatches + m - 1u];
      last_end = r.pos + r.len;
    }
    sub_size[k] = last_end;  // (parked here until step B has read it)
  }
  __syncthreads();
  // B: a sub-range's first sequence takes its literals from the end of the last match before it
  if (tid == 0u) {
    uint32_t a = 0u;
    for (uint32_t k = 0; k < nsub; ++k) {
      anchor_in[k] = a;
      if (cnt[k]) a = sub_size[k];
    }
    tail_anchor = a;
  }
  __syncthreads();
  // C: bytes of the sequences of every sub-range
  auto seq_fields = [&](uint32_t k, uint32_t j, uint32_t& lit, uint32_t& ml, uint32_t& anchor, LzMatch& r) {
    const LzMatch* list = mbase + (size_t)k * kLzMaxMatches;
    r = list[j];
    if (j) {
      const LzMatch q = list[j - 1u];
      anchor = q.pos + q.len;
    } else {
      anchor = anchor_in[k];
    }
    lit = r.pos - anchor;
    ml = (uint32_t)r.len - 4u;
  };
  for (uint32_t k = wave; k < nsub; k += NW) {
    uint32_t acc = 0u;
    for (uint32_t j = lane; j < cnt[k]; j += 64u) {
      uint32_t lit, ml, anchor;
      LzMatch r;
      seq_fields(k, j, lit, ml, anchor, r);
      acc += 1u + lz_ext_bytes(lit) + lit + 2u + lz_ext_bytes(ml);
    }
    uint32_t total;
    (void)lz_wave_excl_scan(acc, lane, &total);
    if (lane == 0u) sub_size[k] = total;
  }
  __syncthreads();
  // D: where every sub-range's sequences go
  if (tid == 0u) {
    uint32_t o = 0u;
    for (uint32_t k = 0; k < nsub; ++k) {
      out_base[k] = o;
      o += sub_size[k];
    }
    tail_at = o;
  }
  __syncthreads();
  // E: the sequences. Headers and match fields by their lanes, literals by the whole wave
  for (uint32_t k = wave; k < nsub; k += NW) {
    uint32_t run = out_base[k];
    const uint32_t m = cnt[k];
    for (uint32_t j0 = 0; j0 < m; j0 += 64u) {
      const uint32_t j = j0 + lane;
      uint32_t lit = 0u, ml = 0u, anchor = 0u, size = 0u;
      LzMatch r;
      r.pos = 0u;
      r.len = 4u;
      r.off = 0u;
      if (j < m) {
        seq_fields(k, j, lit, ml, anchor, r);
        size = 1u + lz_ext_bytes(lit) + lit + 2u + lz_ext_bytes(ml);
      }
      uint32_t tile_total;
      const uint32_t at = run + lz_wave_excl_scan(size, lane, &tile_total);
      uint32_t lit_at = at;
      if (j < m) {
        uint8_t* o = out + at;
        *o++ = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (ml < 15u ? ml : 15u));
        if (lit >= 15u) o += lz_put_ext(o, lit);
        lit_at = (uint32_t)(o - out);
        o += lit;
        *o++ = (uint8_t)(r.off & 0xffu);
        *o++ = (uint8_t)(r.off >> 8);
        if (ml >= 15u) (void)lz_put_ext(o, ml);
      }
      const uint32_t in_tile = min(64u, m - j0);
      for (uint32_t q = 0; q < in_tile; ++q) {
        const uint32_t ql = (uint32_t)__shfl((int)lit, (int)q);
        if (ql == 0u) continue;  // uniform
        const uint32_t qa = (uint32_t)__shfl((int)anchor, (int)q), qo = (uint32_t)__shfl((int)lit_at, (int)q);
        lz_copy(in + qa, out + qo, ql, lane, 64u);
      }
      run += tile_total;
    }
  }
  // F: the last sequence, literals only
  {
    const uint32_t a = tail_anchor, lit = n - a;
    uint8_t* o = out + tail_at;
    const uint32_t hdr = 1u + lz_ext_bytes(lit);
    if (tid == 0u) {
      *o = (uint8_t)((lit < 15u ? lit : 15u) << 4);
      if (lit >= 15u) (void)lz_put_ext(o + 1u, lit);
      Seg sg;
      sg.off = 0u;
      sg.size = tail_at + hdr + lit;
      out_segs[c] = sg;
    }
    lz_copy(in + a, o + hdr, lit, tid, kLzEmitThreads);
  }
}

int lz4_launch(const Lz4Launch& L) {
  if (L.n_chunks == 0u) return CLDN_HIP_OK;
  hipLaunchKernelGGL(k_lz4_match, dim3(L.n_chunks * L.subs_per_chunk), dim3(64), 0, L.stream, L.stage1, L.chunk_dst, L.chunk_payload,
                     L.subs_per_chunk, L.matches, L.counts);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return launch_fail(e, "k_lz4_match");
  hipLaunchKernelGGL(k_lz4_emit, dim3(L.n_chunks), dim3(kLzEmitThreads), 0, L.stream, L.stage1, L.chunk_dst, L.chunk_payload,
                     L.subs_per_chunk, L.matches, L.counts, L.out_slots, L.out_stride, L.out_segs);
  if ((e = hipGetLastError()) != hipSuccess) return launch_fail(e, "k_lz4_emit");
  return CLDN_HIP_OK;
}

}  // namespace cldn
