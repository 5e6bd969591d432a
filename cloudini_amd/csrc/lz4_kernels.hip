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
//   k_lz4_plan    sub-ranges per chunk -> compact numbering (one scan).
//   k_lz4_match   one wave per 8 KiB sub-range of a chunk's payload (CLDN_HIP_STAGE2_LZ4_FAST: 4 KiB): the sub-range is
//                 staged in LDS next to a hash table of its own positions (2048 / 1024 entries, atomicMax = the most recent
//                 position wins), so every byte the parser touches is an LDS access. Per step the 64 lanes look at 64
//                 consecutive positions: every lane with a verified 4-byte hit extends its own match (4 bytes per round; the
//                 whole wave finishes what is still open after 3 rounds), all 64 positions enter the table, and the step's
//                 matches are taken greedily in position order (a scalar loop over lengths; the chosen lanes write their
//                 records side by side). Matches never leave their sub-range; a step is only parsed while the list has room
//                 for the 16 matches a step can yield. Round 5: the loop is predicated vector code -- the CU's scalar unit
//                 was what the round-4 version waited for.
//   k_lz4_sizes   per sub-range: bytes of its sequences (a sequence's literals start where the previous match ended,
//                 whichever sub-range that was in).
//   k_lz4_layout  per chunk: the sequence bytes in front of every sub-range, the first match behind it, the block's size.
//   k_lz4_offsets one scan over the chunks: where every [u32 size][block] goes in the caller's output, the size headers, the
//                 stream offsets of the clouds (what k_finish does for stage-1 payloads).
//   k_lz4_emit_lds  per sub-range: the sub-range staged in LDS, its sequences assembled in an LDS image of the output span it
//                 owns, 16-byte stores STRAIGHT INTO THE FRAMED STREAMS (round 5: no block slots, no second framing pass).
//                 Every literal byte is copied by the wave of the sub-range it lies in, however long a literal run is.
//                 k_lz4_emit (CLDN_HIP_LZ4_EMIT_DIRECT=1) is the round-3 kernel without the LDS image: the fallback for a
//                 span that does not fit the image, and the A/B reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "cloudini_hip.h"
#include "stage1_device.h"
#include "stage1_launch.h"

namespace cldn {

namespace {

constexpr uint32_t kLzHashMul = 2654435761u;

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
  return __builtin_amdgcn_alignbyte(hi, lo, byte_off);  // (v_alignbyte_b32 uses the two low bits of the shift)
}

}  // namespace

// n bytes, any alignment, by one wave: 16 bytes per lane and step through unaligned accesses
__device__ __forceinline__ void lz_copy_wave(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t n, uint32_t lane) {
  const uint32_t n16 = n >> 4;
  for (uint32_t i = lane; i < n16; i += 64u) {
    uint4 w;
    __builtin_memcpy(&w, src + 16u * i, 16);
    __builtin_memcpy(dst + 16u * i, &w, 16);
  }
  const uint32_t done = n16 << 4;
  if (lane < (n & 15u)) dst[done + lane] = src[done + lane];
}

// chunk of compact sub-range index `idx`: the last c with sub_first[c] <= idx
__device__ __forceinline__ uint32_t lz_chunk_of(const uint32_t* __restrict__ sub_first, uint32_t n_chunks, uint32_t idx) {
  uint32_t lo = 0u, hi = n_chunks;  // invariant: sub_first[lo] <= idx < sub_first[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (sub_first[mid] <= idx) lo = mid;
    else hi = mid;
  }
  return lo;
}

// one workgroup: sub_first[c] = sub-ranges of the chunks before c (sub_first[n_chunks] = all of them)
// (a chunk without a byte of payload has no sub-range and no wave below: its block, the single token 0x00, is written by k_lz4_offsets)
// max_subs: what the match lists and the per-sub-range arrays hold. The plan never numbers a sub-range beyond it (chunks that
// would need more get none, ST_OUT_OVERFLOW is raised): sizes that do not fit the workspace -- stale ones, say, read behind a
// stage 1 that gave up -- cannot make the kernels below write out of bounds.
__global__ __launch_bounds__(1024) void k_lz4_plan(const uint32_t* __restrict__ chunk_payload, uint32_t n_chunks,
                                                   uint32_t* __restrict__ sub_first, uint32_t sub_bytes, uint32_t max_subs,
                                                   uint32_t* __restrict__ status) {
  __shared__ uint32_t wtot[16];
  __shared__ uint32_t carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0u) carry = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < n_chunks; base += 1024u) {
    const uint32_t c = base + tid;
    const uint32_t mine = c < n_chunks ? (chunk_payload[c] + sub_bytes - 1u) / sub_bytes : 0u;
    uint32_t wsum;
    const uint32_t excl = lz_wave_excl_scan(mine, lane, &wsum);
    if (lane == 0u) wtot[wave] = wsum;
    __syncthreads();
    uint32_t before = carry;
    for (uint32_t w = 0; w < wave; ++w) before += wtot[w];
    if (c < n_chunks) sub_first[c] = min(before + excl, max_subs);
    __syncthreads();
    if (tid == 1023u) carry = before + excl + mine;
    __syncthreads();
  }
  if (tid == 0u) {
    sub_first[n_chunks] = min(carry, max_subs);
    if (carry > max_subs) atomicOr(status, ST_OUT_OVERFLOW);
  }
}

// workgroups of one wave; every workgroup takes the sub-ranges blockIdx.x, blockIdx.x + gridDim.x, ...
// SUB / HASH_BITS / MAX_MATCHES: kLzSubBytes, kLzHashBits, kLzMaxMatches, or the kLzFast* set
template <uint32_t SUB, uint32_t HASH_BITS, uint32_t MAX_MATCHES>
__global__ __launch_bounds__(64) void k_lz4_match(const uint8_t* __restrict__ stream, const uint64_t* __restrict__ chunk_dst,
                                                  const uint32_t* __restrict__ chunk_payload, uint32_t n_chunks,
                                                  const uint32_t* __restrict__ sub_first, LzMatch* __restrict__ matches,
                                                  uint32_t* __restrict__ counts, uint32_t* __restrict__ last_end,
                                                  uint32_t* __restrict__ sub_chunk) {
  constexpr uint32_t kTable = 1u << HASH_BITS;
  __shared__ __attribute__((aligned(16))) uint32_t data[SUB / 4u + 8u];  // the sub-range (+ slack for the straddling dword reads)
  // position inside the sub-range + 1; 0 = free. (Round 5, measured and dropped: 16-bit entries -- 12.3 KB of LDS per wave,
  // 13 waves per CU instead of 9 -- with a store / read back / store-again loop in place of the 32-bit atomicMax: byte-equal
  // blocks, 8 % SLOWER (2.43 -> 2.62 ms on 32 x 1 M XYZI): the step's extra LDS round trip costs more than the waves return)
  __shared__ uint32_t table[kTable];
  const uint32_t lane = threadIdx.x;
  const uint32_t total = sub_first[n_chunks];
  for (uint32_t idx = blockIdx.x; idx < total; idx += gridDim.x) {
    const uint32_t c = lz_chunk_of(sub_first, n_chunks, idx);
    const uint32_t k = idx - sub_first[c];
    const uint32_t n = chunk_payload[c];
    const uint32_t s = k * SUB;
    const uint32_t e = n - s < SUB ? n : s + SUB;
    const uint8_t* in = stream + chunk_dst[c] + 4u;
    LzMatch* out = matches + (size_t)idx * MAX_MATCHES;
    uint32_t count = 0u, lend = 0u;
    __syncthreads();  // the previous sub-range's LDS contents are done with
    if (n >= 13u && e - s >= 4u) {
      {  // stage [s, e) in LDS: unaligned dword loads, aligned LDS stores; bytes behind e read as 0
        const uint32_t bytes = e - s;
        const uint8_t* src = in + s;
        const uint32_t full = bytes >> 4;  // 16 bytes per lane and load (unaligned dwordx4), all loads before the stores
        constexpr uint32_t kRounds = SUB / 16u / 64u;
        uint4 w[kRounds];
#pragma unroll
        for (uint32_t r = 0; r < kRounds; ++r) {
          const uint32_t i = r * 64u + lane;
          w[r] = make_uint4(0u, 0u, 0u, 0u);
          if (i < full) __builtin_memcpy(&w[r], src + 16u * i, 16);
        }
#pragma unroll
        for (uint32_t r = 0; r < kRounds; ++r) reinterpret_cast<uint4*>(data)[r * 64u + lane] = w[r];
        if (lane < 2u) reinterpret_cast<uint4*>(data)[SUB / 16u + lane] = make_uint4(0u, 0u, 0u, 0u);  // the slack
        __syncthreads();
        if (lane < (bytes & 15u)) reinterpret_cast<uint8_t*>(data)[(full << 4) + lane] = src[(full << 4) + lane];  // last partial unit
        for (uint32_t i = lane; i < kTable; i += 64u) table[i] = 0u;
      }
      __syncthreads();
      // positions below are relative to s
      const int32_t last_start = min((int32_t)(e - s) - 4, (int32_t)n - 12 - (int32_t)s);  // last position a match may start at
      const uint32_t end_limit = min(e, n - 5u) - s;                                        // where a match ends at the latest
      const uint8_t* bytes = reinterpret_cast<const uint8_t*>(data);
      // One step = 64 consecutive positions. (Round 5: the scalar unit of the CU was what this loop waited for -- 113 scalar
      // instructions per step against 58 vector ones, `SQ_INSTS_SALU`: the per-lane work is now predicated vector code
      // without exec-mask changes, matches still open after the lanes' rounds are finished in front of the selection, and
      // the selection loop only reads lengths.)
      // (A step yields 16 matches at most -- they do not overlap and have 4 bytes or more --, and a step is only parsed
      // while the list has room for 16: the selection loop needs no check of its own.)
      static_assert(MAX_MATCHES >= 16u, "a step may take 16 matches");
      // (Measured and dropped: the next step's look-ups -- 4 bytes, table entry, verify read -- issued while this step's matches
      // are selected: no change, 1.53 ms either way for 8 KiB windows.)
      int32_t i = 0;
      while (i <= last_start && count + 16u <= MAX_MATCHES) {
        const int32_t p = i + (int32_t)lane;
        const bool active = p <= last_start;
        // (a lane behind the last position repeats it: the same look-up, the same table entry written again -- no hit)
        const uint32_t pc = (uint32_t)min(p, last_start);
        const uint32_t seq = lz_lds_u32(data, pc);
        const uint32_t h = (seq * kLzHashMul) >> (32u - HASH_BITS);
        const uint32_t cand = table[h];
        const uint32_t cm1 = cand != 0u ? cand - 1u : 0u;
        const uint32_t vseq = lz_lds_u32(data, cm1);  // (read by every lane: no exec-mask change)
        const bool ok = (int)active & (int)(cand != 0u) & (int)(vseq == seq);
        // every hit extends its own match, 4 bytes per round, up to kLaneRounds rounds
        constexpr uint32_t kLaneRounds = 3u;
        const uint32_t maxlen = ok ? end_limit - pc : 0u;
        uint32_t len = ok ? min(4u, maxlen) : 0u;
        bool going = ok && len < maxlen;
        for (uint32_t r = 0; r < kLaneRounds; ++r) {
          if (__ballot(going) == 0ull) break;
          const uint32_t x = lz_lds_u32(data, pc + len) ^ lz_lds_u32(data, cm1 + len);
          const uint32_t adv = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
          const uint32_t nl = min(len + adv, maxlen);
          const bool on = going && x == 0u && nl < maxlen;
          len = going ? nl : len;
          going = on;
        }
        // (the lanes have read the table: the LDS operations of a wave are performed in order)
        atomicMax(&table[h], pc + 1u);
        const uint64_t okmask = __ballot(ok);
        // what is still open is finished by the whole wave, 64 bytes per compare
        for (uint64_t om = __ballot(going); om != 0ull; om &= om - 1ull) {
          const uint32_t f = (uint32_t)__builtin_ctzll(om);
          uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)f);
          const uint32_t cf = (uint32_t)__builtin_amdgcn_readlane((int)cm1, (int)f);
          const uint32_t pm = (uint32_t)i + f;
          const uint32_t ml = end_limit - pm;
          while (L < ml) {
            const uint32_t q = L + lane;
            const bool differ = q >= ml || bytes[pm + q] != bytes[cf + q];
            const uint64_t d = __ballot(differ);
            const uint32_t same = d ? (uint32_t)__builtin_ctzll(d) : 64u;
            L += same;
            if (same < 64u) break;
          }
          if (lane == f) len = L;
        }
        // the step's matches, greedily in position order: a scalar loop over the lengths (rem = the hits at and behind
        // lane `cur`, shifted down by cur); the records are then written by the chosen lanes themselves, side by side
        uint32_t cur = 0u;
        if (okmask != 0ull) {
          uint64_t taken = 0ull, rem = okmask;
          do {
            const uint32_t f = cur + (uint32_t)__builtin_ctzll(rem);
            const uint32_t L = (uint32_t)__builtin_amdgcn_readlane((int)len, (int)f);
            taken |= 1ull << f;
            cur = f + L;
            rem = cur < 64u ? okmask >> cur : 0ull;
          } while (rem != 0ull);
          lend = s + (uint32_t)i + cur;
          if ((taken >> lane) & 1ull) {
            const uint32_t k = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(taken >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)taken, 0u));
            LzMatch rec;
            rec.pos = s + pc;
            rec.len = (uint16_t)len;  // <= kLzSubBytes
            rec.off = (uint16_t)(pc - cm1);
            out[k] = rec;
          }
          count += (uint32_t)__builtin_popcountll(taken);
        }
        i += (int32_t)max(64u, cur);
      }
    }
    if (lane == 0u) {
      counts[idx] = count;
      last_end[idx] = lend;  // 0 = no match (a match never ends at 0)
      sub_chunk[idx] = c;    // (the kernels behind this one do not search again)
    }
  }
}

namespace {
// fields of sequence j of sub-range idx (anchor_in = where the literals of its first sequence start)
__device__ __forceinline__ void lz_seq_fields(const LzMatch* __restrict__ list, uint32_t j, uint32_t anchor_in, uint32_t& lit,
                                              uint32_t& ml, uint32_t& anchor, LzMatch& r) {
  r = list[j];
  if (j) {
    const LzMatch q = list[j - 1u];
    anchor = q.pos + q.len;
  } else {
    anchor = anchor_in;
  }
  lit = r.pos - anchor;
  ml = (uint32_t)r.len - 4u;
}
}  // namespace

// per sub-range: where the literals of its first sequence start (the end of the last match before it in the chunk) and
// the bytes its sequences take
__global__ __launch_bounds__(64) void k_lz4_sizes(uint32_t n_chunks, const uint32_t* __restrict__ sub_first,
                                                  const LzMatch* __restrict__ matches, const uint32_t* __restrict__ counts,
                                                  const uint32_t* __restrict__ last_end, uint32_t* __restrict__ anchor_in,
                                                  uint32_t* __restrict__ sub_size, uint32_t max_matches,
                                                  const uint32_t* __restrict__ sub_chunk) {
  const uint32_t lane = threadIdx.x;
  const uint32_t total = sub_first[n_chunks];
  for (uint32_t idx = blockIdx.x; idx < total; idx += gridDim.x) {
    const uint32_t c = sub_chunk[idx];
    const uint32_t first = sub_first[c];
    uint32_t a = 0u;
    for (uint32_t j = idx; j > first;) {  // (uniform; usually one step)
      --j;
      const uint32_t le = last_end[j];
      if (le) {
        a = le;
        break;
      }
    }
    const LzMatch* list = matches + (size_t)idx * max_matches;
    const uint32_t m = counts[idx];
    uint32_t acc = 0u;
    for (uint32_t j = lane; j < m; j += 64u) {
      uint32_t lit, ml, anchor;
      LzMatch r;
      lz_seq_fields(list, j, a, lit, ml, anchor, r);
      acc += 1u + lz_ext_bytes(lit) + lit + 2u + lz_ext_bytes(ml);
    }
    uint32_t sum;
    (void)lz_wave_excl_scan(acc, lane, &sum);
    if (lane == 0u) {
      anchor_in[idx] = a;
      sub_size[idx] = sum;
    }
  }
}

// per sub-range: its sequences (headers and match fields by their lanes) and every LITERAL byte that lies inside the
// sub-range, whichever sequence it belongs to -- its own, or the next one of the chunk (the bytes behind the sub-range's
// last match; that sequence may be the block's last, literals-only one). All copies are local: 8 KiB per wave at most.
struct LzEmitArgs {
  const uint8_t* stream;
  const uint64_t* chunk_dst;
  const uint32_t* chunk_payload;
  uint32_t n_chunks;
  const uint32_t* sub_first;
  const LzMatch* matches;
  const uint32_t* counts;
  const uint32_t* last_end;
  const uint32_t* anchor_in;
  const uint32_t* sub_size;
  const uint32_t* sub_chunk;  // [sub-range] -> chunk (k_lz4_match wrote it)
  const uint32_t* before;      // [sub-range]: bytes of the sequences of the chunk's earlier sub-ranges (k_lz4_layout)
  const uint32_t* next_pos;    // [sub-range]: where the chunk's next match behind the sub-range starts (none: the payload size)
  const uint32_t* block_size;  // [chunk] (k_lz4_layout)
  const uint64_t* block_dst;   // [chunk]: where [u32 size][block] goes in `out` (k_lz4_offsets)
  uint8_t* out;                // the framed streams themselves
  uint64_t out_capacity;
  uint32_t sub_bytes, max_matches;
};

// sub-range idx (of chunk c) straight from and to global memory: every lane moves the literals of its own sequence
__device__ __forceinline__ void lz_emit_direct(const LzEmitArgs& A, uint32_t idx, uint32_t c, uint32_t lane) {
  const uint8_t* __restrict__ stream = A.stream;
  const uint64_t* __restrict__ chunk_dst = A.chunk_dst;
  const uint32_t* __restrict__ chunk_payload = A.chunk_payload;
  const uint32_t* __restrict__ sub_first = A.sub_first;
  const LzMatch* __restrict__ matches = A.matches;
  const uint32_t* __restrict__ counts = A.counts;
  const uint32_t* __restrict__ last_end = A.last_end;
  const uint32_t* __restrict__ anchor_in = A.anchor_in;
  const uint32_t* __restrict__ sub_size = A.sub_size;
  const uint32_t sub_bytes = A.sub_bytes, max_matches = A.max_matches;
  {
    const uint32_t first = sub_first[c], end = sub_first[c + 1u];
    const uint32_t n = chunk_payload[c];
    const uint32_t s = (idx - first) * sub_bytes;
    const uint32_t e = n - s < sub_bytes ? n : s + sub_bytes;
    const uint8_t* in = stream + chunk_dst[c] + 4u;
    if (A.block_dst[c] + 4ull + A.block_size[c] > A.out_capacity) return;  // (k_lz4_offsets raised ST_OUT_OVERFLOW)
    uint8_t* out = A.out + A.block_dst[c] + 4u;
    // bytes of the sequences before mine / of all sequences; where the last match of the chunk ends
    uint32_t before = 0u, all = 0u, tail_anchor = 0u;
    for (uint32_t j = first + lane; j < end; j += 64u) {
      const uint32_t sz = sub_size[j];
      all += sz;
      before += j < idx ? sz : 0u;
      tail_anchor = max(tail_anchor, last_end[j]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      before += (uint32_t)__shfl_xor((int)before, d);
      all += (uint32_t)__shfl_xor((int)all, d);
      tail_anchor = max(tail_anchor, (uint32_t)__shfl_xor((int)tail_anchor, d));
    }
    const LzMatch* list = matches + (size_t)idx * max_matches;
    const uint32_t m = counts[idx];
    const uint32_t a_in = anchor_in[idx];

    // ---- my sequences
    uint32_t run = before;
    for (uint32_t j0 = 0; j0 < m; j0 += 64u) {
      const uint32_t j = j0 + lane;
      uint32_t lit = 0u, ml = 0u, anchor = 0u, size = 0u;
      LzMatch r;
      r.pos = 0u;
      r.len = 4u;
      r.off = 0u;
      if (j < m) {
        lz_seq_fields(list, j, a_in, lit, ml, anchor, r);
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
      // literals of these sequences that lie inside my sub-range (the first sequence's may begin before it: those bytes
      // are copied by the sub-ranges they lie in). Runs of up to kLaneLit bytes by their own lane (all lanes at once,
      // dwords through unaligned accesses, the last 1-3 bytes singly), longer ones by the whole wave one after the other
      constexpr uint32_t kLaneLit = 128u;
      const uint32_t skip = (j < m && anchor < s) ? min(lit, s - anchor) : 0u;  // (only sequence 0 can start before s)
      const uint32_t clit = lit - skip;
      const uint8_t* src = in + anchor + skip;
      uint8_t* dst = out + lit_at + skip;
      {
        const bool mine = j < m && clit <= kLaneLit;
        const uint32_t words = mine ? (clit >> 2) : 0u;
        uint32_t wmax = words;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d));
        for (uint32_t i = 0; i < wmax; ++i) {
          if (i < words) {
            uint32_t w;
            __builtin_memcpy(&w, src + 4u * i, 4);
            __builtin_memcpy(dst + 4u * i, &w, 4);
          }
        }
        if (mine)
          for (uint32_t b = words << 2; b < clit; ++b) dst[b] = src[b];
      }
      uint64_t long_runs = __ballot(j < m && clit > kLaneLit);
      while (long_runs) {
        const int q = __builtin_ctzll(long_runs);
        long_runs &= long_runs - 1ull;
        const uint32_t ql = (uint32_t)__builtin_amdgcn_readlane((int)clit, q);
        const uint32_t qa = (uint32_t)__builtin_amdgcn_readlane((int)(anchor + skip), q);
        const uint32_t qo = (uint32_t)__builtin_amdgcn_readlane((int)(lit_at + skip), q);
        lz_copy_wave(in + qa, out + qo, ql, lane);
      }
      run += tile_total;
    }

    // ---- the bytes behind my last match (all of my bytes if I have none): literals of the chunk's NEXT sequence
    const uint32_t t0 = m ? last_end[idx] : s;
    if (t0 < e) {
      // the next sub-range with a match (uniform search; usually the neighbour)
      uint32_t nxt = end;
      for (uint32_t j0 = idx + 1u; j0 < end && nxt == end; j0 += 64u) {
        const uint32_t j = j0 + lane;
        const uint64_t has = __ballot(j < end && counts[j] != 0u);
        if (has) nxt = j0 + (uint32_t)__builtin_ctzll(has);
      }
      uint32_t a2, dst0;
      if (nxt < end) {  // first sequence of sub-range nxt: everything between has no sequences, so it starts where mine end
        a2 = anchor_in[nxt];
        const LzMatch r0 = matches[(size_t)nxt * max_matches];
        dst0 = before + sub_size[idx] + 1u + lz_ext_bytes(r0.pos - a2);
      } else {  // the block's last sequence
        a2 = tail_anchor;
        dst0 = all + 1u + lz_ext_bytes(n - a2);
      }
      lz_copy_wave(in + t0, out + dst0 + (t0 - a2), e - t0, lane);
    }

    // ---- the last sub-range of the chunk also writes the header of the last sequence and the block's size
    if (idx + 1u == end && lane == 0u) {
      const uint32_t lit = n - tail_anchor;
      uint8_t* o = out + all;
      *o = (uint8_t)((lit < 15u ? lit : 15u) << 4);
      if (lit >= 15u) (void)lz_put_ext(o + 1u, lit);
    }
  }
}

#ifdef CLDN_DEV  // (the round-3 emit kernel: an A/B reference of the development build, CLDN_HIP_LZ4_EMIT_DIRECT=1)
__global__ __launch_bounds__(64) void k_lz4_emit(const LzEmitArgs A) {
  const uint32_t total = A.sub_first[A.n_chunks];
  for (uint32_t idx = blockIdx.x; idx < total; idx += gridDim.x) lz_emit_direct(A, idx, A.sub_chunk[idx], threadIdx.x);
}
#endif

// k_lz4_emit_lds (round 5): the same bytes as lz_emit_direct, moved through LDS. The direct kernel's lanes read and write
// their own literal runs with scattered 4-byte accesses (64 different lines per instruction: the texture addresser is what
// it waits for); here the sub-range's input is staged with whole-line loads, the sequences are assembled in an LDS image
// of the output span the wave owns, and the image leaves in 16-byte units.
// What a sub-range writes (offsets in the chunk's block):
//   A  token + literal-length bytes of its first sequence              -> straight to global (1-3 bytes)
//   B  [before + |A| + skip, before + sub_size): its sequences without the literals that lie in earlier sub-ranges
//   C  the literals behind its last match (they belong to the chunk's NEXT sequence, whose header -- the gap between
//      B and C -- is written by that sequence's owner)
// B, the gap and C are one contiguous span: the LDS image covers it; a span beyond the image (literal-length fields of
// kilobytes) goes through lz_emit_direct.
template <uint32_t SUB>
__global__ __launch_bounds__(64) void k_lz4_emit_lds(const LzEmitArgs A) {
  constexpr uint32_t kImg = SUB + 320u;  // bytes of the output image (a span is SUB - matches + their extension bytes + the gap)
  __shared__ __attribute__((aligned(16))) uint32_t in32[SUB / 4u + 8u];
  __shared__ __attribute__((aligned(16))) uint32_t img32[kImg / 4u];
  const uint32_t lane = threadIdx.x;
  const uint32_t total = A.sub_first[A.n_chunks];
  for (uint32_t idx = blockIdx.x; idx < total; idx += gridDim.x) {
    const uint32_t c = A.sub_chunk[idx];
    const uint32_t first = A.sub_first[c], end = A.sub_first[c + 1u];
    const uint32_t n = A.chunk_payload[c];
    const uint32_t s = (idx - first) * SUB;
    const uint32_t e = n - s < SUB ? n : s + SUB;
    const uint8_t* in = A.stream + A.chunk_dst[c] + 4u;
    if (A.block_dst[c] + 4ull + A.block_size[c] > A.out_capacity) continue;  // (k_lz4_offsets raised ST_OUT_OVERFLOW)
    uint8_t* out = A.out + A.block_dst[c] + 4u;                               // the block goes straight into the framed stream
    // ---- the sub-range's bytes: requested first, stored to LDS once the bookkeeping below is through
    constexpr uint32_t kRounds = SUB / 16u / 64u;
    const uint32_t bytes = e - s;
    const uint32_t full = bytes >> 4;
    uint4 w[kRounds];
#pragma unroll
    for (uint32_t r = 0; r < kRounds; ++r) {
      const uint32_t i = r * 64u + lane;
      w[r] = make_uint4(0u, 0u, 0u, 0u);
      if (i < full) __builtin_memcpy(&w[r], in + s + 16u * i, 16);
    }
    const LzMatch* list = A.matches + (size_t)idx * A.max_matches;
    const uint32_t m = A.counts[idx];
    const uint32_t a_in = A.anchor_in[idx];
    const uint32_t my_size = A.sub_size[idx];
    const uint32_t before = A.before[idx];
    const uint32_t t0 = m ? A.last_end[idx] : s;
    // region C: where the bytes behind my last match go. They are literals of the chunk's next sequence (the block's last
    // one if no match follows): its anchor is my last match's end -- or my own anchor if I have no match --, its header
    // follows my sequences
    const uint32_t a2 = m ? t0 : a_in;
    const uint32_t c_at = before + my_size + 1u + lz_ext_bytes(A.next_pos[idx] - a2) + (t0 - a2);
    // the span [W0, W1) and the gap [G0, G1) inside it that is not mine
    uint32_t W0, W1, G0 = 0u, G1 = 0u;
    if (m) {
      const LzMatch r0 = list[0];
      const uint32_t lit0 = r0.pos - a_in;
      const uint32_t skip0 = a_in < s ? min(lit0, s - a_in) : 0u;
      W0 = before + 1u + lz_ext_bytes(lit0) + skip0;
      W1 = before + my_size;
      if (t0 < e) {
        G0 = W1;
        G1 = c_at;
        W1 = c_at + (e - t0);
      }
    } else {
      W0 = c_at;
      W1 = c_at + (e - t0);
    }
    const uint32_t pad = (uint32_t)((uintptr_t)(out + W0) & 15u);
    __syncthreads();  // (the previous sub-range's LDS contents are done with)
    if (W1 - W0 + pad > kImg) {  // (uniform)
      lz_emit_direct(A, idx, c, lane);
      continue;
    }
#pragma unroll
    for (uint32_t r = 0; r < kRounds; ++r) reinterpret_cast<uint4*>(in32)[r * 64u + lane] = w[r];
    if (lane < (bytes & 15u)) reinterpret_cast<uint8_t*>(in32)[(full << 4) + lane] = in[s + (full << 4) + lane];
    __syncthreads();
    const uint8_t* inb = reinterpret_cast<const uint8_t*>(in32);
    uint8_t* img = reinterpret_cast<uint8_t*>(img32);
    // block offset x <-> image byte x - W0 + pad; payload position q <-> inb[q - s]
    const uint32_t ib = pad - W0;  // (wraps; x + ib is the image byte)
    // nb bytes inb[sb ...] -> img[db ...], all by this lane: bytes up to a dword boundary of the image, dwords (the source
    // through v_alignbyte), the last bytes
    auto copy_lane = [&](uint32_t sb, uint32_t db, uint32_t nb, uint32_t wmax_hint) __attribute__((always_inline)) {
      (void)wmax_hint;
      uint32_t head = min((4u - (db & 3u)) & 3u, nb);
      for (uint32_t k = 0; k < 3u; ++k)
        if (k < head) img[db + k] = inb[sb + k];
      sb += head;
      db += head;
      nb -= head;
      const uint32_t words = nb >> 2;
      uint32_t wmax = words;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, d));
      uint32_t lo = in32[sb >> 2];
      for (uint32_t k = 0; k < wmax; ++k) {
        if (k < words) {
          const uint32_t hi = in32[(sb >> 2) + k + 1u];
          img32[(db >> 2) + k] = __builtin_amdgcn_alignbyte(hi, lo, sb);
          lo = hi;
        }
      }
      const uint32_t done = words << 2;
      for (uint32_t k = 0; k < 3u; ++k)
        if (done + k < nb) img[db + done + k] = inb[sb + done + k];
    };
    // the same by the whole wave (one run)
    auto copy_wave = [&](uint32_t sb, uint32_t db, uint32_t nb) __attribute__((always_inline)) {
      const uint32_t head = min((4u - (db & 3u)) & 3u, nb);
      if (lane < head) img[db + lane] = inb[sb + lane];
      sb += head;
      db += head;
      nb -= head;
      const uint32_t words = nb >> 2;
      for (uint32_t k = lane; k < words; k += 64u) {
        const uint32_t si = (sb >> 2) + k;
        img32[(db >> 2) + k] = __builtin_amdgcn_alignbyte(in32[si + 1u], in32[si], sb);
      }
      const uint32_t done = words << 2;
      if (lane < nb - done) img[db + done + lane] = inb[sb + done + lane];
    };

    // ---- my sequences
    uint32_t run = before;
    for (uint32_t j0 = 0; j0 < m; j0 += 64u) {
      const uint32_t j = j0 + lane;
      uint32_t lit = 0u, ml = 0u, anchor = 0u, size = 0u;
      LzMatch r;
      r.pos = 0u;
      r.len = 4u;
      r.off = 0u;
      if (j < m) {
        lz_seq_fields(list, j, a_in, lit, ml, anchor, r);
        size = 1u + lz_ext_bytes(lit) + lit + 2u + lz_ext_bytes(ml);
      }
      uint32_t tile_total;
      const uint32_t at = run + lz_wave_excl_scan(size, lane, &tile_total);
      const uint32_t lit_at = at + 1u + lz_ext_bytes(lit);
      if (j < m) {
        const uint8_t token = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (ml < 15u ? ml : 15u));
        if (j == 0u) {  // region A
          uint8_t* o = out + at;
          *o++ = token;
          if (lit >= 15u) (void)lz_put_ext(o, lit);
        } else {
          uint8_t* o = img + (at + ib);
          *o++ = token;
          if (lit >= 15u) (void)lz_put_ext(o, lit);
        }
        uint8_t* o = img + (lit_at + lit + ib);
        *o++ = (uint8_t)(r.off & 0xffu);
        *o++ = (uint8_t)(r.off >> 8);
        if (ml >= 15u) (void)lz_put_ext(o, ml);
      }
      constexpr uint32_t kLaneLit = 128u;
      const uint32_t skip = (j < m && anchor < s) ? min(lit, s - anchor) : 0u;  // (only sequence 0 can start before s)
      const uint32_t clit = j < m ? lit - skip : 0u;
      const uint32_t sb = j < m ? anchor + skip - s : 0u;
      const uint32_t db = j < m ? lit_at + skip + ib : 0u;
      copy_lane(sb, db, clit <= kLaneLit ? clit : 0u, 0u);
      uint64_t long_runs = __ballot(clit > kLaneLit);
      while (long_runs) {
        const int q = __builtin_ctzll(long_runs);
        long_runs &= long_runs - 1ull;
        copy_wave((uint32_t)__builtin_amdgcn_readlane((int)sb, q), (uint32_t)__builtin_amdgcn_readlane((int)db, q),
                  (uint32_t)__builtin_amdgcn_readlane((int)clit, q));
      }
      run += tile_total;
    }
    // ---- region C
    if (t0 < e) copy_wave(t0 - s, c_at + ib, e - t0);
    __syncthreads();
    // ---- the image leaves: whole 16-byte units where all 16 bytes are mine, byte by byte at the span's ends and the gap
    {
      const uint32_t img_end = pad + (W1 - W0);
      const uint32_t g0 = G1 > G0 ? G0 + ib : 0u, g1 = G1 > G0 ? G1 + ib : 0u;  // the gap in image bytes
      uint8_t* gbase = out + W0 - pad;                                          // (16-byte aligned)
      const uint32_t units = (img_end + 15u) >> 4;
      for (uint32_t u = lane; u < units; u += 64u) {
        const uint32_t b0 = u << 4, b1 = b0 + 16u;
        const bool whole = b0 >= pad && b1 <= img_end && (g1 <= b0 || g0 >= b1);
        if (whole) {
          *reinterpret_cast<uint4*>(gbase + b0) = reinterpret_cast<const uint4*>(img32)[u];
        } else {
          for (uint32_t k = b0; k < b1; ++k)
            if (k >= pad && k < img_end && !(k >= g0 && k < g1)) gbase[k] = img[k];
        }
      }
    }
    // ---- the last sub-range of the chunk also writes the header of the last sequence
    if (idx + 1u == end && lane == 0u) {
      const uint32_t lit = n - a2;
      uint8_t* o = out + before + my_size;
      *o = (uint8_t)((lit < 15u ? lit : 15u) << 4);
      if (lit >= 15u) (void)lz_put_ext(o + 1u, lit);
    }
  }
}

// one wave per chunk: before[] (exclusive sum of the sub-ranges' sequence bytes), next_pos[] (the first match behind each
// sub-range) and the size of the chunk's block
__global__ __launch_bounds__(64) void k_lz4_layout(const uint32_t* __restrict__ chunk_payload, const uint32_t* __restrict__ sub_first,
                                                  const LzMatch* __restrict__ matches, const uint32_t* __restrict__ counts,
                                                  const uint32_t* __restrict__ last_end, const uint32_t* __restrict__ sub_size,
                                                  uint32_t* __restrict__ before, uint32_t* __restrict__ next_pos,
                                                  uint32_t* __restrict__ block_size, uint32_t max_matches) {
  const uint32_t c = blockIdx.x, lane = threadIdx.x;
  const uint32_t first = sub_first[c], end = sub_first[c + 1u];
  const uint32_t n = chunk_payload[c];
  uint32_t run = 0u, tail = 0u;
  for (uint32_t j0 = first; j0 < end; j0 += 64u) {
    const uint32_t j = j0 + lane;
    const uint32_t sz = j < end ? sub_size[j] : 0u;
    uint32_t tot;
    const uint32_t excl = lz_wave_excl_scan(sz, lane, &tot);
    if (j < end) before[j] = run + excl;
    run += tot;
    tail = max(tail, j < end ? last_end[j] : 0u);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) tail = max(tail, (uint32_t)__shfl_xor((int)tail, d));
  // backwards: the first match of the nearest later sub-range that has one
  uint32_t carry = n;
  for (uint32_t t = (end - first + 63u) / 64u; t-- > 0u;) {
    const uint32_t j = first + t * 64u + lane;
    uint32_t v = (j < end && counts[j] != 0u) ? matches[(size_t)j * max_matches].pos : 0xffffffffu;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {  // inclusive suffix minimum
      const uint32_t o = (uint32_t)__shfl_down((int)v, d);
      if (lane + (uint32_t)d < 64u) v = min(v, o);
    }
    const uint32_t behind = (uint32_t)__shfl_down((int)v, 1);
    if (j < end) next_pos[j] = min(lane < 63u ? behind : 0xffffffffu, carry);
    carry = min(carry, (uint32_t)__shfl((int)v, 0));
  }
  if (lane == 0u) {
    const uint32_t lit = n - tail;  // the block's last sequence: literals only
    block_size[c] = first == end ? 1u : run + 1u + lz_ext_bytes(lit) + lit;
  }
}

// one workgroup: where every chunk's [u32 size][block] goes (the chunks of the batch back to back, as k_finish lays the
// stage-1 streams out), the size headers, the stream offsets of the clouds; a chunk that does not fit raises
// ST_OUT_OVERFLOW and is not written
__global__ __launch_bounds__(1024) void k_lz4_offsets(const uint32_t* __restrict__ block_size, const uint32_t* __restrict__ sub_first,
                                                     uint32_t n_chunks, const uint32_t* __restrict__ cloud_first_chunk, uint32_t n_clouds,
                                                     uint64_t* __restrict__ block_dst, uint32_t* __restrict__ chunk_sizes,
                                                     uint64_t* __restrict__ stream_offsets, uint8_t* __restrict__ out,
                                                     uint64_t out_capacity, uint32_t* __restrict__ status) {
  __shared__ unsigned long long wtot[16];
  __shared__ unsigned long long carry;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  if (tid == 0u) carry = 0ull;
  __syncthreads();
  for (uint32_t base = 0; base < n_chunks; base += 1024u) {
    const uint32_t c = base + tid;
    const uint32_t bs = c < n_chunks ? block_size[c] : 0u;
    const unsigned long long mine = c < n_chunks ? 4ull + bs : 0ull;
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long o = (unsigned long long)__shfl_up((long long)incl, d);
      if (lane >= (uint32_t)d) incl += o;
    }
    if (lane == 63u) wtot[wave] = incl;
    __syncthreads();
    unsigned long long dst = carry + incl - mine;
    for (uint32_t w = 0; w < wave; ++w) dst += wtot[w];
    if (c < n_chunks) {
      block_dst[c] = dst;
      chunk_sizes[c] = bs;
      if (dst + mine > out_capacity) {
        atomicOr(status, (uint32_t)ST_OUT_OVERFLOW);
      } else {
        uint8_t* o = out + dst;
        o[0] = (uint8_t)bs;
        o[1] = (uint8_t)(bs >> 8);
        o[2] = (uint8_t)(bs >> 16);
        o[3] = (uint8_t)(bs >> 24);
        if (sub_first[c] == sub_first[c + 1u]) o[4] = 0u;  // a chunk without a byte of payload: the single token 0x00
      }
    }
    __syncthreads();
    if (tid == 1023u) carry = dst + mine;
    __syncthreads();
  }
  __threadfence_block();
  __syncthreads();
  // stream offset of a cloud = where its first chunk goes; a cloud without chunks: where the next one's does
  const unsigned long long total = carry;
  for (uint32_t k = tid; k <= n_clouds; k += 1024u) {
    const uint32_t fc = k < n_clouds ? cloud_first_chunk[k] : n_chunks;
    stream_offsets[k] = fc < n_chunks ? block_dst[fc] : total;
  }
}

int lz4_launch(const Lz4Launch& L) {
  if (L.n_chunks == 0u) return CLDN_HIP_OK;
  hipError_t e;
  const uint32_t sub_bytes = L.fast ? kLzFastSubBytes : kLzSubBytes;
  const uint32_t max_matches = L.fast ? kLzFastMaxMatches : kLzMaxMatches;
  hipLaunchKernelGGL(k_lz4_plan, dim3(1), dim3(1024), 0, L.stream, L.chunk_payload, L.n_chunks, L.sub_first, sub_bytes,
                     (uint32_t)std::min<uint64_t>(L.max_subs, 0xffffffffu), L.status);
  if ((e = hipGetLastError()) != hipSuccess) return launch_fail(e, "k_lz4_plan");
  // one wave per sub-range, at most `max_subs` of them: workgroups beyond the real number find nothing to do
  // (round 5: one workgroup per sub-range -- the dispatcher balances them; 8192 looping workgroups ran 1.68 rounds)
  const uint32_t grid = (uint32_t)std::min<uint64_t>(L.max_subs, 1u << 20);
  if (L.fast)
    hipLaunchKernelGGL((k_lz4_match<kLzFastSubBytes, kLzFastHashBits, kLzFastMaxMatches>), dim3(grid), dim3(64), 0, L.stream, L.stage1,
                       L.chunk_dst, L.chunk_payload, L.n_chunks, L.sub_first, L.matches, L.counts, L.last_end, L.sub_chunk);
  else
    hipLaunchKernelGGL((k_lz4_match<kLzSubBytes, kLzHashBits, kLzMaxMatches>), dim3(grid), dim3(64), 0, L.stream, L.stage1, L.chunk_dst,
                       L.chunk_payload, L.n_chunks, L.sub_first, L.matches, L.counts, L.last_end, L.sub_chunk);
  if ((e = hipGetLastError()) != hipSuccess) return launch_fail(e, "k_lz4_match");
  hipLaunchKernelGGL(k_lz4_sizes, dim3(grid), dim3(64), 0, L.stream, L.n_chunks, L.sub_first, L.matches, L.counts, L.last_end,
                     L.anchor_in, L.sub_size, max_matches, L.sub_chunk);
  if ((e = hipGetLastError()) != hipSuccess) return launch_fail(e, "k_lz4_sizes");
  hipLaunchKernelGGL(k_lz4_layout, dim3(L.n_chunks), dim3(64), 0, L.stream, L.chunk_payload, L.sub_first, L.matches, L.counts, L.last_end,
                     L.sub_size, L.before, L.next_pos, L.block_size, max_matches);
  if ((e = hipGetLastError()) != hipSuccess) return launch_fail(e, "k_lz4_layout");
  hipLaunchKernelGGL(k_lz4_offsets, dim3(1), dim3(1024), 0, L.stream, L.block_size, L.sub_first, L.n_chunks, L.cloud_first_chunk, L.n_clouds,
                     L.block_dst, L.block_sizes_out, L.stream_offsets, L.out, L.out_capacity, L.status);
  if ((e = hipGetLastError()) != hipSuccess) return launch_fail(e, "k_lz4_offsets");
  LzEmitArgs A;
  A.stream = L.stage1;
  A.chunk_dst = L.chunk_dst;
  A.chunk_payload = L.chunk_payload;
  A.n_chunks = L.n_chunks;
  A.sub_first = L.sub_first;
  A.matches = L.matches;
  A.counts = L.counts;
  A.last_end = L.last_end;
  A.anchor_in = L.anchor_in;
  A.sub_size = L.sub_size;
  A.sub_chunk = L.sub_chunk;
  A.before = L.before;
  A.next_pos = L.next_pos;
  A.block_size = L.block_size;
  A.block_dst = L.block_dst;
  A.out = L.out;
  A.out_capacity = L.out_capacity;
  A.sub_bytes = sub_bytes;
  A.max_matches = max_matches;
#ifdef CLDN_DEV
  static const bool direct = dev_env("CLDN_HIP_LZ4_EMIT_DIRECT") != nullptr;  // (the kernel of rounds 3-4, for A/B runs)
  if (direct) hipLaunchKernelGGL(k_lz4_emit, dim3(grid), dim3(64), 0, L.stream, A);
  else
#endif
  if (L.fast) hipLaunchKernelGGL(k_lz4_emit_lds<kLzFastSubBytes>, dim3(grid), dim3(64), 0, L.stream, A);
  else hipLaunchKernelGGL(k_lz4_emit_lds<kLzSubBytes>, dim3(grid), dim3(64), 0, L.stream, A);
  if ((e = hipGetLastError()) != hipSuccess) return launch_fail(e, "k_lz4_emit");
  return CLDN_HIP_OK;
}

}  // namespace cldn
