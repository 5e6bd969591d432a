// stage1_sections.h -- fast paths of the V5 adaptive-int sections (included by stage1_kernels.hip).
//
// The general section kernel (k_encode_sections in stage1_kernels.hip) walks a chunk tile by tile with several
// barriers per tile. The kernels here take over the common cases with a handful of barriers per chunk; chunks
// they cannot handle (palette tables that overflow) are left to the general kernel.
//
//   k_section_palette<RawT>   Palette (src/v5_codec.cpp:462-469, :369-379, :209-227): parallel LDS hash insert
//                             (64-bit CAS on the key, atomicMin on the first-occurrence index), first-occurrence
//                             ranks from ONE block-wide scan, 32 indexes per thread bit-packed into `bits` dwords.
#pragma once

namespace cldn {

constexpr int kS2Threads = 1024;
constexpr uint32_t kS2PalSlots = 4096;
constexpr uint32_t kS2PalCapacity = 3072;  // load factor 0.75
constexpr uint32_t kInf = 0xffffffffu;

struct Pal2 {
  unsigned long long* keys;  // [kS2PalSlots]      ~0 = free
  uint32_t* first;           // [kS2PalSlots + 1]  first-occurrence index; slot kS2PalSlots = the 64-bit value ~0
  uint16_t* rank;            // [kS2PalSlots + 1]
  uint32_t* misc;            // [0] inserted keys, [1] overflow flag
  uint32_t* wtot;            // [64] scan scratch
};
constexpr uint32_t kS2PalLds = kS2PalSlots * 8u + (kS2PalSlots + 4u) * 4u + (kS2PalSlots + 8u) * 2u + 16u + 256u;

__device__ __forceinline__ Pal2 pal2_carve(uint8_t* lds) {
  Pal2 p;
  uint32_t o = 0;
  p.keys = reinterpret_cast<unsigned long long*>(lds + o);
  o += kS2PalSlots * 8u;
  p.first = reinterpret_cast<uint32_t*>(lds + o);
  o += (kS2PalSlots + 4u) * 4u;
  p.rank = reinterpret_cast<uint16_t*>(lds + o);
  o += (kS2PalSlots + 8u) * 2u;
  p.misc = reinterpret_cast<uint32_t*>(lds + o);
  o += 16u;
  p.wtot = reinterpret_cast<uint32_t*>(lds + o);
  return p;
}

template <typename RawT>
__device__ __forceinline__ void pal2_insert(const Pal2 p, RawT raw, uint32_t index) {
  const unsigned long long v = (unsigned long long)raw;
  uint32_t slot = kS2PalSlots;
  if (sizeof(RawT) < 8 || v != ~0ull) {
    slot = hash_u64(v) & (kS2PalSlots - 1u);
    uint32_t probes = 0u;
    for (;;) {
      const unsigned long long k = p.keys[slot];
      if (k == v) break;
      if (k == ~0ull) {
        const unsigned long long old = atomicCAS(&p.keys[slot], ~0ull, v);
        if (old == ~0ull) {
          atomicAdd(&p.misc[0], 1u);
          break;
        }
        if (old == v) break;
      }
      slot = (slot + 1u) & (kS2PalSlots - 1u);
      if (++probes > kS2PalSlots) {  // table full: give up, the chunk goes to the general kernel
        p.misc[1] = 1u;
        return;
      }
    }
  }
  if (p.first[slot] > index) atomicMin(&p.first[slot], index);
}

template <typename RawT>
__device__ __forceinline__ uint32_t pal2_find(const Pal2 p, RawT raw) {  // the value is known to be present
  const unsigned long long v = (unsigned long long)raw;
  if (sizeof(RawT) == 8 && v == ~0ull) return kS2PalSlots;
  uint32_t slot = hash_u64(v) & (kS2PalSlots - 1u);
  while (p.keys[slot] != v) slot = (slot + 1u) & (kS2PalSlots - 1u);
  return slot;
}

// 8 consecutive values starting at element `i0`; elements >= n read as 0
template <typename RawT>
__device__ __forceinline__ void load8(const RawT* col, uint32_t i0, uint32_t n, RawT (&v)[8]) {
  constexpr uint32_t kBytes = 8u * sizeof(RawT);
  if (i0 + 8u <= n && (((uintptr_t)(col + i0)) & 15u) == 0u) {
    uint32_t dw[kBytes / 4u];
    const uint4* q = reinterpret_cast<const uint4*>(col + i0);
#pragma unroll
    for (uint32_t k = 0; k < kBytes / 16u; ++k) {
      const uint4 x = q[k];
      dw[4 * k] = x.x;
      dw[4 * k + 1] = x.y;
      dw[4 * k + 2] = x.z;
      dw[4 * k + 3] = x.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (sizeof(RawT) == 2) v[j] = (RawT)((dw[j >> 1] >> ((j & 1) * 16)) & 0xffffu);
      else if (sizeof(RawT) == 4) v[j] = (RawT)dw[j % (kBytes / 4u)];
      else v[j] = (RawT)((((uint64_t)dw[(2 * j + 1) % (kBytes / 4u)]) << 32) | dw[(2 * j) % (kBytes / 4u)]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (i0 + (uint32_t)j < n) ? col[i0 + j] : (RawT)0;
  }
}

// grid = n_chunks (one launch per adaptive field). Chunks whose mode is not Palette exit at once; chunks whose
// table overflows exit without setting handled_flags[c * n_adaptive + a] and are encoded by the general kernel.
template <typename RawT>
__global__ __launch_bounds__(kS2Threads) void k_section_palette(const DevPlan plan, uint32_t a,
                                                                const ChunkDesc* __restrict__ chunks,
                                                                const ColumnPtrs cols, const uint8_t* __restrict__ modes,
                                                                uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                                uint64_t reg_stride, Seg* __restrict__ segs,
                                                                uint32_t segs_per_chunk, uint32_t subs,
                                                                uint8_t* __restrict__ handled_flags) {
  constexpr int T = kS2Threads;
  constexpr int NW = T / 64;
  constexpr uint32_t ROUND = T * 8u;  // values per round
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t c = blockIdx.x;
  const ChunkDesc cd = chunks[c];
  if (modes[cd.cloud * plan.n_adaptive + a] != 1u) return;
  const uint32_t n = cd.n_points;
  const RawT* col = reinterpret_cast<const RawT*>(cols.p[a]) + cd.first_point;
  const uint32_t sec_off = (uint32_t)reg_stride + a * kSectionStride;
  uint8_t* dst = slots + (size_t)c * slot_stride + sec_off;
  const Pal2 p = pal2_carve(smem);
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t rounds = (n + ROUND - 1u) / ROUND;  // <= 4

  for (uint32_t s = tid; s < kS2PalSlots; s += T) p.keys[s] = ~0ull;
  for (uint32_t s = tid; s <= kS2PalSlots; s += T) p.first[s] = kInf;
  if (tid < 4u) p.misc[tid] = 0u;
  __syncthreads();

  // pass 1: insert every value (any order; first occurrence = atomicMin of the index)
  for (uint32_t r = 0; r < rounds; ++r) {
    const uint32_t i0 = r * ROUND + tid * 8u;
    if (i0 < n) {
      RawT v[8];
      load8<RawT>(col, i0, n, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i0 + (uint32_t)j < n) pal2_insert<RawT>(p, v[j], i0 + (uint32_t)j);
      }
    }
  }
  __syncthreads();
  if (p.misc[1] != 0u || p.misc[0] > kS2PalCapacity) return;  // uniform: leave the chunk to the general kernel

  // pass 2: "I am the first occurrence" flags in index order (round, thread, j) -> ranks from one scan over the
  // (round, wave) totals
  uint32_t fmask = 0u;  // 8 flag bits per round
  uint32_t incl[4], cnt[4];
#pragma unroll
  for (uint32_t r = 0; r < 4u; ++r) {
    uint32_t m = 0u;
    const uint32_t i0 = r * ROUND + tid * 8u;
    if (r < rounds && i0 < n) {
      RawT v[8];
      load8<RawT>(col, i0, n, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i0 + (uint32_t)j < n && p.first[pal2_find<RawT>(p, v[j])] == i0 + (uint32_t)j) m |= 1u << j;
      }
    }
    fmask |= m << (8u * r);
    cnt[r] = (uint32_t)__builtin_popcount(m);
    incl[r] = wave_inclusive_scan(cnt[r]);
    if (lane == 63u) p.wtot[r * NW + wave] = incl[r];
  }
  __syncthreads();
  const uint32_t wincl = wave_inclusive_scan(p.wtot[lane]);  // 4 * 16 = 64 totals, one per lane
  const uint32_t U = (uint32_t)__builtin_amdgcn_readlane((int)wincl, 63);
#pragma unroll
  for (uint32_t r = 0; r < 4u; ++r) {
    const int f = (int)(r * NW + wave);
    const uint32_t rowbase = (f == 0) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)wincl, f - 1);
    const uint32_t m = (fmask >> (8u * r)) & 0xffu;
    if (m) {
      uint32_t rk = rowbase + incl[r] - cnt[r];
      const uint32_t i0 = r * ROUND + tid * 8u;
      RawT v[8];
      load8<RawT>(col, i0, n, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (m & (1u << j)) {
          p.rank[pal2_find<RawT>(p, v[j])] = (uint16_t)rk;
          const unsigned long long val = (unsigned long long)v[j];
          uint8_t* out = dst + 3u + (size_t)rk * sizeof(RawT);  // palette value rk behind the 3-byte header
#pragma unroll
          for (uint32_t k = 0; k < sizeof(RawT); ++k) out[k] = (uint8_t)(val >> (8u * k));
          ++rk;
        }
      }
    }
  }
  if (tid == 0) {
    dst[0] = 1u;
    dst[1] = (uint8_t)(U & 0xffu);
    dst[2] = (uint8_t)((U >> 8) & 0xffu);  // static_cast<uint16_t>(palette.size()), v5_codec.cpp:464
  }
  __syncthreads();

  // pass 3: appendBitpackedIndexes (v5_codec.cpp:209-227); thread t packs indexes [32t, 32t+32) into `bits` dwords
  const uint32_t bits = palette_bits(U);
  if (bits != 0u && tid * 32u < n) {
    uint32_t* idx_out = reinterpret_cast<uint32_t*>(dst + kPaletteIndexOffset) + (size_t)tid * bits;
    uint64_t scratch = 0u;
    uint32_t held = 0u, w = 0u;
    for (uint32_t g = 0; g < 4u; ++g) {
      const uint32_t i0 = tid * 32u + g * 8u;
      if (i0 >= n) break;
      RawT v[8];
      load8<RawT>(col, i0, n, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i0 + (uint32_t)j < n) {
          scratch |= ((uint64_t)p.rank[pal2_find<RawT>(p, v[j])]) << held;
          held += bits;
          if (held >= 32u) {
            idx_out[w++] = (uint32_t)scratch;
            scratch >>= 32;
            held -= 32u;
          }
        }
      }
    }
    if (held > 0u) idx_out[w] = (uint32_t)scratch;
  }
  if (tid == 0) {
    Seg s;
    s.off = sec_off;
    s.size = 3u + U * (uint32_t)sizeof(RawT);
    segs[(size_t)c * segs_per_chunk + subs + 2u * a] = s;
    s.off = sec_off + kPaletteIndexOffset;
    s.size = (bits * n + 7u) >> 3;
    segs[(size_t)c * segs_per_chunk + subs + 1u + 2u * a] = s;
    handled_flags[(size_t)c * plan.n_adaptive + a] = 1u;
  }
}

}  // namespace cldn
