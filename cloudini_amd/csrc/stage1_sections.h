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
// values of a chunk the Palette build inserts one per thread and round, in index order, before the blocked pass (same-box A/B of
// k_finish on C2, 4096 / 2048 / 1024: 117.8 / 109.4 / 110.5 us -- the seed's compare-and-swaps contend for the few slots a
// palette-coded field has; tools/dev/variants.sh stage1_kernels CLDN_PAL_SEED ...)
#ifndef CLDN_PAL_SEED
#define CLDN_PAL_SEED 2048u
#endif

namespace cldn {

// the adaptive fields one section-kernel launch covers: blockIdx.y indexes the list
struct SectionFields {
  uint32_t n;
  uint8_t a[kMaxAdaptive];
};

constexpr int kS2Threads = 1024;
constexpr uint32_t kS2PalSlots = 4096;
constexpr uint32_t kS2PalCapacity = 3072;  // load factor 0.75
constexpr uint32_t kInf = 0xffffffffu;
constexpr uint32_t kProbeSlots = 8192;                       // mode probe: distinct-value count of <= 4096 values
constexpr uint32_t kProbeLds = kProbeSlots * 8u + 16u + 256u;  // keys, scan scratch (+ counters at word 40 of the scratch)

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
      if (p.misc[0] > kS2PalCapacity) return;  // over capacity: the caller gives the chunk up; do not fill the table
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

// ---------------------------------------------------------------------------------------------------------
// Mode probe (analyzeAdaptiveIntField + selectBestAdaptiveIntMode, src/v5_codec.cpp:258-316, :381-412) over the
// first W <= 4096 values of a cloud: thread t owns values [4t, 4t+4), every size is thread-local work plus one
// block-wide scan / reduction.
// ---------------------------------------------------------------------------------------------------------

template <int T>
__device__ __forceinline__ uint32_t block_sum(uint32_t x, uint32_t* wtot) {
  uint32_t total;
  (void)block_exclusive_scan<T>(x, wtot, &total);
  __syncthreads();
  return total;
}

// min over all threads with a larger thread index of their `mine` (kInf if none)
template <int T>
__device__ __forceinline__ uint32_t block_suffix_min_exclusive(uint32_t mine, uint32_t* wmin) {
  constexpr int NW = T / 64;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t s = mine;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const uint32_t o = (uint32_t)__shfl_down((int)s, dlt);
    if (lane + (uint32_t)dlt < 64u) s = min(s, o);
  }
  uint32_t excl = (uint32_t)__shfl_down((int)s, 1);
  if (lane == 63u) excl = kInf;
  if (lane == 0u) wmin[wave] = s;
  __syncthreads();
  uint32_t cross = kInf;
  for (uint32_t w = wave + 1u; w < (uint32_t)NW; ++w) cross = min(cross, wmin[w]);
  __syncthreads();
  return min(excl, cross);
}

// T threads, VPT = 4096 / T consecutive values per thread. `at(i)` = value i of the window (a column, or the AoS input).
// LDS: `smem` = distinct-value structure (16-bit keys: presence bitmap of 8 KiB; wider keys: hash table of `slots` keys),
// `wtot` = 64 words of scan scratch behind it.
template <typename RawT, int T, typename At>
__device__ __forceinline__ uint8_t probe_mode_t(At at, uint32_t n, uint32_t type, uint8_t* smem, uint32_t slots, uint32_t* wtot) {
  constexpr uint32_t VPT = 4096u / (uint32_t)T;
  const uint32_t t0 = threadIdx.x * VPT;
  const uint32_t cnt = t0 < n ? min(VPT, n - t0) : 0u;
  RawT v[VPT];
#pragma unroll
  for (uint32_t j = 0; j < VPT; ++j) v[j] = (j < cnt) ? at(t0 + j) : (RawT)0;
  const RawT pm1 = (cnt && t0 >= 1u) ? at(t0 - 1u) : (RawT)0;
  const RawT pm2 = (cnt && t0 >= 2u) ? at(t0 - 2u) : (RawT)0;
  auto as64 = [&](RawT r) { return int_field_as_i64((uint64_t)r, type); };
  // first differences (DeltaVarint / DeltaRle keys), values[-1] = 0: recomputed where needed (registers)
  auto diff_of = [&](uint32_t j) -> uint64_t {
    const int64_t cur = as64(v[j]);
    const int64_t prev = j ? as64(v[j - 1u]) : (t0 >= 1u ? as64(pm1) : 0);
    return (uint64_t)cur - (uint64_t)prev;
  };
  const uint64_t diff_before = (uint64_t)(t0 >= 1u ? as64(pm1) : 0) - (uint64_t)(t0 >= 2u ? as64(pm2) : 0);

  // DeltaVarint: 1 + sum of token lengths
  uint32_t dv = 0u;
#pragma unroll
  for (uint32_t j = 0; j < VPT; ++j)
    if (j < cnt) dv += varint64_len((int64_t)diff_of(j));
  const uint32_t delta_size = 1u + block_sum<T>(dv, wtot);

  // run heads of both run codings
  uint32_t hr = 0u, hd = 0u;
  {
    uint64_t kr = (uint64_t)pm1, kd = diff_before;
#pragma unroll
    for (uint32_t j = 0; j < VPT; ++j) {
      if (j < cnt) {
        const bool first = (t0 + j) == 0u;
        const uint64_t d = diff_of(j);
        if (first || (uint64_t)v[j] != kr) hr |= 1u << j;
        if (first || d != kd) hd |= 1u << j;
        kr = (uint64_t)v[j];
        kd = d;
      }
    }
  }
  auto run_bytes = [&](uint32_t heads, bool delta) -> uint32_t {
    const uint32_t my_first = heads ? (t0 + (uint32_t)__builtin_ctz(heads)) : kInf;
    uint32_t next_after = block_suffix_min_exclusive<T>(my_first, wtot);
    if (next_after == kInf) next_after = n;
    uint32_t bytes = 0u;
#pragma unroll
    for (uint32_t j = 0; j < VPT; ++j) {
      if (heads & (1u << j)) {
        const uint32_t above = heads & ~((2u << j) - 1u);
        const uint32_t nxt = above ? (t0 + (uint32_t)__builtin_ctz(above)) : next_after;
        bytes += (delta ? varint64_len((int64_t)diff_of(j)) : (uint32_t)sizeof(RawT)) + uvarint32_len(nxt - (t0 + j));
      }
    }
    return 5u + block_sum<T>(bytes, wtot);
  };
  const uint32_t rle_size = run_bytes(hr, false);
  const uint32_t drle_size = run_bytes(hd, true);

  // Palette: number of distinct values, exact for any window
  __syncthreads();
  uint32_t U;
  if (sizeof(RawT) == 2) {  // presence bitmap of the 65536 possible keys
    uint32_t* bm = reinterpret_cast<uint32_t*>(smem);
    for (uint32_t w = threadIdx.x; w < 2048u; w += T) bm[w] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < VPT; ++j)
      if (j < cnt) atomicOr(&bm[(uint32_t)v[j] >> 5], 1u << ((uint32_t)v[j] & 31u));
    __syncthreads();
    uint32_t pc = 0u;
    for (uint32_t w = threadIdx.x; w < 2048u; w += T) pc += (uint32_t)__builtin_popcount(bm[w]);
    U = block_sum<T>(pc, wtot);
  } else {
    // hash table of `slots` keys for at most 4096 values: the load stays below 0.7, so probe sequences stay short and the
    // table cannot fill up (`slots` need not be a power of two: multiply-shift range reduction, probing modulo `slots`)
    using Key = typename std::conditional<sizeof(RawT) == 8, unsigned long long, uint32_t>::type;
    constexpr Key kFreeKey = ~(Key)0;
    Key* keys = reinterpret_cast<Key*>(smem);
    uint32_t* cnt_l = wtot + 40;  // [0] distinct keys, [1] the all-ones value seen
    for (uint32_t sidx = threadIdx.x; sidx < slots; sidx += T) keys[sidx] = kFreeKey;
    if (threadIdx.x < 2u) cnt_l[threadIdx.x] = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < VPT; ++j) {
      if (j < cnt) {
        const Key key = (Key)v[j];
        if (key == kFreeKey) {
          cnt_l[1] = 1u;
        } else {
          const uint32_t h32 = sizeof(RawT) == 8 ? hash_u64((uint64_t)key) * 0x9e3779b1u : (uint32_t)key * 0x9e3779b1u;
          uint32_t slot = (uint32_t)(((uint64_t)h32 * slots) >> 32);
          for (;;) {
            Key k = keys[slot];
            if (k == kFreeKey) {
              k = atomicCAS(&keys[slot], kFreeKey, key);
              if (k == kFreeKey) {
                atomicAdd(&cnt_l[0], 1u);
                break;
              }
            }
            if (k == key) break;
            slot = slot + 1u == slots ? 0u : slot + 1u;
          }
        }
      }
    }
    __syncthreads();
    U = cnt_l[0] + cnt_l[1];
    __syncthreads();
  }
  const uint32_t pal_size = 3u + U * (uint32_t)sizeof(RawT) + ((palette_bits(U) * n + 7u) >> 3);

  uint32_t mode = 0u, best = delta_size;  // strict '<' in this order
  if (pal_size < best) { best = pal_size; mode = 1u; }
  if (rle_size < best) { best = rle_size; mode = 2u; }
  if (drle_size < best) { mode = 3u; }
  return (uint8_t)mode;
}

template <typename RawT>
__device__ __forceinline__ uint8_t probe_mode(const RawT* col, uint32_t n, uint32_t type, uint8_t* smem) {
  // 8192 key slots of 8 bytes (k_probe_fast's LDS), scan scratch behind them
  uint32_t* wtot = reinterpret_cast<uint32_t*>(smem + kProbeSlots * 8u + 16u);
  return probe_mode_t<RawT, kS2Threads>([&](uint32_t i) { return col[i]; }, n, type, smem, kProbeSlots, wtot);
}

// grid = (n_clouds, n_adaptive). Exact for every window (the general probe kernel is only the A/B reference).
__global__ __launch_bounds__(kS2Threads) void k_probe_fast(const DevPlan plan, const ChunkDesc* __restrict__ chunks,
                                                           const uint32_t* __restrict__ cloud_first_chunk,
                                                           const ColumnPtrs cols, uint8_t* __restrict__ modes) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t cloud = blockIdx.x, a = blockIdx.y;
  uint8_t* mode_out = modes + cloud * plan.n_adaptive + a;
  const uint32_t fc = cloud_first_chunk[cloud];
  if (fc == cloud_first_chunk[cloud + 1u]) {  // empty cloud
    if (threadIdx.x == 0) *mode_out = 0u;
    return;
  }
  const ChunkDesc cd = chunks[fc];
  const uint32_t n = cd.n_points > kProbePoints ? kProbePoints : cd.n_points;
  const uint32_t bpv = plan.adaptive[a].bpv, type = plan.adaptive[a].type;
  const uint8_t* col = cols.p[a] + (size_t)cd.first_point * bpv;
  uint8_t mode;
  if (bpv == 2u) mode = probe_mode<uint16_t>(reinterpret_cast<const uint16_t*>(col), n, type, smem);
  else if (bpv == 4u) mode = probe_mode<uint32_t>(reinterpret_cast<const uint32_t*>(col), n, type, smem);
  else mode = probe_mode<uint64_t>(reinterpret_cast<const uint64_t*>(col), n, type, smem);
  if (threadIdx.x == 0) *mode_out = mode;
}

// grid = n_chunks (one launch per adaptive field). Chunks whose mode is not Palette exit at once; chunks whose
// table overflows exit without setting handled_flags[c * n_adaptive + a] and are encoded by the general kernel.
template <typename RawT>
__global__ __launch_bounds__(kS2Threads) void k_section_palette(const DevPlan plan, const SectionFields fl,
                                                                const ChunkDesc* __restrict__ chunks,
                                                                const ColumnPtrs cols, const uint8_t* __restrict__ modes,
                                                                uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                                uint64_t reg_stride, Seg* __restrict__ segs,
                                                                uint32_t segs_per_chunk, uint32_t subs,
                                                                uint8_t* __restrict__ handled_flags, uint32_t /*append*/) {
  constexpr int T = kS2Threads;
  constexpr int NW = T / 64;
  constexpr uint32_t ROUND = T * 8u;  // values per round
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t c = blockIdx.x;
  const uint32_t a = fl.a[blockIdx.y];  // one launch covers every field of this kernel's type (grid.y)
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

// 8 consecutive values of a 2- or 4-byte column as raw dwords, loaded WITHOUT a branch (any alignment: gfx950 performs
// unaligned dwordx4 loads natively). The column buffers end with 64 spare bytes, so a group that starts inside a chunk
// may run past the chunk's end: the callers ignore those elements. Branch-free loads are what lets the compiler keep
// several groups in flight (exact s_waitcnt counting, see k_encode_fused).
template <typename RawT>
struct Grp8 {
  uint32_t dw[2 * sizeof(RawT)];
  __device__ __forceinline__ uint32_t get(int j) const {  // j known at compile time
    return sizeof(RawT) == 2 ? ((dw[j >> 1] >> ((j & 1) * 16)) & 0xffffu) : dw[j];
  }
  __device__ __forceinline__ uint32_t get_dyn(uint32_t j) const {  // run-time j: select chain, no indexed registers
    const uint32_t di = sizeof(RawT) == 2 ? (j >> 1) : j;
    uint32_t x = 0u;
#pragma unroll
    for (uint32_t k = 0; k < 2 * sizeof(RawT); ++k) {
      uint32_t a = dw[k];
      asm volatile("" : "+v"(a));  // (keeps the chain a chain: folded into an indexed load, the group would move to scratch)
      x = di == k ? a : x;
    }
    return sizeof(RawT) == 2 ? ((x >> ((j & 1u) * 16u)) & 0xffffu) : x;
  }
};
template <typename RawT>
__device__ __forceinline__ Grp8<RawT> grp8_load(const RawT* col, uint32_t i0) {
  Grp8<RawT> g;
  __builtin_memcpy(g.dw, col + i0, 8u * sizeof(RawT));
  return g;
}

// ---------------------------------------------------------------------------------------------------------
// k_section_palette32<RawT> (2- and 4-byte fields): same output as k_section_palette, built for few LDS round
// trips per value and no per-value state in registers. A table word holds key and first-occurrence index
// together (key << SHIFT | index, all-ones = free): claiming is a CAS, lowering the index an atomicMin on the
// same word; once the ranks are known the index is replaced by the palette rank.
//   seed    the chunk's first CLDN_PAL_SEED (2048) values in index order, one per thread and round: palette-coded fields have
//           few distinct values, so afterwards (nearly) every key sits at its true first index
//   pass 1  all values, 8 per thread and step; the home words of the 8 are read together and the CAS / probe /
//           atomicMin path runs only for new keys or lower indexes
//   ranks   bitmap over the chunk's indexes with one bit per first occurrence; rank = set bits below (one block scan)
//   pass 3  thread t packs the ranks of values [32t, 32t+32) into `bits` dwords (appendBitpackedIndexes,
//           v5_codec.cpp:209-227)
// ---------------------------------------------------------------------------------------------------------
template <typename RawT>
struct Pal32 {
  using Word = typename std::conditional<sizeof(RawT) == 2, uint32_t, unsigned long long>::type;
  static constexpr uint32_t kShift = sizeof(RawT) == 2 ? 16u : 32u;
  static constexpr Word kFree = ~(Word)0;
  static constexpr uint32_t kLds = kS2PalSlots * (uint32_t)sizeof(Word) + 2u * kS2Threads * 4u + 16u + 256u;
  Word* tab;         // [kS2PalSlots]
  uint32_t* bitmap;  // [kS2Threads]  bit i = index i is a first occurrence (32 * kS2Threads = kPointsPerChunk bits)
  uint32_t* prefix;  // [kS2Threads]  set bits in the words before
  uint32_t* misc;    // [0] inserted keys, [1] overflow flag
  uint32_t* wtot;    // [64]
  __device__ __forceinline__ explicit Pal32(uint8_t* lds) {
    uint32_t o = 0;
    tab = reinterpret_cast<Word*>(lds + o);
    o += kS2PalSlots * (uint32_t)sizeof(Word);
    bitmap = reinterpret_cast<uint32_t*>(lds + o);
    o += kS2Threads * 4u;
    prefix = reinterpret_cast<uint32_t*>(lds + o);
    o += kS2Threads * 4u;
    misc = reinterpret_cast<uint32_t*>(lds + o);
    o += 16u;
    wtot = reinterpret_cast<uint32_t*>(lds + o);
  }
  static __device__ __forceinline__ uint32_t home(uint32_t v) { return (v * 0x9e3779b1u) >> 20; }  // 12 bits
  static __device__ __forceinline__ uint32_t key_of(Word w) { return (uint32_t)(w >> kShift); }
  static __device__ __forceinline__ uint32_t low_of(Word w) { return (uint32_t)w & 0xffffu; }  // index, later rank

  // insert (v, index); `w` = the word read at home(v)
  __device__ __forceinline__ void insert(uint32_t v, uint32_t index, Word w) const {
    const Word mine = ((Word)v << kShift) | index;
    uint32_t slot = home(v);
    uint32_t probes = 0u;
    for (;;) {
      if (misc[0] > kS2PalCapacity) return;  // over capacity: the chunk is given up anyway; do not fill the table
      if (w == kFree) {
        w = atomicCAS(&tab[slot], kFree, mine);
        if (w == kFree) {
          atomicAdd(&misc[0], 1u);
          return;
        }
      }
      if (key_of(w) == v) {
        if (low_of(w) > index) atomicMin(&tab[slot], mine);
        return;
      }
      slot = (slot + 1u) & (kS2PalSlots - 1u);
      if (++probes > kS2PalSlots) {
        misc[1] = 1u;
        return;
      }
      w = tab[slot];
    }
  }
  // rank of a present key whose home word `w` holds another key
  __device__ __forceinline__ uint32_t find_rank(uint32_t v, Word w) const {
    uint32_t slot = home(v);
    while (w == kFree || key_of(w) != v) {
      slot = (slot + 1u) & (kS2PalSlots - 1u);
      w = tab[slot];
    }
    return low_of(w);
  }
};

// ranks of values [i0, i0 + cnt) (cnt <= 32) packed BITS bits each into BITS dwords at `out` (any byte alignment: the
// stores are unaligned dword / dwordx4 stores, which gfx950 performs natively)
template <typename RawT, uint32_t BITS, bool AHEAD = false>
__device__ __forceinline__ void pal32_pack(const Pal32<RawT>& p, const RawT* col, uint32_t i0, uint32_t n, uint32_t cnt,
                                           uint8_t* out) {
  using P = Pal32<RawT>;
  uint32_t w[BITS];
#pragma unroll
  for (uint32_t k = 0; k < BITS; ++k) w[k] = 0u;
  // AHEAD (the 1024-thread workgroups of small batches, 128 VGPRs): the four groups of 8 values are requested together,
  // without a branch (round 6: a conditional load per group cost one memory latency each -- 4.5 us of a one-cloud call's
  // k_finish); elements behind the chunk's end are read and ignored. The 512-thread workgroups of large batches have 64
  // VGPRs and enough chunks in flight to hide the latency: one group at a time.
  Grp8<RawT> grp[AHEAD ? 4 : 1];
  if constexpr (AHEAD) {
#pragma unroll
    for (int g = 0; g < 4; ++g) grp[g] = grp8_load<RawT>(col, i0 + 8u * g < n ? i0 + 8u * g : i0);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    uint32_t v[8];
    if constexpr (AHEAD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = grp[g].get(j);
    } else {
      RawT rv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) rv[j] = (RawT)0;
      if (i0 + 8u * g < n) load8<RawT>(col, i0 + 8u * g, n, rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (uint32_t)rv[j];
    }
    typename P::Word tw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) tw[j] = p.tab[P::home(v[j])];  // independent reads
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t e = (uint32_t)(8 * g + j);
      uint32_t x = 0u;
      if (e < cnt) {
        const bool hit = tw[j] != P::kFree && P::key_of(tw[j]) == v[j];
        x = hit ? P::low_of(tw[j]) : p.find_rank(v[j], tw[j]);
      }
      const uint32_t bit = e * BITS;
      w[bit >> 5] |= x << (bit & 31u);
      if ((bit & 31u) + BITS > 32u) w[(bit >> 5) + 1u] |= x >> (32u - (bit & 31u));
    }
  }
  // byte-exact: the chunk's last (short) group ends with the section's last byte -- in the final stream the next byte
  // belongs to the following chunk
  const uint32_t n_bytes = (cnt * BITS + 7u) >> 3;
  const uint32_t n_dw = n_bytes >> 2;
  if (n_dw == BITS && BITS % 4u == 0u) {
#pragma unroll
    for (uint32_t k = 0; k < BITS; k += 4u) {
      const uint4 q = make_uint4(w[k], w[k + 1u], w[k + 2u], w[k + 3u]);
      __builtin_memcpy(out + 4u * k, &q, 16);
    }
  } else {
#pragma unroll
    for (uint32_t k = 0; k < BITS; ++k) {
      if (k < n_dw) __builtin_memcpy(out + 4u * k, &w[k], 4);
      else if (k == n_dw) {
        for (uint32_t b = 0; b < (n_bytes & 3u); ++b) out[4u * k + b] = (uint8_t)(w[k] >> (8u * b));
      }
    }
  }
}

// ---- the steps of the palette section, shared by k_section_palette32 (section into the chunk's slot) and k_finish
// (stage1_finish.h: section straight to its final place). T threads: 1024 (one bitmap word and 32 values per thread) or
// 512 (two of each: four workgroups fit a CU, so that up to 1024 chunks are in flight at once instead of 512).

// clear + seed + pass 1: afterwards the table holds every distinct value with its first index and misc[0] = their
// number -- unless it returns false (more than kS2PalCapacity distinct values: pal32_slow_* take over)
template <typename RawT, int T>
__device__ __forceinline__ bool pal32_build(const Pal32<RawT>& p, const RawT* col, uint32_t n, unsigned long long* tr = nullptr) {
  using P = Pal32<RawT>;
  using Word = typename P::Word;
  constexpr uint32_t WPT = kS2Threads / T;
  const uint32_t tid = threadIdx.x;
  // the seed's values are requested before the table is cleared, all rounds at once (branch-free loads: one memory
  // latency instead of one per round)
  uint32_t seedv[CLDN_PAL_SEED / T];
#pragma unroll
  for (uint32_t r = 0; r < CLDN_PAL_SEED / T; ++r) seedv[r] = (uint32_t)col[min(r * T + tid, n - 1u)];
  for (uint32_t s = tid; s < kS2PalSlots; s += T) p.tab[s] = P::kFree;
#pragma unroll
  for (uint32_t w = 0; w < WPT; ++w) p.bitmap[w * T + tid] = 0u;
  if (tid < 4u) p.misc[tid] = 0u;
  __syncthreads();
  if (tr != nullptr && tid == 0u) tr[0] = wall_clock64();
  // seed
#pragma unroll
  for (uint32_t r = 0; r < CLDN_PAL_SEED / T; ++r) {
    const uint32_t i = r * T + tid;
    if (i < n) p.insert(seedv[r], i, p.tab[P::home(seedv[r])]);
  }
  __syncthreads();
  if (tr != nullptr && tid == 0u) tr[1] = wall_clock64();
  // pass 1: the groups of a batch are requested together (branch-free loads, one memory latency per batch: 8 groups of
  // 2-byte values or 4 groups of 4-byte values = 32 VGPRs); a group's values that are not settled yet -- rare after
  // the seed -- go through ONE insert site, picked off a bit mask
  constexpr uint32_t STEPS = 4096u / T;
  constexpr uint32_t GB = sizeof(RawT) == 2 ? (STEPS < 8u ? STEPS : 8u) : 4u;
#pragma unroll
  for (uint32_t b = 0; b < STEPS; b += GB) {
    Grp8<RawT> g[GB];
#pragma unroll
    for (uint32_t k = 0; k < GB; ++k) {
      const uint32_t i0 = ((b + k) * T + tid) * 8u;
      g[k] = grp8_load<RawT>(col, i0 < n ? i0 : 0u);
    }
#pragma unroll
    for (uint32_t k = 0; k < GB; ++k) {
      const uint32_t i0 = ((b + k) * T + tid) * 8u;
      if (i0 >= n || i0 + 8u <= CLDN_PAL_SEED) continue;  // (the seed settled the first CLDN_PAL_SEED values)
      Word w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = p.tab[P::home(g[k].get(j))];
      uint32_t todo = 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t idx = i0 + (uint32_t)j;
        const bool settled = w[j] != P::kFree && P::key_of(w[j]) == g[k].get(j) && P::low_of(w[j]) <= idx;
        todo |= (idx < n && !settled) ? (1u << j) : 0u;
      }
      while (todo != 0u) {
        const uint32_t j = (uint32_t)__builtin_ctz(todo);
        todo &= todo - 1u;
        const uint32_t v = g[k].get_dyn(j);
        p.insert(v, i0 + j, p.tab[P::home(v)]);
      }
    }
  }
  __syncthreads();
  return p.misc[1] == 0u && p.misc[0] <= kS2PalCapacity;  // uniform
}

// first occurrences -> bitmap, prefix; ranks into the table words; palette values to vals_out (any alignment).
// Returns the number of distinct values.
template <typename RawT, int T>
__device__ __forceinline__ uint32_t pal32_rank(const Pal32<RawT>& p, uint8_t* vals_out) {
  using P = Pal32<RawT>;
  using Word = typename P::Word;
  constexpr uint32_t WPT = kS2Threads / T;
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (uint32_t q = 0; q < kS2PalSlots / T; ++q) {
    const Word w = p.tab[q * T + tid];
    if (w != P::kFree) atomicOr(&p.bitmap[P::low_of(w) >> 5], 1u << (P::low_of(w) & 31u));
  }
  __syncthreads();
  uint32_t U;
  {
    // thread t owns bitmap words WPT * t ... (consecutive, so that the scan order is the index order)
    uint32_t pc[WPT], mine = 0u;
#pragma unroll
    for (uint32_t w = 0; w < WPT; ++w) {
      pc[w] = (uint32_t)__builtin_popcount(p.bitmap[WPT * tid + w]);
      mine += pc[w];
    }
    uint32_t run = block_exclusive_scan<T>(mine, p.wtot, &U);
#pragma unroll
    for (uint32_t w = 0; w < WPT; ++w) {
      p.prefix[WPT * tid + w] = run;
      run += pc[w];
    }
  }
  __syncthreads();
#pragma unroll
  for (uint32_t q = 0; q < kS2PalSlots / T; ++q) {
    const Word w = p.tab[q * T + tid];
    if (w != P::kFree) {
      const uint32_t f = P::low_of(w);
      const uint32_t rk = p.prefix[f >> 5] + (uint32_t)__builtin_popcount(p.bitmap[f >> 5] & ((1u << (f & 31u)) - 1u));
      p.tab[q * T + tid] = (w & ~(Word)0xffffu) | rk;
      if (vals_out != nullptr) {
        const uint32_t val = P::key_of(w);
        uint8_t* out = vals_out + (size_t)rk * sizeof(RawT);  // palette value rk
#pragma unroll
        for (uint32_t b = 0; b < sizeof(RawT); ++b) out[b] = (uint8_t)(val >> (8u * b));
      }
    }
  }
  __syncthreads();
  return U;
}

// palette values of a table whose words already hold the ranks (pal32_rank with vals_out = nullptr ran before the
// section's place was known: k_finish ranks while it waits for the sizes of the chunks in front)
template <typename RawT, int T>
__device__ __forceinline__ void pal32_values(const Pal32<RawT>& p, uint8_t* vals_out) {
  using P = Pal32<RawT>;
  using Word = typename P::Word;
  const uint32_t tid = threadIdx.x;
#pragma unroll
  for (uint32_t q = 0; q < kS2PalSlots / T; ++q) {
    const Word w = p.tab[q * T + tid];
    if (w != P::kFree) {
      const uint32_t val = P::key_of(w);
      uint8_t* out = vals_out + (size_t)P::low_of(w) * sizeof(RawT);
#pragma unroll
      for (uint32_t b = 0; b < sizeof(RawT); ++b) out[b] = (uint8_t)(val >> (8u * b));
    }
  }
}

// pass 3: thread t packs the ranks of its groups of 32 values into `bits` dwords each (appendBitpackedIndexes,
// v5_codec.cpp:209-227); idx_out = where the packed indexes begin (any alignment)
template <typename RawT, int T>
__device__ __forceinline__ void pal32_pack_chunk(const Pal32<RawT>& p, const RawT* col, uint32_t n, uint32_t bits,
                                                 uint8_t* idx_out) {
  constexpr uint32_t WPT = kS2Threads / T;
  const uint32_t tid = threadIdx.x;
  for (uint32_t g = 0; g < WPT; ++g) {
    const uint32_t grp = g * T + tid;  // group of 32 values
    const uint32_t i0 = grp * 32u;
    const uint32_t cnt = i0 < n ? min(32u, n - i0) : 0u;
    if (bits != 0u && cnt != 0u) {
      uint8_t* o = idx_out + (size_t)grp * bits * 4u;
      switch (bits) {  // block-uniform
        case 1: pal32_pack<RawT, 1, T == 1024>(p, col, i0, n, cnt, o); break;
        case 2: pal32_pack<RawT, 2, T == 1024>(p, col, i0, n, cnt, o); break;
        case 3: pal32_pack<RawT, 3, T == 1024>(p, col, i0, n, cnt, o); break;
        case 4: pal32_pack<RawT, 4, T == 1024>(p, col, i0, n, cnt, o); break;
        case 5: pal32_pack<RawT, 5, T == 1024>(p, col, i0, n, cnt, o); break;
        case 6: pal32_pack<RawT, 6, T == 1024>(p, col, i0, n, cnt, o); break;
        case 7: pal32_pack<RawT, 7, T == 1024>(p, col, i0, n, cnt, o); break;
        case 8: pal32_pack<RawT, 8, T == 1024>(p, col, i0, n, cnt, o); break;
        case 9: pal32_pack<RawT, 9, T == 1024>(p, col, i0, n, cnt, o); break;
        case 10: pal32_pack<RawT, 10, T == 1024>(p, col, i0, n, cnt, o); break;
        case 11: pal32_pack<RawT, 11, T == 1024>(p, col, i0, n, cnt, o); break;
        default: pal32_pack<RawT, 12, T == 1024>(p, col, i0, n, cnt, o); break;  // U <= kS2PalCapacity = 3072: <= 12 bits
      }
    }
  }
}

// ---- chunks with more than kS2PalCapacity distinct values (up to 32768): the values are split into P partitions, one
// table pass per partition finds the first occurrence of every value of that partition, and the first-occurrence index
// of EVERY value goes to a scratch column in global memory (`first`, n entries). Ranks then come from the bitmap of
// first occurrences alone: rank(i) = first occurrences below first[i]. Slow (P + 2 sweeps over the chunk), exact.
// 16-bit keys are partitioned by key range (32 ranges of 2048 keys can never overflow a table); 32-bit keys by a hash,
// P grows until every partition fits (kPalSlowMaxParts: beyond it the call fails loudly with ST_PALETTE_FULL).
constexpr uint32_t kPalSlowMaxParts = 4096;

template <typename RawT>
__device__ __forceinline__ uint32_t pal32_part(uint32_t v, uint32_t parts) {
  if (sizeof(RawT) == 2) return (v * parts) >> 16;
  uint32_t h = v * 0x85ebca6bu;
  h ^= h >> 15;
  h *= 0xc2b2ae35u;
  return (uint32_t)(((uint64_t)h * parts) >> 32);
}

// fills first[0..n); returns false (status raised) if the values could not be partitioned
template <typename RawT, int T>
__device__ __forceinline__ bool pal32_slow_first(const Pal32<RawT> p, const RawT* col, uint32_t n, uint16_t* first, uint32_t* status) {
  using P = Pal32<RawT>;
  const uint32_t tid = threadIdx.x;
  uint32_t parts = 32u;
  for (;;) {
    bool ok = true;
    for (uint32_t part = 0; part < parts && ok; ++part) {
      __syncthreads();
      for (uint32_t s = tid; s < kS2PalSlots; s += T) p.tab[s] = P::kFree;
      if (tid < 4u) p.misc[tid] = 0u;
      __syncthreads();
      for (uint32_t i = tid; i < n; i += T) {
        const uint32_t v = (uint32_t)col[i];
        if (pal32_part<RawT>(v, parts) == part) p.insert(v, i, p.tab[P::home(v)]);
      }
      __syncthreads();
      ok = p.misc[1] == 0u && p.misc[0] <= kS2PalCapacity;  // uniform
      if (!ok) break;
      for (uint32_t i = tid; i < n; i += T) {
        const uint32_t v = (uint32_t)col[i];
        if (pal32_part<RawT>(v, parts) == part) {
          uint32_t slot = P::home(v);
          typename P::Word w = p.tab[slot];
          while (w == P::kFree || P::key_of(w) != v) {
            slot = (slot + 1u) & (kS2PalSlots - 1u);
            w = p.tab[slot];
          }
          first[i] = (uint16_t)P::low_of(w);
        }
      }
    }
    if (ok) break;
    if (sizeof(RawT) == 2 || parts >= kPalSlowMaxParts) {  // (16-bit keys cannot get here)
      if (tid == 0) atomicOr(status, (uint32_t)ST_PALETTE_FULL);
      return false;
    }
    parts *= 4u;
  }
  __threadfence_block();
  __syncthreads();
  return true;
}

// bitmap / prefix from first[]; returns U
template <typename RawT, int T>
__device__ __forceinline__ uint32_t pal32_slow_rank(const Pal32<RawT> p, const RawT* col, uint32_t n, const uint16_t* first,
                                    uint8_t* vals_out) {
  constexpr uint32_t WPT = kS2Threads / T;
  const uint32_t tid = threadIdx.x;
  uint32_t U;
  uint32_t pc[WPT], bm[WPT], mine = 0u;
#pragma unroll
  for (uint32_t w = 0; w < WPT; ++w) {
    const uint32_t word = WPT * tid + w;
    uint32_t m = 0u;
    for (uint32_t b = 0; b < 32u; ++b) {
      const uint32_t i = word * 32u + b;
      if (i < n && first[i] == (uint16_t)i) m |= 1u << b;
    }
    bm[w] = m;
    p.bitmap[word] = m;
    pc[w] = (uint32_t)__builtin_popcount(m);
    mine += pc[w];
  }
  uint32_t run = block_exclusive_scan<T>(mine, p.wtot, &U);
#pragma unroll
  for (uint32_t w = 0; w < WPT; ++w) {
    const uint32_t word = WPT * tid + w;
    p.prefix[word] = run;
    uint32_t m = bm[w];
    uint32_t rk = run;
    while (m) {
      const uint32_t b = (uint32_t)__builtin_ctz(m);
      m &= m - 1u;
      const uint64_t val = (uint64_t)col[word * 32u + b];
      uint8_t* out = vals_out + (size_t)rk * sizeof(RawT);
#pragma unroll
      for (uint32_t k = 0; k < sizeof(RawT); ++k) out[k] = (uint8_t)(val >> (8u * k));
      ++rk;
    }
    run += pc[w];
  }
  __syncthreads();
  return U;
}

template <typename RawT, int T>
__device__ __forceinline__ void pal32_slow_pack(const Pal32<RawT> p, uint32_t n, uint32_t bits, const uint16_t* first, uint8_t* idx_out) {
  constexpr uint32_t WPT = kS2Threads / T;
  const uint32_t tid = threadIdx.x;
  if (bits == 0u) return;
  for (uint32_t g = 0; g < WPT; ++g) {
    const uint32_t grp = g * T + tid;
    const uint32_t i0 = grp * 32u;
    if (i0 >= n) continue;
    const uint32_t cnt = min(32u, n - i0);
    uint8_t* o = idx_out + (size_t)grp * bits * 4u;
    uint64_t scratch = 0u;
    uint32_t held = 0u, w = 0u;
    for (uint32_t j = 0; j < cnt; ++j) {
      const uint32_t f = first[i0 + j];
      const uint32_t rk = p.prefix[f >> 5] + (uint32_t)__builtin_popcount(p.bitmap[f >> 5] & ((1u << (f & 31u)) - 1u));
      scratch |= ((uint64_t)rk) << held;
      held += bits;
      if (held >= 32u) {
        const uint32_t d = (uint32_t)scratch;
        __builtin_memcpy(o + 4u * w, &d, 4);
        ++w;
        scratch >>= 32;
        held -= 32u;
      }
    }
    for (uint32_t b = 0; b < ((held + 7u) >> 3); ++b) o[4u * w + b] = (uint8_t)(scratch >> (8u * b));  // byte-exact end
  }
}

template <typename RawT, int T = kS2Threads>
__global__ __launch_bounds__(T, (T == 512 ? 8 : 4)) __attribute__((amdgpu_num_sgpr(80))) void k_section_palette32(
    const DevPlan plan, const SectionFields fl, const ChunkDesc* __restrict__ chunks, const ColumnPtrs cols,
    const uint8_t* __restrict__ modes, uint8_t* __restrict__ slots, uint64_t slot_stride, uint64_t reg_stride,
    Seg* __restrict__ segs, uint32_t segs_per_chunk, uint32_t subs, uint8_t* __restrict__ handled_flags, uint32_t append,
    const ColumnPtrs first_cols, uint32_t* __restrict__ status) {
  static_assert(sizeof(RawT) == 2 || sizeof(RawT) == 4, "16- or 32-bit keys");
  static_assert(kS2Threads * 32u == 32768u && (T == kS2Threads || 2 * T == kS2Threads), "bitmap words per thread: 1 or 2");
  using P = Pal32<RawT>;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t c = blockIdx.x;
  const uint32_t a = fl.a[blockIdx.y];  // one launch covers every field of this kernel's type (grid.y)
  const ChunkDesc cd = chunks[c];
  if (modes[cd.cloud * plan.n_adaptive + a] != 1u) return;
  const uint32_t n = cd.n_points;
  const RawT* col = reinterpret_cast<const RawT*>(cols.p[a]) + cd.first_point;
  const uint32_t sec_off = append ? segs[(size_t)c * segs_per_chunk].size : (uint32_t)reg_stride + a * kSectionStride;
  uint8_t* dst = slots + (size_t)c * slot_stride + sec_off;
  const P p(smem);
  const uint32_t tid = threadIdx.x;

  // the section in one piece: header, palette values, packed indexes right behind them (any alignment)
  uint32_t U;
  if (pal32_build<RawT, T>(p, col, n)) {
    U = p.misc[0];
    (void)pal32_rank<RawT, T>(p, dst + 3u);
    pal32_pack_chunk<RawT, T>(p, col, n, palette_bits(U), dst + 3u + U * (uint32_t)sizeof(RawT));
  } else {
    uint16_t* first = reinterpret_cast<uint16_t*>(first_cols.p[a]) + cd.first_point;
    if (!pal32_slow_first<RawT, T>(p, col, n, first, status)) return;
    uint32_t mine = 0u;
    for (uint32_t i = tid; i < n; i += T) mine += first[i] == (uint16_t)i ? 1u : 0u;
    (void)block_exclusive_scan<T>(mine, p.wtot, &U);
    __syncthreads();
    (void)pal32_slow_rank<RawT, T>(p, col, n, first, dst + 3u);
    pal32_slow_pack<RawT, T>(p, n, palette_bits(U), first, dst + 3u + U * (uint32_t)sizeof(RawT));
  }
  if (tid == 0) {
    dst[0] = 1u;
    dst[1] = (uint8_t)(U & 0xffu);
    dst[2] = (uint8_t)((U >> 8) & 0xffu);  // static_cast<uint16_t>(palette.size()), v5_codec.cpp:464
    Seg s;
    s.off = sec_off;
    s.size = 3u + U * (uint32_t)sizeof(RawT) + ((palette_bits(U) * n + 7u) >> 3);
    segs[(size_t)c * segs_per_chunk + subs + 2u * a] = s;
    s.size = 0u;
    segs[(size_t)c * segs_per_chunk + subs + 1u + 2u * a] = s;
    handled_flags[(size_t)c * plan.n_adaptive + a] = 1u;
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_section_delta32<RawT> (2- and 4-byte fields): DeltaVarint section (appendDeltaVarintSection,
// src/v5_codec.cpp:423-432) = mode byte 0, then zigzag(+1) varints of the int64 differences. With 16/32-bit
// values a difference has at most 33 bits -> tokens of 1..5 bytes, two dwords. The chunk is done in quarters of
// 8 values per thread: token lengths -> block scan -> tokens OR-ed into a zeroed LDS buffer -> the buffer leaves
// in 16-byte units (the partial last unit stays for the next quarter). The mode byte is the zero the buffer
// starts with.
// ---------------------------------------------------------------------------------------------------------
constexpr uint32_t kD32Ring = 65536;  // >= 8192 values * 5 bytes + 16
constexpr uint32_t kD32Lds = kD32Ring + 256u;

template <typename RawT>
__device__ __forceinline__ void section_delta32_body(const DevPlan& plan, uint32_t c, uint32_t a, const ChunkDesc& cd, const ColumnPtrs& cols,
                                                     uint8_t* __restrict__ slots, uint64_t slot_stride, uint64_t reg_stride,
                                                     Seg* __restrict__ segs, uint32_t segs_per_chunk, uint32_t subs,
                                                     uint8_t* __restrict__ handled_flags, uint32_t append, uint8_t* smem) {
  static_assert(sizeof(RawT) == 2 || sizeof(RawT) == 4, "differences of at most 33 bits");
  constexpr int T = kS2Threads;
  uint32_t* ring = reinterpret_cast<uint32_t*>(smem);
  uint32_t* wtot = reinterpret_cast<uint32_t*>(smem + kD32Ring);
  const uint32_t n = cd.n_points;
  const uint32_t type = plan.adaptive[a].type;
  const RawT* col = reinterpret_cast<const RawT*>(cols.p[a]) + cd.first_point;
  const uint32_t sec_off = append ? segs[(size_t)c * segs_per_chunk].size : (uint32_t)reg_stride + a * kSectionStride;
  uint8_t* dst = slots + (size_t)c * slot_stride + sec_off;
  const uint32_t tid = threadIdx.x;
  // every value the thread will look at is requested before the ring is cleared (branch-free loads: the four quarters
  // used to wait for two memory round trips each -- the kernel is bound by exactly that latency)
  Grp8<RawT> pre[4];
  RawT pre_prev[4];
#pragma unroll
  for (uint32_t q = 0; q < 4u; ++q) {
    const uint32_t i0 = q * (T * 8u) + threadIdx.x * 8u;
    pre[q] = grp8_load<RawT>(col, i0 < n ? i0 : 0u);
    pre_prev[q] = col[(i0 > 0u && i0 < n) ? i0 - 1u : 0u];
  }
  for (uint32_t i = tid; i < kD32Ring / 16u; i += T) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();

  uint32_t R = 1u, F = 0u;  // byte 0 = mode 0
#pragma unroll
  for (uint32_t q = 0; q < 4u; ++q) {
    const uint32_t i0 = q * (T * 8u) + tid * 8u;
    if (q * (T * 8u) >= n) break;
    RawT v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (RawT)pre[q].get(j);  // (elements behind the chunk's end get length 0 below)
    int64_t prev = (i0 > 0u && i0 < n) ? int_field_as_i64((uint64_t)pre_prev[q], type) : 0;
    uint32_t w0[8], w1[8], lens = 0u, total = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t cur = int_field_as_i64((uint64_t)v[j], type);
      const int64_t d = (int64_t)((uint64_t)cur - (uint64_t)prev);
      prev = cur;
      const uint64_t u = (((uint64_t)d << 1) ^ (uint64_t)(d >> 63)) + 1ull;  // <= 34 bits
      const uint32_t bl = (u >> 32) ? (64u - (uint32_t)__builtin_clzll(u)) : (32u - (uint32_t)__builtin_clz((uint32_t)u));
      const uint32_t l = (i0 + (uint32_t)j < n) ? groups7(bl) : 0u;
      w0[j] = spread28((uint32_t)u & 0x0fffffffu) | cont_mask(l ? l - 1u : 0u);
      w1[j] = (uint32_t)(u >> 28);
      lens |= l << (4 * j);
      total += l;
    }
    uint32_t tile_total;
    uint32_t off = R + block_exclusive_scan<T>(total, wtot, &tile_total);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t l = (lens >> (4 * j)) & 0xfu;
      if (l) ring_put5<kD32Ring, false>(ring, off, w0[j], w1[j], l, 0u);
      off += l;
    }
    const uint32_t r_end = R + tile_total;
    const bool last = (q + 1u) * (T * 8u) >= n;
    const uint32_t target = last ? ((r_end + 15u) & ~15u) : (r_end & ~15u);
    __syncthreads();
    ring_flush_n<T, kD32Ring>(ring, dst, F, target);
    F = target;
    R = r_end;
    __syncthreads();
  }
  if (tid == 0) {
    Seg sg;
    sg.off = sec_off;
    sg.size = R;
    segs[(size_t)c * segs_per_chunk + subs + 2u * a] = sg;
    sg.off = sec_off;
    sg.size = 0u;
    segs[(size_t)c * segs_per_chunk + subs + 1u + 2u * a] = sg;
    handled_flags[(size_t)c * plan.n_adaptive + a] = 1u;
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_section_runs<RawT> (2- and 4-byte fields): Rle and DeltaRle sections (appendRleSection / appendDeltaRleSection,
// src/v5_codec.cpp:471-491, :447-460): mode byte, u32 run count, then per run {raw value | varint(diff)} and the
// LEB128 run length. Thread t owns values [32t, 32t+32): run heads = key changes (the key before the thread's
// first value is read from the column), run index and byte offset from block scans, the length of a thread's last
// run from a suffix-min over the following threads' first heads. Run-coded sections are small by construction, so
// the records are written straight to the slot.
// ---------------------------------------------------------------------------------------------------------
template <typename RawT>
__device__ __forceinline__ void section_runs_body(const DevPlan& plan, uint32_t c, uint32_t a, const ChunkDesc& cd, uint32_t mode,
                                                  const ColumnPtrs& cols, uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                  uint64_t reg_stride, Seg* __restrict__ segs, uint32_t segs_per_chunk, uint32_t subs,
                                                  uint8_t* __restrict__ handled_flags, uint32_t append, uint8_t* smem) {
  static_assert(sizeof(RawT) == 2 || sizeof(RawT) == 4, "differences of at most 33 bits");
  constexpr int T = kS2Threads;
  uint32_t* wtot = reinterpret_cast<uint32_t*>(smem);  // [64]
  const bool delta = (mode == 3u);
  const uint32_t n = cd.n_points;
  const uint32_t type = plan.adaptive[a].type;
  const RawT* col = reinterpret_cast<const RawT*>(cols.p[a]) + cd.first_point;
  const uint32_t sec_off = append ? segs[(size_t)c * segs_per_chunk].size : (uint32_t)reg_stride + a * kSectionStride;
  uint8_t* dst = slots + (size_t)c * slot_stride + sec_off;
  const uint32_t tid = threadIdx.x;
  const uint32_t i0 = tid * 32u;
  const uint32_t cnt = i0 < n ? min(32u, n - i0) : 0u;
  auto as64 = [&](RawT r) { return int_field_as_i64((uint64_t)r, type); };

  // keys of my values and of the value before them
  uint64_t key[32];
  uint64_t key_before = 0u;
  {
    // the thread's 32 values and the two in front of them: six branch-free loads in flight together
    RawT v[32];
    Grp8<RawT> g8[4];
#pragma unroll
    for (uint32_t g = 0; g < 4u; ++g) g8[g] = grp8_load<RawT>(col, i0 + 8u * g < n ? i0 + 8u * g : 0u);
    const RawT pm1r = col[(cnt && i0 >= 1u) ? i0 - 1u : 0u];
    const RawT pm2r = col[(cnt && i0 >= 2u) ? i0 - 2u : 0u];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[8 * g + j] = (i0 + (uint32_t)(8 * g + j) < n) ? (RawT)g8[g].get(j) : (RawT)0;
    const RawT pm1 = (cnt && i0 >= 1u) ? pm1r : (RawT)0;
    const RawT pm2 = (cnt && i0 >= 2u) ? pm2r : (RawT)0;
    if (delta) {
      int64_t prev = i0 >= 1u ? as64(pm1) : 0;
      key_before = (uint64_t)prev - (uint64_t)(i0 >= 2u ? as64(pm2) : 0);
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const int64_t cur = as64(v[j]);
        key[j] = (uint64_t)cur - (uint64_t)prev;
        prev = cur;
      }
    } else {
      key_before = (uint64_t)pm1;
#pragma unroll
      for (int j = 0; j < 32; ++j) key[j] = (uint64_t)v[j];
    }
  }
  uint32_t heads = 0u;
  {
    uint64_t kb = key_before;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      if ((uint32_t)j < cnt && ((i0 + (uint32_t)j) == 0u || key[j] != kb)) heads |= 1u << j;
      kb = key[j];
    }
  }
  const uint32_t my_first = heads ? (i0 + (uint32_t)__builtin_ctz(heads)) : kInf;
  uint32_t next_after = block_suffix_min_exclusive<T>(my_first, wtot);  // barriers inside
  if (next_after == kInf) next_after = n;
  // bytes of my runs
  uint32_t bytes = 0u;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    if (heads & (1u << j)) {
      const uint32_t above = heads & ~((2u << j) - 1u);
      const uint32_t nxt = above ? (i0 + (uint32_t)__builtin_ctz(above)) : next_after;
      bytes += (delta ? varint64_len((int64_t)key[j]) : (uint32_t)sizeof(RawT)) + uvarint32_len(nxt - (i0 + (uint32_t)j));
    }
  }
  uint32_t total_bytes, total_runs;
  uint32_t boff = block_exclusive_scan<T>(bytes, wtot, &total_bytes);
  __syncthreads();
  (void)block_exclusive_scan<T>((uint32_t)__builtin_popcount(heads), wtot, &total_runs);
  uint8_t* out = dst + 5u + boff;
#pragma unroll
  for (int j = 0; j < 32; ++j) {
    if (heads & (1u << j)) {
      const uint32_t above = heads & ~((2u << j) - 1u);
      const uint32_t nxt = above ? (i0 + (uint32_t)__builtin_ctz(above)) : next_after;
      if (delta) {
        const Tok t = varint64_tok((int64_t)key[j]);  // <= 5 bytes here
        const uint64_t w = ((uint64_t)t.w1 << 32) | t.w0;
        for (uint32_t b = 0; b < t.len; ++b) *out++ = (uint8_t)(w >> (8u * b));
      } else {
#pragma unroll
        for (uint32_t b = 0; b < sizeof(RawT); ++b) *out++ = (uint8_t)(key[j] >> (8u * b));
      }
      const Tok r = uvarint32_tok(nxt - (i0 + (uint32_t)j));
      for (uint32_t b = 0; b < r.len; ++b) *out++ = (uint8_t)(r.w0 >> (8u * b));  // run <= 32768: 3 bytes
    }
  }
  if (tid == 0) {
    dst[0] = (uint8_t)mode;
    dst[1] = (uint8_t)total_runs;
    dst[2] = (uint8_t)(total_runs >> 8);
    dst[3] = (uint8_t)(total_runs >> 16);
    dst[4] = (uint8_t)(total_runs >> 24);
    Seg sg;
    sg.off = sec_off;
    sg.size = 5u + total_bytes;
    segs[(size_t)c * segs_per_chunk + subs + 2u * a] = sg;
    sg.size = 0u;
    segs[(size_t)c * segs_per_chunk + subs + 1u + 2u * a] = sg;
    handled_flags[(size_t)c * plan.n_adaptive + a] = 1u;
  }
}

// ---------------------------------------------------------------------------------------------------------
// k_section_fast: ONE launch for the DeltaVarint / Rle / DeltaRle sections of every 2- and 4-byte field (grid.y = field).
// Rounds 2-3 launched k_section_delta32 and k_section_runs per width -- four launches in a row for a sensor layout, in
// each of which every (chunk, field) whose mode belonged to the other kernel cost a workgroup that returned at once. The
// mode a cloud committed picks the body here, so every workgroup of the launch has work and the fields' workgroups share
// the chip (1024-thread workgroups: two per CU whatever their LDS).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kS2Threads) void k_section_fast(const DevPlan plan, const SectionFields fl,
                                                             const ChunkDesc* __restrict__ chunks, const ColumnPtrs cols,
                                                             const uint8_t* __restrict__ modes, uint8_t* __restrict__ slots,
                                                             uint64_t slot_stride, uint64_t reg_stride, Seg* __restrict__ segs,
                                                             uint32_t segs_per_chunk, uint32_t subs,
                                                             uint8_t* __restrict__ handled_flags, uint32_t append) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t c = blockIdx.x;
  const uint32_t a = fl.a[blockIdx.y];
  const ChunkDesc cd = chunks[c];
  const uint32_t mode = modes[cd.cloud * plan.n_adaptive + a];  // (uniform)
  const bool wide = plan.adaptive[a].bpv == 4u;
  if (mode == 0u) {
    if (wide) section_delta32_body<uint32_t>(plan, c, a, cd, cols, slots, slot_stride, reg_stride, segs, segs_per_chunk, subs, handled_flags, append, smem);
    else section_delta32_body<uint16_t>(plan, c, a, cd, cols, slots, slot_stride, reg_stride, segs, segs_per_chunk, subs, handled_flags, append, smem);
  } else if (mode == 2u || mode == 3u) {
    if (wide) section_runs_body<uint32_t>(plan, c, a, cd, mode, cols, slots, slot_stride, reg_stride, segs, segs_per_chunk, subs, handled_flags, append, smem);
    else section_runs_body<uint16_t>(plan, c, a, cd, mode, cols, slots, slot_stride, reg_stride, segs, segs_per_chunk, subs, handled_flags, append, smem);
  }
}

}  // namespace cldn
