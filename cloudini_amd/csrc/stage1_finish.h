// stage1_finish.h -- k_finish: the last kernel of an encode call (included by stage1_kernels.hip).
//
// The regular kernels leave every chunk's stream as segments in the chunk's worst-case sized slot. What remains is
//   (1) the V5 sections of the chunk (src/v5_codec.cpp:423-491),
//   (2) the chunk's payload size, the position of the chunk in the framed stream = sum of (4 + payload) over all
//       chunks before it (src/chunk_writer.cpp:27-48), the per-cloud stream offsets,
//   (3) the byte-exact concatenation [u32 size][segments ...] at that position.
// Rounds 1-2 ran these as separate launches (section kernels, k_chunk_offsets: one workgroup scanning all chunks,
// k_compact). k_finish does (2) and (3), and for the common schema -- ONE adaptive field of 2 or 4 bytes that committed
// Palette -- also (1), in one launch:
//
//   * one workgroup per chunk (plus `splits - 1` helpers per chunk for small batches). Workgroups take a ticket, so a
//     workgroup's predecessors in ticket order have all started; every workgroup PUBLISHES before it WAITS, so the
//     protocol cannot deadlock whatever the dispatch order is;
//   * the chunk's framed size is published as one 8-byte record {epoch, size} with an agent-scope (write-through)
//     store; a chunk's position is the sum of the records of all chunks before it, read by the workgroup's threads in
//     parallel (agent-scope loads, bounded spin) -- ONE inter-workgroup hop, no chain: no size depends on another
//     chunk's position. The epoch (a per-call counter) makes stale records of earlier calls harmless, so the record
//     array is never cleared. Batches of more than 1024 chunks add one level: the chunk that closes a block of 1024
//     also publishes the stream position behind it (anchor), later blocks start from the anchor before them;
//   * fused Palette: the size of the section (3 + U * bpv + ceil(bits * n / 8), v5_codec.cpp:298-306) is known as soon
//     as the hash table holds the chunk's distinct values -- halfway through the section's work. The workgroup
//     publishes then, and finishes the section (ranks, palette values, bit-packed indexes) straight into the final
//     stream, no slot round trip. Even chunks finish their section first and copy their regular segments second, odd
//     chunks the other way round, so that the LDS-latency-bound half of one workgroup overlaps the HBM-bound half of
//     its neighbours on the same CU.
#pragma once

namespace cldn {

constexpr uint32_t kFinMaxSegs = 96u + 2u * kMaxAdaptive;  // sub-chunk or piece segments + two per section
constexpr uint32_t kFinItemUnits = 256u;                   // 4 KiB per work item
constexpr uint32_t kFinMaxItems = 512u;
constexpr uint32_t kFinSpinLimit = 1u << 22;               // polls (each >= ~0.2 us) before a workgroup gives up

struct FinishArgs {
  const ChunkDesc* chunks;
  uint32_t n_chunks;
  uint32_t splits;                    // workgroups per chunk
  const uint32_t* cloud_first_chunk;  // [n_clouds + 1]
  uint32_t n_clouds;
  const uint8_t* slots;
  uint64_t slot_stride;
  const Seg* segs;
  uint32_t segs_per_chunk;
  uint32_t subs;                      // regular segments per chunk; the sections follow, two segments each
  unsigned long long* rec;            // [n_chunks] {epoch << 32 | 4 + payload}
  unsigned long long* rec2;           // [n_chunks] {epoch << 32 | U} of the fused Palette section (helpers need it)
  unsigned long long* anchor;         // [n_chunks / 1024 + 1], zero at launch: 1 + end of chunk 1024 k + 1023 in the stream
  uint32_t epoch;                     // != 0, changes with every call
  uint32_t* ticket;                   // zero at launch
  uint32_t use_ticket;
  uint32_t test_timeout;              // test hook (cldn_hip_debug_finish_timeout_once): without the ticket the launch reports ST_FINISH_TIMEOUT at once
  uint32_t copy_mode;                 // fin_copy: 1 = the round-3 loop (one unit per lane, both source units loaded by the lane);
                                      // anything else = neighbour's unit by DPP, two rows in flight, non-temporal accesses (default)
  uint32_t ablate;                    // profiling only: 1 no copy, 2 no section body (wrong output)
  unsigned long long* trace;          // profiling only (CLDN_HIP_FINISH_TRACE): [n_chunks][16] wall_clock64() stamps of the leaders' phases
  uint32_t order;                     // fused Palette: 0 even chunks section first, odd chunks copy first; 1 all section first; 2 all copy first
  uint32_t* chunk_payload;            // out [n_chunks]
  uint64_t* chunk_dst;                // out [n_chunks]
  uint64_t* stream_offsets;           // out [n_clouds + 1]
  uint8_t* out;
  uint64_t out_capacity;
  uint32_t* status;
  // fused Palette section (FUSE_BPV != 0)
  const uint8_t* modes;               // [n_clouds * n_adaptive]
  uint32_t n_adaptive;
  uint32_t fuse_field;
  const uint8_t* fuse_col;            // SoA column of the field (whole batch)
  uint16_t* fuse_first;               // scratch column for the > 3072-distinct-values path
};

struct FinishLds {
  Seg seg[kFinMaxSegs];
  uint32_t doff[kFinMaxSegs];  // destination offset of every segment behind the size word
  uint32_t item_seg[kFinMaxItems], item_u0[kFinMaxItems];
  uint32_t n_items, item_units, payload, ticket, timeout, pad;
  unsigned long long base;
};

__device__ __forceinline__ unsigned long long fin_load(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fin_store(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wait for record `p` of this call; 0 when the spin budget ran out (L.timeout raised)
__device__ __forceinline__ uint32_t fin_wait(const unsigned long long* p, uint32_t epoch, FinishLds& L) {
  for (uint32_t spins = 0;; ++spins) {
    const unsigned long long x = fin_load(p);
    if ((uint32_t)(x >> 32) == epoch) return (uint32_t)x;
    if (spins >= kFinSpinLimit || ((spins & 255u) == 255u && *(volatile uint32_t*)&L.timeout != 0u)) {
      L.timeout = 1u;
      return 0u;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

// wave 0: payload size, destination offsets and the work-item table from the segment table in LDS (lane l owns segments
// l, l + 64, ...: wave scans instead of a serial walk). Segments skip0 / skip1 (the fused section's two, written by the
// workgroup itself) get a place but no items.
__device__ __forceinline__ void fin_layout(FinishLds& L, uint32_t n_segs, uint32_t skip0, uint32_t skip1,
                                           uint32_t dst_mis /* (dst0 + 4) & 15 */, uint32_t lane) {
  constexpr uint32_t K = (kFinMaxSegs + 63u) / 64u;
  uint32_t z[K], d[K], cnt[K], ibase[K];
  uint32_t payload = 0u;
#pragma unroll
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t s = lane + 64u * k;
    z[k] = s < n_segs ? L.seg[s].size : 0u;
    const uint32_t incl = wave_inclusive_scan(z[k]);
    d[k] = payload + incl - z[k];
    payload += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
  // work items of 4 KiB, larger when the chunk is so big (wide raw points: up to 32768 * 1024 bytes) that 4 KiB items
  // would not fit the table
  const uint32_t item_units = max(kFinItemUnits, (payload >> 4) / (kFinMaxItems - kFinMaxSegs - 8u) + 1u);
  uint32_t n_items = 0u;  // <= payload / 16 / item_units + n_segs < kFinMaxItems
#pragma unroll
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t s = lane + 64u * k;
    cnt[k] = 0u;
    if (z[k] != 0u && s != skip0 && s != skip1) {
      // items cover the destination-aligned 16-byte units of the segment (+ one item for a segment without any)
      const uint32_t head = min(z[k], (16u - ((dst_mis + d[k]) & 15u)) & 15u);
      const uint32_t units = (z[k] - head) >> 4;
      cnt[k] = max(1u, (units + item_units - 1u) / item_units);
    }
    const uint32_t incl = wave_inclusive_scan(cnt[k]);
    ibase[k] = n_items + incl - cnt[k];
    n_items += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  }
#pragma unroll
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t s = lane + 64u * k;
    if (s < n_segs) L.doff[s] = d[k];
    for (uint32_t j = 0, it = ibase[k]; j < cnt[k] && it < kFinMaxItems; ++j, ++it) {
      L.item_seg[it] = s;
      L.item_u0[it] = j * item_units;
    }
  }
  if (lane == 0u) {
    L.n_items = min(n_items, kFinMaxItems);
    L.item_units = item_units;
    L.payload = payload;
  }
}

// the waves of the chunk's workgroups take the items round-robin: byte-exact copy of the segments, source segments
// start 16-byte aligned, the destination position is arbitrary (16-byte destination units built with byte funnel shifts)
__device__ __forceinline__ void fin_copy(const FinishLds& L, const uint8_t* __restrict__ slot, uint8_t* __restrict__ chunk_out,
                                         uint32_t wid, uint32_t n_waves, uint32_t lane, uint32_t mode) {
  const uint32_t n_items = L.n_items, item_units = L.item_units;
  for (uint32_t it = wid; it < n_items; it += n_waves) {
    const uint32_t sidx = L.item_seg[it], u0 = L.item_u0[it];
    const Seg sg = L.seg[sidx];
    const uint8_t* src = slot + sg.off;
    uint8_t* dst = chunk_out + 4u + L.doff[sidx];
    const uint32_t size = sg.size;
    // head: bytes until dst is 16-byte aligned; tail: bytes behind the last whole unit (first item of the segment)
    const uint32_t head = min(size, (uint32_t)((16u - (uint32_t)((uintptr_t)dst & 15u)) & 15u));
    const uint32_t body_units = (size - head) >> 4;
    const uint32_t tail = (size - head) & 15u;
    if (u0 == 0u) {
      if (lane < head) dst[lane] = src[lane];
      if (lane >= 32u && lane < 32u + tail) {
        const uint32_t k = head + body_units * 16u + (lane - 32u);
        dst[k] = src[k];
      }
    }
    // body: dst-aligned 16-byte units; source bytes [head + 16j, head + 16j + 16) straddle two aligned units
    const uint32_t sdw = head >> 2, sb = head & 3u;
    const uint4* src4 = reinterpret_cast<const uint4*>(src);
    uint4* dst4 = reinterpret_cast<uint4*>(dst + head);
    const uint32_t u1 = min(body_units, u0 + item_units);
    auto build = [&](const uint4& a, const uint4& b) __attribute__((always_inline)) {
      uint32_t w0, w1, w2, w3, w4;
      switch (sdw) {
        case 0: w0 = a.x; w1 = a.y; w2 = a.z; w3 = a.w; w4 = b.x; break;
        case 1: w0 = a.y; w1 = a.z; w2 = a.w; w3 = b.x; w4 = b.y; break;
        case 2: w0 = a.z; w1 = a.w; w2 = b.x; w3 = b.y; w4 = b.z; break;
        default: w0 = a.w; w1 = b.x; w2 = b.y; w3 = b.z; w4 = b.w; break;
      }
      uint4 o;
      o.x = funnel_bytes(w0, w1, sb);
      o.y = funnel_bytes(w1, w2, sb);
      o.z = funnel_bytes(w2, w3, sb);
      o.w = funnel_bytes(w3, w4, sb);
      return o;
    };
    if (mode == 1u) {  // A/B switch: one unit per lane, both of its source units loaded by the lane itself (rounds 2-3)
      for (uint32_t j = u0 + lane; j < u1; j += 64u) {
        const uint4 a = src4[j];
        uint4 b = make_uint4(0u, 0u, 0u, 0u);
        if (head != 0u) b = src4[j + 1u];
        dst4[j] = build(a, b);
      }
    } else if (u0 < u1) {
      // the unit behind mine is my neighbour's: lane l takes lane l + 1's load by DPP (wave_shl:1), lane 63 the next
      // row's first unit. All lanes load (clamped index), the loop is wave-uniform. `last` = the last unit that may be
      // read: one behind the item's last when the units straddle (the per-lane variant reads it as well).
      const uint32_t last = head != 0u ? u1 : u1 - 1u;
      auto shl1 = [&](const uint4& a, const uint4& lane63) __attribute__((always_inline)) {
        uint4 r;
        r.x = (uint32_t)__builtin_amdgcn_update_dpp((int)lane63.x, (int)a.x, 0x130, 0xf, 0xf, false);
        r.y = (uint32_t)__builtin_amdgcn_update_dpp((int)lane63.y, (int)a.y, 0x130, 0xf, 0xf, false);
        r.z = (uint32_t)__builtin_amdgcn_update_dpp((int)lane63.z, (int)a.z, 0x130, 0xf, 0xf, false);
        r.w = (uint32_t)__builtin_amdgcn_update_dpp((int)lane63.w, (int)a.w, 0x130, 0xf, 0xf, false);
        return r;
      };
      auto lane0 = [&](const uint4& a) __attribute__((always_inline)) {
        return make_uint4((uint32_t)__builtin_amdgcn_readfirstlane((int)a.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)a.y),
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)a.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)a.w));
      };
      auto ld = [&](uint32_t j, bool nt) __attribute__((always_inline)) {
        const uint4* q = src4 + min(j, last);
        if (!nt) return *q;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q));
        return make_uint4(v.x, v.y, v.z, v.w);
      };
      auto st = [&](uint32_t j, const uint4& v, bool nt) __attribute__((always_inline)) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 q;
        q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
        if (nt) __builtin_nontemporal_store(q, reinterpret_cast<u32x4*>(dst4 + j));
        else dst4[j] = v;
      };
      auto rows_loop = [&](auto rows_tag, auto nt_tag) __attribute__((always_inline)) {
        constexpr uint32_t RW = decltype(rows_tag)::value;
        constexpr bool NT = decltype(nt_tag)::value;
        for (uint32_t jb = u0; jb < u1; jb += 64u * RW) {  // RW rows of 64 units in flight per iteration
          const uint32_t j = jb + lane;
          uint4 a[RW + 1];
#pragma unroll
          for (uint32_t r = 0; r < RW; ++r) a[r] = ld(j + 64u * r, NT);
          a[RW] = make_uint4(0u, 0u, 0u, 0u);
          if (lane == 63u && head != 0u) a[RW] = ld(j + 64u * (RW - 1u) + 1u, NT);
#pragma unroll
          for (uint32_t r = 0; r < RW; ++r) {
            const uint4 b = shl1(a[r], r + 1u < RW ? lane0(a[r + 1u]) : a[RW]);
            if (j + 64u * r < u1) st(j + 64u * r, build(a[r], b), NT);
          }
        }
      };
      rows_loop(std::integral_constant<uint32_t, 2>{}, std::true_type{});
    }
  }
}

// k_chunk_sizes: chunk-table output (no framing): payload bytes of every chunk = sum of its segments; *not_contiguous (zero
// at launch) becomes 1 if some chunk's non-empty segments do not form one run of its slot (segment k + 1 starts where
// segment k ends).
__global__ __launch_bounds__(256) void k_chunk_sizes(const Seg* __restrict__ segs, uint32_t segs_per_chunk, uint32_t n_chunks,
                                                     uint32_t* __restrict__ chunk_payload, uint32_t* __restrict__ not_contiguous) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= n_chunks) return;
  uint32_t payload = 0u, next = 0xffffffffu;
  bool contiguous = true;
  for (uint32_t s = 0; s < segs_per_chunk; ++s) {
    const Seg sg = segs[(size_t)c * segs_per_chunk + s];
    if (sg.size == 0u) continue;
    if (next != 0xffffffffu && sg.off != next) contiguous = false;
    next = sg.off + sg.size;
    payload += sg.size;
  }
  chunk_payload[c] = payload;
  if (!contiguous) atomicOr(not_contiguous, 1u);
}

// T threads; FUSE_BPV = 0 (no section work), 2 or 4 (Palette section of A.fuse_field in-kernel; T = 512, or 1024 for small
// batches: a chunk's section then has twice the threads)
template <int T, int FUSE_BPV>
__global__ __launch_bounds__(T, (T == 1024 ? 4 : 8)) __attribute__((amdgpu_num_sgpr(80))) void k_finish(const FinishArgs A) {
  using RawT = typename std::conditional<FUSE_BPV == 4, uint32_t, uint16_t>::type;
  using P = Pal32<RawT>;
  __shared__ FinishLds L;
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];  // Pal32 (FUSE_BPV != 0)
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  if (A.test_timeout != 0u && A.use_ticket == 0u) {  // (uniform)
    if (tid == 0u && blockIdx.x == 0u) atomicOr(A.status, (uint32_t)ST_FINISH_TIMEOUT);
    return;
  }
  if (tid == 0u) {
    // Order: the workgroup index. The hardware hands out a grid's workgroups in index order (per XCD), so every
    // workgroup a workgroup waits for -- all of lower index -- has started or finished. A.use_ticket replaces that
    // observation by a ticket counter (CLDN_HIP_FINISH_TICKET=1; one contended atomic per workgroup: 11 us per 1000).
    L.ticket = A.use_ticket ? atomicAdd(A.ticket, 1u) : blockIdx.x;
    L.timeout = 0u;
    L.base = 0ull;
  }
  __syncthreads();
  const uint32_t t = L.ticket;
  const uint32_t c = t % A.n_chunks;   // tickets [0, n_chunks) are the chunks' leaders: they publish
  const uint32_t y = t / A.n_chunks;
  const bool leader = (y == 0u);
  const ChunkDesc cd = A.chunks[c];
  const uint32_t n_segs = A.segs_per_chunk;
  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 0u] = wall_clock64();
  if (tid < n_segs) L.seg[tid] = A.segs[(size_t)c * n_segs + tid];

  // ---- fused Palette: build the table, which gives the section's size
  bool fuse = false, slow = false;
  uint32_t U = 0u;
  const uint32_t n = cd.n_points;
  const RawT* col = nullptr;
  uint16_t* first = nullptr;
  uint32_t s_a = 0xffffffffu, s_b = 0xffffffffu;  // segment indexes of the fused section
  if (FUSE_BPV != 0) {
    fuse = A.modes[cd.cloud * A.n_adaptive + A.fuse_field] == 1u;  // uniform
    if (fuse) {
      s_a = A.subs + 2u * A.fuse_field;
      s_b = s_a + 1u;
      col = reinterpret_cast<const RawT*>(A.fuse_col) + cd.first_point;
      first = A.fuse_first + cd.first_point;
      if (leader) {
        const P p(smem);
        if (pal32_build<RawT, T>(p, col, n, A.trace ? A.trace + (size_t)c * 16u + 8u : nullptr)) {
          U = p.misc[0];
        } else {
          slow = true;
          if (!pal32_slow_first<RawT, T>(p, col, n, first, A.status)) return;  // status raised: the call fails loudly
          // U = number of first occurrences (counted again, with the ranks, by pal32_slow_rank)
          uint32_t mine = 0u;
          for (uint32_t i = tid; i < n; i += T) mine += first[i] == (uint16_t)i ? 1u : 0u;
          (void)block_exclusive_scan<T>(mine, p.wtot, &U);
          __syncthreads();
        }
        if (tid == 0u) fin_store(A.rec2 + c, ((unsigned long long)A.epoch << 32) | U);
      } else {
        if (tid == 0u) L.pad = fin_wait(A.rec2 + c, A.epoch, L);
        __syncthreads();
        if (L.timeout) {
          if (tid == 0u) atomicOr(A.status, (uint32_t)ST_FINISH_TIMEOUT);
          return;
        }
        U = L.pad;
      }
    }
  }
  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 1u] = wall_clock64();
  __syncthreads();  // L.seg is complete
  if (fuse && tid == 0u) {
    L.seg[s_a].size = 3u + U * (uint32_t)sizeof(RawT);
    L.seg[s_b].size = (palette_bits(U) * n + 7u) >> 3;
  }
  __syncthreads();

  // ---- sizes: publish mine, add up everybody's before me
  if (tid == 0u) {
    uint32_t payload = 0u;
    for (uint32_t s = 0; s < n_segs; ++s) payload += L.seg[s].size;
    L.payload = payload;
    if (leader) fin_store(A.rec + c, ((unsigned long long)A.epoch << 32) | (payload + 4u));
  }
  {
    unsigned long long part = 0ull;
    const uint32_t blk = c >> 10;
    {  // at most 1024 / T records per thread: all loads in flight before the first check
      constexpr uint32_t PER = 1024u / T;
      unsigned long long x[PER];
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t j = (blk << 10) + tid + k * T;
        x[k] = j < c ? fin_load(A.rec + j) : ((unsigned long long)A.epoch << 32);
      }
      // the ranks of the fused section need nothing from outside: they are computed while the records arrive
      if (FUSE_BPV != 0 && fuse && leader && !slow && !(A.ablate & 2u)) (void)pal32_rank<RawT, T>(P(smem), nullptr);
#pragma unroll
      for (uint32_t k = 0; k < PER; ++k) {
        const uint32_t j = (blk << 10) + tid + k * T;
        part += (uint32_t)(x[k] >> 32) == A.epoch ? (uint32_t)x[k] : fin_wait(A.rec + j, A.epoch, L);
      }
    }
    if (blk != 0u && tid == T - 1) {  // everything before my block of 1024 chunks
      for (uint32_t spins = 0;; ++spins) {
        const unsigned long long x = fin_load(A.anchor + (blk - 1u));
        if (x != 0ull) {
          part += x - 1ull;
          break;
        }
        if (spins >= kFinSpinLimit || ((spins & 255u) == 255u && *(volatile uint32_t*)&L.timeout != 0u)) {
          L.timeout = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    // wave sum of the 64-bit partials (framed sizes stay below 2^26, a wave's sum below 2^32 only for ordinary points:
    // keep 64 bits), then one LDS atomic per wave
    uint32_t lo = (uint32_t)part, hi = (uint32_t)(part >> 32);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const uint32_t olo = (uint32_t)__shfl_down((int)lo, d), ohi = (uint32_t)__shfl_down((int)hi, d);
      const unsigned long long sum = ((((unsigned long long)hi) << 32) | lo) + ((((unsigned long long)ohi) << 32) | olo);
      lo = (uint32_t)sum;
      hi = (uint32_t)(sum >> 32);
    }
    if (lane == 0u && (lo | hi)) atomicAdd(&L.base, (((unsigned long long)hi) << 32) | lo);
  }
  __syncthreads();
  if (L.timeout) {
    if (tid == 0u) atomicOr(A.status, (uint32_t)ST_FINISH_TIMEOUT);
    return;
  }
  const unsigned long long dst0 = L.base;
  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 2u] = wall_clock64();
  if (wave == 0u) fin_layout(L, n_segs, s_a, s_b, (uint32_t)(((uintptr_t)A.out + dst0 + 4ull) & 15ull), lane);  // (the ADDRESS decides: `out` may have any alignment)
  __syncthreads();
  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 7u] = wall_clock64();
  const uint32_t payload = L.payload;
  if (dst0 + 4ull + payload > A.out_capacity) {
    if (tid == 0u && leader) atomicOr(A.status, (uint32_t)ST_OUT_OVERFLOW);
    return;
  }
  uint8_t* chunk_out = A.out + dst0;
  if (leader) {
    if (tid < 4u) chunk_out[tid] = (uint8_t)(payload >> (8u * tid));
    if (tid == 0u) {
      A.chunk_payload[c] = payload;
      A.chunk_dst[c] = dst0;
      if ((c & 1023u) == 1023u) fin_store(A.anchor + (c >> 10), dst0 + 4ull + payload + 1ull);
    }
    // stream offset of cloud k = destination of its first chunk; clouds without chunks inherit the next cloud's
    if (tid == 64u) {
      for (uint32_t k = cd.cloud + 1u; k-- > 0u && A.cloud_first_chunk[k] == c;) A.stream_offsets[k] = dst0;
      if (c + 1u == A.n_chunks)
        for (uint32_t k = A.n_clouds + 1u; k-- > 0u && A.cloud_first_chunk[k] == A.n_chunks;) A.stream_offsets[k] = dst0 + 4ull + payload;
    }
  }

  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 3u] = wall_clock64();
  // ---- placement
  const uint8_t* slot = A.slots + (size_t)c * A.slot_stride;
  // a chunk with three helpers or more (small batches): the leader builds the section, the helpers copy -- the section is the
  // longer half (4.5 against 2.2 us for one 1 M-point cloud), the leader's share of the copy only lengthened it
  const bool leader_copies = !(fuse && A.splits >= 3u);
  const uint32_t wid = leader_copies ? y * (T / 64) + wave : (y - 1u) * (T / 64) + wave;
  const uint32_t n_waves = (leader_copies ? A.splits : A.splits - 1u) * (T / 64);
  const bool copies = leader_copies || !leader;
  const bool section_first = !fuse || !leader || A.order == 1u || (A.order == 0u && (c & 1u) == 0u);
  if (!section_first && copies && !(A.ablate & 1u)) fin_copy(L, slot, chunk_out, wid, n_waves, lane, A.copy_mode);
  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 4u] = wall_clock64();
  if (FUSE_BPV != 0 && fuse && leader && !(A.ablate & 2u)) {
    const P p(smem);
    uint8_t* sec = chunk_out + 4u + L.doff[s_a];
    uint8_t* idx = chunk_out + 4u + L.doff[s_b];
    if (!slow) {
      pal32_values<RawT, T>(p, sec + 3u);
      pal32_pack_chunk<RawT, T>(p, col, n, palette_bits(U), idx);
    } else {
      (void)pal32_slow_rank<RawT, T>(p, col, n, first, sec + 3u);
      pal32_slow_pack<RawT, T>(p, n, palette_bits(U), first, idx);
    }
    if (tid == 0u) {
      sec[0] = 1u;
      sec[1] = (uint8_t)(U & 0xffu);
      sec[2] = (uint8_t)((U >> 8) & 0xffu);  // static_cast<uint16_t>(palette.size()), v5_codec.cpp:464
    }
  }
  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 5u] = wall_clock64();
  if (section_first && copies && !(A.ablate & 1u)) fin_copy(L, slot, chunk_out, wid, n_waves, lane, A.copy_mode);
  if (A.trace != nullptr && tid == 0u && leader) A.trace[(size_t)c * 16u + 6u] = wall_clock64();
}

}  // namespace cldn
