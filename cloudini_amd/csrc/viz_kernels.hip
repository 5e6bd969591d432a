// viz_kernels.hip -- cloudini_ros::applyVizLossyPreprocessing on the GPU (reference:
// cloudini_lib/src/ros_msg_utils.cpp:249-341, voxel key packVoxelKey21 :42-49).
//
// The reference walks the points once, keeps a hash set of voxel keys and copies every point whose key is new:
// "first occurrence wins, survivors keep their order". The same result without the serial walk:
//   k_viz_insert   every finite point inserts its key into an open-addressing table in HBM (64-bit CAS on the key)
//                  and lowers the slot's first-occurrence index with atomicMin -- the primitive of the Palette
//                  section encoder, at cloud scale (round 5: behind a per-workgroup LDS table, 16-byte entries)
//   k_viz_count    a point survives iff it is the first occurrence of its slot; survivors per 1024-point block
//   k_viz_offsets  exclusive scan of the block counts (one workgroup) -> output position of every block, total
//   k_viz_gather   block-local ranks (ballot + popcount) and the copy of the surviving points, order preserved
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1_launch.h"
#include "stage1_math.h"

namespace cldn {

namespace {
constexpr uint64_t kVizFree = ~0ull;  // keys have 63 bits
constexpr int kVizBlock = 1024;

__device__ __forceinline__ float viz_load_f32(const uint8_t* p) {
  if ((((uintptr_t)p) & 3u) == 0u) return *reinterpret_cast<const float*>(p);
  return __uint_as_float((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24));
}

// packVoxelKey21(static_cast<int32_t>(std::lround(f * inv_res)), ...) -- float product, half away from zero,
// long -> int32 truncation (out-of-range products give what x86-64 gives: LONG_MIN -> 0)
__device__ __forceinline__ uint64_t viz_key(float fx, float fy, float fz, float inv_res) {
  const int32_t q[3] = {(int32_t)quant_away_i64_f32(fx, inv_res), (int32_t)quant_away_i64_f32(fy, inv_res),
                        (int32_t)quant_away_i64_f32(fz, inv_res)};
  uint64_t key = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    key |= ((uint64_t)((int64_t)q[a] + ((int64_t)1 << 20)) & ((1ull << 21) - 1ull)) << (21 * a);
  return key;
}

__device__ __forceinline__ bool viz_finite(float f) { return (__float_as_uint(f) & 0x7f800000u) != 0x7f800000u; }

// Round 5: (1) a table entry is ONE 16-byte record {key, first index, pad} -- the compare-and-swap on the key and the atomicMin
// on the index touch the same line (two random HBM lines per point before); (2) the points of a workgroup (1024 consecutive
// ones) first meet in an LDS table: only the workgroup's first occurrence of a voxel goes to the table in HBM, every other point
// of the workgroup is dropped at once (a point of the same voxel with a lower index exists). Lidar points that share a voxel
// are neighbours in memory (adjacent columns of the scan), so at coarse resolutions most duplicates never leave the CU.
constexpr int kVizInsertThreads = 1024;
constexpr uint32_t kVizLocalSlots = 2048;  // LDS: 16 KiB of keys + 8 KiB of first indexes

__global__ __launch_bounds__(kVizInsertThreads) void k_viz_insert(const uint8_t* __restrict__ points, uint64_t n, uint32_t step,
                                                                  uint32_t xyz_off, float inv_res, unsigned long long* tab,
                                                                  uint64_t cap_mask, uint32_t* __restrict__ slot_of) {
  __shared__ unsigned long long lkeys[kVizLocalSlots];
  __shared__ uint32_t lfirst[kVizLocalSlots];
  const uint32_t tid = threadIdx.x;
  for (uint32_t k = tid; k < kVizLocalSlots; k += kVizInsertThreads) {
    lkeys[k] = kVizFree;
    lfirst[k] = 0xffffffffu;
  }
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * kVizInsertThreads + tid;
  bool valid = i < n;
  uint64_t key = 0;
  if (valid) {
    const uint8_t* p = points + i * step + xyz_off;
    const float fx = viz_load_f32(p), fy = viz_load_f32(p + 4), fz = viz_load_f32(p + 8);
    valid = viz_finite(fx) && viz_finite(fy) && viz_finite(fz);
    if (valid) key = viz_key(fx, fy, fz, inv_res);
  }
  const uint64_t hash = key * 0x9E3779B97F4A7C15ull;
  uint32_t ls = 0u;
  if (valid) {
    ls = (uint32_t)(hash >> 40) & (kVizLocalSlots - 1u);
    for (;;) {  // (2048 slots for at most 1024 keys: the table cannot fill)
      unsigned long long k = lkeys[ls];
      if (k == kVizFree) k = atomicCAS(&lkeys[ls], kVizFree, (unsigned long long)key);
      if (k == kVizFree || k == key) break;
      ls = (ls + 1u) & (kVizLocalSlots - 1u);
    }
    atomicMin(&lfirst[ls], tid);
  }
  __syncthreads();
  if (i >= n) return;
  if (!valid || lfirst[ls] != tid) {
    slot_of[i] = 0xffffffffu;  // dropped: not finite, or not the workgroup's first point of its voxel
    return;
  }
  uint64_t h = hash >> 20;
  for (;;) {
    h &= cap_mask;
    unsigned long long k = tab[2u * h];
    if (k == kVizFree) k = atomicCAS(&tab[2u * h], kVizFree, (unsigned long long)key);
    if (k == kVizFree || k == key) break;
    ++h;
  }
  uint32_t* first = reinterpret_cast<uint32_t*>(&tab[2u * h + 1u]);
  if (*first > (uint32_t)i) atomicMin(first, (uint32_t)i);
  slot_of[i] = (uint32_t)h;
}

// (`first` = the table as 32-bit words: slot s keeps its first-occurrence index in word 4 s + 2)
__device__ __forceinline__ bool viz_survives(uint64_t i, uint64_t n, const uint32_t* slot_of, const uint32_t* first) {
  if (i >= n) return false;
  const uint32_t s = slot_of[i];
  return s != 0xffffffffu && first[4u * (size_t)s + 2u] == (uint32_t)i;
}

__global__ __launch_bounds__(kVizBlock) void k_viz_count(uint64_t n, const uint32_t* __restrict__ slot_of,
                                                         const uint32_t* __restrict__ first,
                                                         uint32_t* __restrict__ block_count) {
  __shared__ uint32_t wcnt[kVizBlock / 64];
  const uint64_t i = (uint64_t)blockIdx.x * kVizBlock + threadIdx.x;
  const unsigned long long b = __ballot(viz_survives(i, n, slot_of, first));
  if ((threadIdx.x & 63u) == 0u) wcnt[threadIdx.x >> 6] = (uint32_t)__popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t c = 0;
    for (int w = 0; w < kVizBlock / 64; ++w) c += wcnt[w];
    block_count[blockIdx.x] = c;
  }
}

// one workgroup: exclusive scan of n_blocks counts (in place) and the total
__global__ __launch_bounds__(1024) void k_viz_offsets(uint32_t* __restrict__ block_count, uint32_t n_blocks,
                                                      unsigned long long* __restrict__ total_out) {
  __shared__ uint32_t wsum[16];
  __shared__ uint32_t carry_s;
  if (threadIdx.x == 0) carry_s = 0u;
  __syncthreads();
  for (uint32_t base = 0; base < n_blocks; base += 1024u) {
    const uint32_t i = base + threadIdx.x;
    const uint32_t x = i < n_blocks ? block_count[i] : 0u;
    uint32_t incl = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
      if ((threadIdx.x & 63u) >= (uint32_t)d) incl += o;
    }
    if ((threadIdx.x & 63u) == 63u) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t before = carry_s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wsum[w];
    if (i < n_blocks) block_count[i] = before + incl - x;
    __syncthreads();
    if (threadIdx.x == 1023u) carry_s = before + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(kVizBlock) void k_viz_gather(const uint8_t* __restrict__ points, uint64_t n, uint32_t step,
                                                          const uint32_t* __restrict__ slot_of,
                                                          const uint32_t* __restrict__ first,
                                                          const uint32_t* __restrict__ block_off,
                                                          uint8_t* __restrict__ out) {
  __shared__ uint32_t wcnt[kVizBlock / 64];
  const uint64_t i = (uint64_t)blockIdx.x * kVizBlock + threadIdx.x;
  const bool keep = viz_survives(i, n, slot_of, first);
  const unsigned long long b = __ballot(keep);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0u) wcnt[wave] = (uint32_t)__popcll(b);
  __syncthreads();
  if (!keep) return;
  uint32_t rank = block_off[blockIdx.x] + (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
  for (uint32_t w = 0; w < wave; ++w) rank += wcnt[w];
  const uint8_t* src = points + i * step;
  uint8_t* dst = out + (size_t)rank * step;
  if ((step & 15u) == 0u && ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0u) {
    for (uint32_t k = 0; k < step; k += 16u) *reinterpret_cast<uint4*>(dst + k) = *reinterpret_cast<const uint4*>(src + k);
  } else if ((step & 3u) == 0u && ((((uintptr_t)src) | ((uintptr_t)dst)) & 3u) == 0u) {
    for (uint32_t k = 0; k < step; k += 4u) *reinterpret_cast<uint32_t*>(dst + k) = *reinterpret_cast<const uint32_t*>(src + k);
  } else {
    for (uint32_t k = 0; k < step; ++k) dst[k] = src[k];
  }
}

int viz_fail(hipError_t e, const char* what) { return launch_fail(e, what); }
}  // namespace

uint64_t viz_table_capacity(uint64_t n_points) {
  uint64_t cap = 1024;
  while (cap < 2 * n_points) cap <<= 1;
  return cap;
}

int viz_launch(const VizLaunch& L) {
  hipError_t e;
  if (L.n_points == 0) {
    if ((e = hipMemsetAsync(L.total, 0, sizeof(unsigned long long), L.stream)) != hipSuccess) return viz_fail(e, "hipMemsetAsync(viz total)");
    return 0;
  }
  const uint64_t cap = viz_table_capacity(L.n_points);
  if ((e = hipMemsetAsync(L.keys, 0xff, cap * 16u, L.stream)) != hipSuccess) return viz_fail(e, "hipMemsetAsync(viz table)");
  const uint32_t n_blocks = (uint32_t)((L.n_points + kVizBlock - 1) / kVizBlock);
  static_assert(kVizInsertThreads == kVizBlock, "one grid shape");
  const uint32_t* first_words = reinterpret_cast<const uint32_t*>(L.keys);
  hipLaunchKernelGGL(k_viz_insert, dim3(n_blocks), dim3(kVizInsertThreads), 0, L.stream, L.points, L.n_points, L.point_step,
                     L.xyz_offset, L.inv_res, L.keys, cap - 1, L.slot_of);
  if ((e = hipGetLastError()) != hipSuccess) return viz_fail(e, "k_viz_insert");
  hipLaunchKernelGGL(k_viz_count, dim3(n_blocks), dim3(kVizBlock), 0, L.stream, L.n_points, L.slot_of, first_words,
                     L.block_count);
  if ((e = hipGetLastError()) != hipSuccess) return viz_fail(e, "k_viz_count");
  hipLaunchKernelGGL(k_viz_offsets, dim3(1), dim3(1024), 0, L.stream, L.block_count, n_blocks, L.total);
  if ((e = hipGetLastError()) != hipSuccess) return viz_fail(e, "k_viz_offsets");
  hipLaunchKernelGGL(k_viz_gather, dim3(n_blocks), dim3(kVizBlock), 0, L.stream, L.points, L.n_points, L.point_step,
                     L.slot_of, first_words, L.block_count, L.out);
  if ((e = hipGetLastError()) != hipSuccess) return viz_fail(e, "k_viz_gather");
  return 0;
}

}  // namespace cldn
