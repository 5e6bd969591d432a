// stage1_prims.h -- wave / workgroup primitives and LDS helpers shared by the encode TU (stage1_kernels.hip) and the
// decode TU (stage1_decode.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cldn {

// ------------------------------------------------------------------------------------------------------------
// wave / block primitives (wave64)
// ------------------------------------------------------------------------------------------------------------

// inclusive prefix sum across the 64 lanes of a wave using DPP row shifts + row broadcasts
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x) {
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return x;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_scan(x), 63);
}

// Block-wide exclusive prefix sum of one uint32 per thread. `wtot` is an LDS array of >= 32 uint32. Contains
// one __syncthreads(); the caller must separate two consecutive calls (or other uses of wtot) by a barrier.
template <int T>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t x, uint32_t* wtot, uint32_t* total) {
  constexpr int NW = T / 64;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t incl = wave_inclusive_scan(x);
  if (lane == 63u) wtot[wave] = incl;
  __syncthreads();
  const uint32_t wt = (lane < (uint32_t)NW) ? wtot[lane] : 0u;
  const uint32_t wincl = wave_inclusive_scan(wt);
  const uint32_t base = (wave == 0u) ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)wincl, (int)wave - 1);
  *total = (uint32_t)__builtin_amdgcn_readlane((int)wincl, NW - 1);
  return base + incl - x;
}

// ------------------------------------------------------------------------------------------------------------
// LDS helpers
// ------------------------------------------------------------------------------------------------------------

// 4 bytes at an arbitrary byte offset of an LDS dword array (reads 2 dwords; buffers carry 8 bytes of slack)
__device__ __forceinline__ uint32_t lds_u32(const uint32_t* base, uint32_t byte_off) {
  const uint32_t i = byte_off >> 2;
  const uint32_t lo = base[i];
  const uint32_t hi = base[i + 1];
  return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> ((byte_off & 3u) * 8u));
}
__device__ __forceinline__ uint64_t lds_u64(const uint32_t* base, uint32_t byte_off) {
  const uint32_t i = byte_off >> 2;
  const uint32_t a = base[i], b = base[i + 1], c = base[i + 2];
  const uint32_t sh = (byte_off & 3u) * 8u;
  const uint32_t lo = (uint32_t)(((((uint64_t)b) << 32) | a) >> sh);
  const uint32_t hi = (uint32_t)(((((uint64_t)c) << 32) | b) >> sh);
  return (((uint64_t)hi) << 32) | lo;
}
__device__ __forceinline__ uint64_t lds_raw(const uint32_t* base, uint32_t byte_off, uint32_t nbytes) {
  if (nbytes == 8u) return lds_u64(base, byte_off);
  const uint32_t v = lds_u32(base, byte_off);
  return nbytes == 4u ? v : (nbytes == 2u ? (v & 0xffffu) : (v & 0xffu));
}


template <int LANES>
struct alignas(4) FloatVec {
  float v[LANES];
};

}  // namespace cldn
