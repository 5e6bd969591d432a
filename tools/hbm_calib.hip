// hbm_calib.hip -- what this box's HBM delivers to hand-written kernels: the ceiling the
// stage-1 kernels are held against, next to the 8 TB/s peak of MI355X_MICROARCH.md. Every kernel is a plain grid-stride
// loop over 16-byte units; figures are total bytes moved (read + written) per second, best and median of the repeats.
//
//   copy        1 read : 1 write   (k_finish's copy half)
//   read2write1 2 reads : 1 write  (the piece kernel's mix: 16 B of points in, ~6 B of stream + column out)
//   write       fill
//   read        sum (one store per workgroup)
// each as plain and as non-temporal (__builtin_nontemporal_load/store) variant, for 256 / 512 / 1024-thread workgroups
// and several grid sizes (workgroups per CU).
//
// build: hipcc --offload-arch=gfx950 -O3 tools/hbm_calib.hip -o cloudini_amd/lib/hbm_calib   (cloudini_amd/build.py does it)
// run:   cloudini_amd/lib/hbm_calib [GiB per buffer, default 1] [piece | points14 [points] | shapes]   (piece: only the piece kernel's own shape;
//        points14: the point decoder's store shape on the bench line's batch; shapes: the store shapes of the other configs' decode legs)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                             \
      exit(1);                                                                            \
    }                                                                                     \
  } while (0)

typedef float v4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ v4 ld(const v4* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st(v4* p, v4 v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <bool NT>
__global__ void k_copy(const v4* __restrict__ a, v4* __restrict__ o, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) st<NT>(o + i, ld<NT>(a + i));
}
template <bool NT>
__global__ void k_r2w1(const v4* __restrict__ a, const v4* __restrict__ b, v4* __restrict__ o, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) st<NT>(o + i, ld<NT>(a + i) + ld<NT>(b + i));
}
template <bool NT>
__global__ void k_write(v4* __restrict__ o, size_t n, float x) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const v4 v = {x, x, x, x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) st<NT>(o + i, v);
}
template <bool NT>
__global__ void k_read(const v4* __restrict__ a, float* __restrict__ o, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += ld<NT>(a + i);
  const float t = s.x + s.y + s.z + s.w;
  if (t == 123.456f) o[blockIdx.x] = t;  // (never true for the zero-filled buffers: keeps the loads alive)
}


// ---- the piece kernel's own shape (round 5) -----------------------------------------------
// k_encode_fused<3,4,...> as a memory pattern, with its arithmetic taken out: 256-thread workgroups, one wave per PIECE of
// 8 rows x 63 points, lane l of row r loads the 16-byte point r*63 + l - 1 (all eight loads requested before the first is
// used, non-temporal), the wave leaves OUT_B bytes per point as 16-byte units (the regular stream, ~5.46 B/pt: staged in
// LDS like the real kernel's region, then read back and stored) and one 2-byte column value per point. LDS bytes per
// workgroup are a launch parameter: they set how many workgroups a CU holds (30.4 KB: 5; 18 KB: 8). PERSIST: the grid is
// wg_per_cu x CUs workgroups that take their pieces round robin instead of one workgroup per four pieces.
template <bool PERSIST, bool NT_STORE>
__global__ __launch_bounds__(256) void k_piece_shape(const v4* __restrict__ pts, uint8_t* __restrict__ out, uint16_t* __restrict__ col,
                                                      uint32_t n_pieces, uint32_t out_bytes_per_piece, uint32_t lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint8_t* region = smem + wave * lds_per_wave;
  const uint32_t step = PERSIST ? gridDim.x * 4u : 0u;
  for (uint32_t g = blockIdx.x * 4u + wave; g < n_pieces; g += step) {
    const size_t first = (size_t)g * 504u;
    v4 rows[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const size_t idx = first + (size_t)(r * 63 + (int)lane) - (first || r || lane ? 1u : 0u);
      rows[r] = __builtin_nontemporal_load(pts + idx);
    }
    // stage: every lane leaves 16 bytes per row in the region (the real kernel ORs ~16 bytes of tokens per lane and row)
    uint32_t R = out_bytes_per_piece;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const uint32_t off = ((uint32_t)r * 64u + lane) * 16u;
      if (off + 16u <= lds_per_wave) *reinterpret_cast<v4*>(region + off) = rows[r];
      if (lane) {
        const uint16_t cv = (uint16_t)__float_as_uint(rows[r].w);
        col[first + (size_t)(r * 63 + (int)lane) - 1u] = cv;
      }
    }
    uint8_t* dst = out + (size_t)g * ((out_bytes_per_piece + 15u) & ~15u);
    for (uint32_t j = lane; j < (R >> 4); j += 64u) {
      const v4 v = *reinterpret_cast<const v4*>(region + j * 16u);
      if (NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<v4*>(dst) + j);
      else reinterpret_cast<v4*>(dst)[j] = v;
    }
    if (!PERSIST) break;
  }
}

// ---- the point decoder's store shape (round 6) -----------------------------------------------------------------------------
// 16-byte points of which a decoder in KEEP mode (the bytes no field covers are left alone, as the reference's decoder leaves
// them) writes 12 + 2: every 32-byte sector of the output is written partially. FULL: the same points as whole 16-byte stores
// (CLDN_HIP_FILL_ZERO). READ: a third of a unit is read per point alongside (the encoded stream: ~6.5 B per point).
template <bool FULL, bool READ>
__global__ void k_points14(const v4* __restrict__ in, uint8_t* __restrict__ o, size_t n_points, float x, float* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_points; i += stride) {
    if (READ && (i % 5u) < 2u) s += in[i / 5u * 2u + i % 5u];  // 2 units per 5 points: 6.4 B per point
    uint8_t* pt = o + i * 16u;
    if (FULL) {
      *reinterpret_cast<v4*>(pt) = v4{x, x, x, s.x};
    } else {
      typedef float v3 __attribute__((ext_vector_type(3)));
      *reinterpret_cast<v3*>(pt) = v3{x, x, s.x};
      *reinterpret_cast<uint16_t*>(pt + 12) = (uint16_t)i;
    }
  }
  if (s.y == 123.456f) sink[blockIdx.x] = s.y;
}

// ---- the other configs' store shapes (round 6): what a decoder that leaves uncovered bytes alone writes per point ----------------
//   KIND 0  C1 / C5: packed 12-byte XYZ points, one 12-byte store (every byte covered)
//   KIND 1  C3: 32-byte points, x y z at 0 (12 bytes) + rgba at 16 (4 bytes): half of every 32-byte sector
//   KIND 2  C3 with CLDN_HIP_FILL_ZERO: two 16-byte stores
//   KIND 3  C4: packed 18-byte points, x y z intensity (one unaligned 16-byte store) + ring (2 bytes): every byte covered
template <int KIND>
__global__ void k_points_shape(uint8_t* __restrict__ o, size_t n_points, float x) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  typedef float v3 __attribute__((ext_vector_type(3)));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_points; i += stride) {
    if (KIND == 0) {
      *reinterpret_cast<v3*>(o + i * 12u) = v3{x, x, x};
    } else if (KIND == 1) {
      *reinterpret_cast<v3*>(o + i * 32u) = v3{x, x, x};
      *reinterpret_cast<uint32_t*>(o + i * 32u + 16u) = (uint32_t)i;
    } else if (KIND == 2) {
      reinterpret_cast<v4*>(o + i * 32u)[0] = v4{x, x, x, 0.0f};
      reinterpret_cast<v4*>(o + i * 32u)[1] = v4{x, 0.0f, 0.0f, 0.0f};
    } else {
      const v4 q = {x, x, x, x};
      __builtin_memcpy(o + i * 18u, &q, 16);
      const uint16_t r = (uint16_t)i;
      __builtin_memcpy(o + i * 18u + 16u, &r, 2);
    }
  }
}

template <class F>
static void run(const char* name, double bytes, int threads, int wg_per_cu, int cus, F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int grid = wg_per_cu * cus;
  for (int w = 0; w < 3; ++w) launch(grid, threads);
  std::vector<float> ms;
  for (int r = 0; r < 9; ++r) {
    CHECK(hipEventRecord(e0));
    launch(grid, threads);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float t;
    CHECK(hipEventElapsedTime(&t, e0, e1));
    ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  printf("%-16s threads %4d  wg/CU %2d  best %.3f ms = %.2f TB/s   median %.3f ms = %.2f TB/s\n", name, threads, wg_per_cu,
         ms[0], bytes / ms[0] * 1e-9, ms[ms.size() / 2], bytes / ms[ms.size() / 2] * 1e-9);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 1.0;
  const size_t bytes = (size_t)(gib * (1ull << 30)) & ~(size_t)4095;
  const size_t n = bytes / 16;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, buffers of %.2f GiB\n", prop.name, cus, bytes / double(1ull << 30));
  v4 *a, *b, *o;
  float* s;
  CHECK(hipMalloc(&a, bytes));
  CHECK(hipMalloc(&b, bytes));
  CHECK(hipMalloc(&o, bytes));
  CHECK(hipMalloc(&s, 1 << 20));
  CHECK(hipMemset(a, 0, bytes));
  CHECK(hipMemset(b, 0, bytes));
  CHECK(hipMemset(o, 0, bytes));

  if (argc > 2 && !strcmp(argv[2], "piece")) {
    // 32 M points of 16 bytes (the bench line's batch), 5.46 B/pt of stream + a u16 column
    const uint32_t n_pieces = (uint32_t)((32000000ull + 503ull) / 504ull) & ~3u;
    const size_t n_pts = (size_t)n_pieces * 504u;
    if (n_pts * 16u > bytes) { fprintf(stderr, "buffer too small\n"); return 1; }
    const uint32_t out_pp = 2752u;  // 504 points x 5.46 B
    uint16_t* colp = reinterpret_cast<uint16_t*>(b);
    const double moved = (double)n_pts * 16.0 + (double)n_pieces * out_pp + (double)n_pts * 2.0;
    const int ldsv[] = {30400, 25600, 18432, 12288};
    for (int lds : ldsv) {
      char nm[64];
      snprintf(nm, sizeof nm, "piece lds %5d", lds);
      CHECK(hipFuncSetAttribute((const void*)k_piece_shape<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 64));
      CHECK(hipFuncSetAttribute((const void*)k_piece_shape<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 64));
      CHECK(hipFuncSetAttribute((const void*)k_piece_shape<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds + 64));
      run(nm, moved, 256, (int)(n_pieces / 4 / cus), cus, [&](int, int) {
        hipLaunchKernelGGL((k_piece_shape<false, false>), dim3(n_pieces / 4), dim3(256), lds, 0, a, (uint8_t*)o, colp, n_pieces, out_pp, (uint32_t)lds / 4u);
      });
      snprintf(nm, sizeof nm, "piece ntst %5d", lds);
      run(nm, moved, 256, (int)(n_pieces / 4 / cus), cus, [&](int, int) {
        hipLaunchKernelGGL((k_piece_shape<false, true>), dim3(n_pieces / 4), dim3(256), lds, 0, a, (uint8_t*)o, colp, n_pieces, out_pp, (uint32_t)lds / 4u);
      });
      const int pw[] = {2, 4, 5, 8};
      for (int w : pw) {
        snprintf(nm, sizeof nm, "piece pers %5d", lds);
        run(nm, moved, 256, w, cus, [&](int g, int) {
          hipLaunchKernelGGL((k_piece_shape<true, false>), dim3(g), dim3(256), lds, 0, a, (uint8_t*)o, colp, n_pieces, out_pp, (uint32_t)lds / 4u);
        });
      }
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "points14")) {
    // 32 M points of 16 bytes (the bench line's decode leg): what the store pattern alone costs
    const size_t n_pts = argc > 3 ? (size_t)atoll(argv[3]) : 32000000;
    if (n_pts * 16u > bytes) { fprintf(stderr, "buffer too small\n"); return 1; }
    printf("points %zu\n", n_pts);
    const int wv[] = {2, 4, 8};
    for (int w : wv) {
      run("12+2 of 16", 16.0 * n_pts, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points14<false, false>), dim3(g), dim3(th), 0, 0, a, (uint8_t*)o, n_pts, 1.0f, s); });
      run("16 of 16", 16.0 * n_pts, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points14<true, false>), dim3(g), dim3(th), 0, 0, a, (uint8_t*)o, n_pts, 1.0f, s); });
      run("12+2, +read", 22.4 * n_pts, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points14<false, true>), dim3(g), dim3(th), 0, 0, a, (uint8_t*)o, n_pts, 1.0f, s); });
      run("16, +read", 22.4 * n_pts, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points14<true, true>), dim3(g), dim3(th), 0, 0, a, (uint8_t*)o, n_pts, 1.0f, s); });
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "shapes")) {
    // the decode legs of the other BASELINE configs: bytes the decoder writes, per point, with no arithmetic and no input
    struct { const char* name; int kind; size_t pts; double step; double written; } sh[] = {
        {"C1 12 of 12", 0, 16777216, 12, 12}, {"C3 12+4 of 32", 1, 16384000, 32, 16}, {"C3 32 of 32", 2, 16384000, 32, 32},
        {"C4 16+2 of 18", 3, 33292288, 18, 18}, {"C5 12 of 12", 0, 10000000, 12, 12}};
    const int wv[] = {2, 4, 8};
    for (auto& q : sh) {
      if (q.pts * q.step > bytes) { fprintf(stderr, "buffer too small\n"); return 1; }
      for (int w : wv) {
        char nm[64];
        snprintf(nm, sizeof(nm), "%s", q.name);
        const size_t np = q.pts;
        // (the rate printed counts the bytes the decoder is credited with: the whole points)
        if (q.kind == 0) run(nm, q.step * np, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points_shape<0>), dim3(g), dim3(th), 0, 0, (uint8_t*)o, np, 1.0f); });
        if (q.kind == 1) run(nm, q.step * np, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points_shape<1>), dim3(g), dim3(th), 0, 0, (uint8_t*)o, np, 1.0f); });
        if (q.kind == 2) run(nm, q.step * np, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points_shape<2>), dim3(g), dim3(th), 0, 0, (uint8_t*)o, np, 1.0f); });
        if (q.kind == 3) run(nm, q.step * np, 1024, w, cus, [&](int g, int th) { hipLaunchKernelGGL((k_points_shape<3>), dim3(g), dim3(th), 0, 0, (uint8_t*)o, np, 1.0f); });
      }
    }
    return 0;
  }
  const int tvals[] = {256, 512, 1024};
  const int wvals[] = {2, 4, 8, 16, 32};
  for (int t : tvals)
    for (int w : wvals) {
      if (t * w > 2048 * 4) continue;  // more than fits a CU several times over tells nothing new
      run("copy", 2.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_copy<false>, dim3(g), dim3(th), 0, 0, a, o, n); });
      run("copy nt", 2.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_copy<true>, dim3(g), dim3(th), 0, 0, a, o, n); });
      run("read2write1", 3.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_r2w1<false>, dim3(g), dim3(th), 0, 0, a, b, o, n); });
      run("read2write1 nt", 3.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_r2w1<true>, dim3(g), dim3(th), 0, 0, a, b, o, n); });
      run("write", 1.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_write<false>, dim3(g), dim3(th), 0, 0, o, n, 1.0f); });
      run("write nt", 1.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_write<true>, dim3(g), dim3(th), 0, 0, o, n, 1.0f); });
      run("read", 1.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_read<false>, dim3(g), dim3(th), 0, 0, a, s, n); });
      run("read nt", 1.0 * bytes, t, w, cus, [&](int g, int th) { hipLaunchKernelGGL(k_read<true>, dim3(g), dim3(th), 0, 0, a, s, n); });
    }
  return 0;
}
