// stage1_decode.hip -- the decode translation unit: stage-1 decode kernels (stage1_decode.h, stage1_decode_fast.h,
// stage1_decode_wave.h) and their launchers. Split from stage1_kernels.hip so that the two halves compile side by side.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <type_traits>

#include "stage1_device.h"
#include "stage1_math.h"
#include "stage1_prims.h"

#include "stage1_decode.h"
#include "stage1_decode_fast.h"
#include "stage1_decode_wave.h"
#include "stage1_decode_stream.h"
#include "stage1_decode_sections_w.h"
#include "stage1_decode_automaton.h"
#include "stage1_decode_dv.h"

#include "cloudini_hip.h"
#include "stage1_launch.h"

namespace cldn {

namespace {
int hip_fail(hipError_t e, const char* what) { return launch_fail(e, what); }

// the plan the stream kernel decodes DeltaVarint SECTIONS with: op a = the integer field a as a stream of its own
// (to_cols: into the field's dense column, offset 0; else into the points)
DevPlan sections_plan(const DevPlan& P, bool to_cols) {
  DevPlan S = P;
  S.n_ops = 1u;
  S.n_gorilla = 0u;
  S.all_varint = 1u;
  S.varint_and_raw = 0u;
  S.max_regular_bytes = 10u;
  S.min_regular_bytes = 1u;
  for (uint32_t a = 0; a < P.n_adaptive && a < 8u; ++a) {
    DevOp op = {};
    op.kind = OP_INT;
    op.type = P.adaptive[a].type;
    op.size = P.adaptive[a].bpv;
    op.max_bytes = 10;
    op.offset = to_cols ? 0u : P.adaptive[a].offset;
    S.ops[a] = op;
  }
  return S;
}
}  // namespace

int stage1_configure_decode() {
  hipError_t e;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_varint<4, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Dv2Lds<4, false, 16>::kTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_decode_varint<4>)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_varint<8, true>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)Dv2Lds<8, true, 8>::kTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_decode_varint<8>)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_tail), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)std::max<uint32_t>(std::max<uint32_t>((uint32_t)Dv2Lds<4, false, 16>::kTotal, kSmallSecLds), (uint32_t)DecSecLds::kTotal));
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_decode_tail)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_sections),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)DecSecLds::kTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_decode_sections)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_sections_cols),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)DecSecLds::kTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_decode_sections_cols)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_stream_w<12, 1>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)SwLds<12, 1>::kTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_decode_stream_w form)");
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_decode_stream_w<12, 2>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)SwLds<12, 2>::kTotal);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(k_decode_stream_w gorilla)");
  return CLDN_HIP_OK;
}

static_assert(sizeof(DecChunk) <= kDecChunkBytes, "DecodeLaunch::chunks entries must hold a DecChunk");

// wire version 2: one unframed payload, decoded by the serial restatement of DecodeV4Stage1Chunk (one lane)
int stage1_launch_decode_unframed(const DevPlan& plan, hipStream_t stream, const uint8_t* payload, uint32_t size,
                                  uint32_t capacity_points, void* chunk_slot, uint8_t* out, uint32_t* status, const WidePlan* wide,
                                  void* wide_state) {
  DecChunk dc;
  dc.src_off = 0;
  dc.src_size = size;
  dc.n_points = capacity_points;
  dc.first_point = 0;
  dc.cloud = 0;
  dc.valid = 2u;
  hipError_t e = hipMemcpyAsync(chunk_slot, &dc, sizeof(dc), hipMemcpyHostToDevice, stream);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(DecChunk)");
  if ((e = hipStreamSynchronize(stream)) != hipSuccess) return hip_fail(e, "hipStreamSynchronize");  // `dc` lives on this frame
  if (wide) {
    hipLaunchKernelGGL(k_decode_wide, dim3(1), dim3(64), 0, stream, *wide, payload, reinterpret_cast<const DecChunk*>(chunk_slot), out, 0u,
                       status, reinterpret_cast<uint8_t*>(wide_state));
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_wide");
    return CLDN_HIP_OK;
  }
  hipLaunchKernelGGL(k_decode_general, dim3(1), dim3(64), 0, stream, plan, payload, reinterpret_cast<const DecChunk*>(chunk_slot),
                     out, 0u, 0u, (const uint32_t*)nullptr, (const uint8_t*)nullptr, status);
  if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_general");
  return CLDN_HIP_OK;
}

// DeltaVarint sections into columns: k_sections_dv_cols (round 4), or the stream kernel's section mode (CLDN_HIP_DV_COLS=0: A/B switch)
// 0 = the stream kernel's section mode, 1 = k_sections_dv_cols as a launch of its own, 2 (default) = inside k_sections_w's launch
static int dv_cols_kernel() {
  static const int mode = dev_env_int("CLDN_HIP_DV_COLS", 2);
  return mode;
}

int stage1_launch_decode(const DecodeLaunch& L) {
  hipError_t e;
  if (L.n_clouds == 0) return CLDN_HIP_OK;
  // timing (cldn_hip_codec_decode_ms): events in front of / behind the kernel that decodes the regular streams
  auto ev_before = [&]() { if (L.events) (void)hipEventRecord(L.events[1], L.stream); };
  auto ev_after = [&]() { if (L.events) (void)hipEventRecord(L.events[2], L.stream); };
  DecTablesArg T;
  memset(&T, 0, sizeof(T));
  if (L.h_stream_offsets != nullptr && L.n_clouds <= kDecInlineClouds) {
    T.n = L.n_clouds;
    for (uint32_t k = 0; k <= L.n_clouds; ++k) {
      T.so[k] = L.h_stream_offsets[k];
      T.fp[k] = L.h_cloud_first_point[k];
      T.fc[k] = L.h_cloud_first_chunk[k];
    }
  }
  if (L.chunk_sizes) {
    hipLaunchKernelGGL(k_build_chunks, dim3(L.n_clouds), dim3(256), 0, L.stream, L.streams, L.stream_offsets, L.cloud_first_point,
                       L.cloud_first_chunk, L.chunk_sizes, reinterpret_cast<DecChunk*>(L.chunks), L.status, T);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_build_chunks");
  } else {
    hipLaunchKernelGGL(k_walk_chunks, dim3((L.n_clouds + 63u) / 64u), dim3(64), 0, L.stream, L.streams, L.stream_offsets,
                       L.cloud_first_point, L.cloud_first_chunk, L.n_clouds, reinterpret_cast<DecChunk*>(L.chunks),
                       L.status, T);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_walk_chunks");
  }
  if (L.n_chunks && L.wide) {  // schemas beyond the launch-argument plan: the serial decoder with the plan in device memory
    hipLaunchKernelGGL(k_decode_wide, dim3(L.n_chunks), dim3(64), 0, L.stream, *L.wide, L.streams, reinterpret_cast<const DecChunk*>(L.chunks),
                       L.out, L.uses_v5, L.status, reinterpret_cast<uint8_t*>(L.wide_state));
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_wide");
    return CLDN_HIP_OK;
  }
  if (L.n_chunks) {
    // regular streams made of varint tokens only go through the parallel kernel; the general kernel then decodes
    // the V5 sections (and whole chunks the fast kernel handed back)
    const DevPlan& P = *L.plan;
    static const bool no_fast = dev_env("CLDN_HIP_NO_FAST_DECODE") != nullptr;  // A/B switch
    bool fast = !no_fast && P.all_varint && P.n_ops <= 8u;  // no regular ops at all (integer-only V5 cloud) is fine too
    bool all_qf32 = true;
    for (uint32_t k = 0; k < P.n_ops; ++k) all_qf32 = all_qf32 && P.ops[k].kind == OP_QF32;
    // FloatN streams (3 or 4 int32-delta tokens per point): point-parallel kernel with the Palette sections folded in;
    // it hands irregular chunks back (reg_end = kDecRedo) and k_decode_varint redoes only those
    static const bool no_points = dev_env("CLDN_HIP_NO_POINT_DECODE") != nullptr;  // A/B switch
    const bool points_kernel = fast && !no_points && all_qf32 && (P.n_ops == 3u || P.n_ops == 4u) && P.n_gorilla == 0u;
    bool stream_cols = false;  // the stream kernel stores the integer fields with the points (columns in front of it)
    bool many_used = false;  // the point kernel merged the columns of 3..8 integer channels (chunks it left: the old section kernels)
    if (points_kernel) {
      // NF: Palette sections the launch can fold into the point pass (sizes its LDS)
      // round 4: the barrier-free kernel (stage1_decode_wave.h) is the default; CLDN_HIP_POINT_KERNEL=tiles brings the
      // tile kernel back (A/B in the same binary), =w8 runs it with 8 waves per workgroup instead of 16
      static const char* pk_env = dev_env("CLDN_HIP_POINT_KERNEL");
      static const int pk = pk_env == nullptr ? 16 : (strcmp(pk_env, "tiles") == 0 ? 0 : 16);
      uint32_t nf = (L.uses_v5 && P.n_adaptive <= kFastPalFields) ? P.n_adaptive : 0u;
      // round 4: 3..8 integer channels (all of 2 or 4 bytes): their sections go to dense columns side by side in front of the
      // point kernel (stage1_decode_sections_w.h), which merges them -- every point is written once
      static const bool no_many = dev_env("CLDN_HIP_NO_SECTIONS_W") != nullptr;  // A/B switch
      bool many = !no_many && pk != 0 && L.uses_v5 && P.n_adaptive > kFastPalFields && P.n_adaptive <= kSoMaxFields && L.dsec != nullptr &&
                  L.sec_cols != nullptr && L.reg_end_pre != nullptr;
      for (uint32_t a = 0; a < P.n_adaptive && many; ++a) many = P.adaptive[a].bpv <= 4u && L.cols[a] != nullptr;
      DecColumns dcols = {};
      for (uint32_t a = 0; a < 8u; ++a) dcols.p[a] = L.cols[a];
      if (many) {
        nf = 8u;
        many_used = true;
        hipLaunchKernelGGL(k_locate_sections<4>, dim3(L.n_chunks), dim3(256), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), P.n_ops, L.reg_end_pre, L.sec_cols, L.slices_done, 0u, 0u, L.status);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_locate_sections");
        DecChunk* dsec = reinterpret_cast<DecChunk*>(L.dsec);
        hipLaunchKernelGGL(k_section_offsets, dim3(L.n_chunks), dim3(kSoThreads), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), L.n_chunks, (const uint32_t*)L.reg_end_pre, dsec, L.secs_ok, L.done_cnt, (const uint8_t*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_section_offsets");
        hipLaunchKernelGGL(k_sections_w, dim3(L.n_chunks, P.n_adaptive), dim3(kSwsThreads), 0, L.stream, P, L.streams,
                           (const DecChunk*)dsec, L.n_chunks, L.out, L.done_cnt, dv_cols_kernel() == 2 ? 2u : 1u, dcols);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_w");
        if (dv_cols_kernel() == 2) {
        } else if (dv_cols_kernel() == 1) {
          hipLaunchKernelGGL(k_sections_dv_cols, dim3(L.n_chunks, P.n_adaptive), dim3(kScfThreads), 0, L.stream, P, L.streams,
                             (const DecChunk*)dsec, L.n_chunks, L.done_cnt, dcols);
          if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_dv_cols");
        } else {
          const DevPlan S = sections_plan(P, true);
          hipLaunchKernelGGL((k_decode_stream_w<16, 0>), dim3(L.n_chunks, P.n_adaptive), dim3(16 * 64), (SwLds<16, false>::kTotal), L.stream, S,
                             L.streams, (const DecChunk*)dsec, (uint8_t*)nullptr, L.done_cnt, L.status, (const uint32_t*)nullptr, L.n_chunks, dcols, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint8_t*)nullptr);
          if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w (sections)");
        }
        hipLaunchKernelGGL(k_sections_done, dim3((L.n_chunks + 255u) / 256u), dim3(256), 0, L.stream, L.n_chunks, P.n_adaptive,
                           (const uint8_t*)L.secs_ok, (const uint32_t*)L.done_cnt, L.sec_cols, L.status, 0u);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_done");
      }
      // sections that are no small palettes go to dense columns first (every point is then written once)
      static const bool no_cols = dev_env("CLDN_HIP_NO_DECODE_COLS") != nullptr;  // A/B switch
      static const bool no_hint = dev_env("CLDN_HIP_NO_PALETTE_HINT") != nullptr;  // A/B switch
      bool cols = !no_cols && !many && nf != 0u && L.cols[0] != nullptr && L.sec_cols != nullptr && !(L.palette_hint && !no_hint && pk != 0);
      for (uint32_t a = 0; a < P.n_adaptive && cols; ++a) cols = P.adaptive[a].bpv <= 4u && L.cols[a] != nullptr;
      if (cols) {
        {
          static const int lw = dev_env_int("CLDN_HIP_LOCATE_WAVES", 0);  // A/B switch
          // (16 waves per chunk measured slower on C3 / C4 / C5: 0.452 / 0.572 / 0.140 against 0.433 / 0.552 / 0.137 ms; small
          // batches -- the ones that take the SPLIT launches -- have CUs to spare: 4 -> 16 waves per chunk)
          const bool wide = lw >= 16 || (lw == 0 && L.n_chunks <= 64u);
          if (wide) hipLaunchKernelGGL(k_locate_sections<16>, dim3(L.n_chunks), dim3(1024), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), P.n_ops, L.reg_end_pre, L.sec_cols, L.slices_done, 0u, 1u, L.status);
          else hipLaunchKernelGGL(k_locate_sections<4>, dim3(L.n_chunks), dim3(256), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), P.n_ops, L.reg_end_pre, L.sec_cols, L.slices_done, 0u, 1u, L.status);
        }
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_locate_sections");
        static const bool no_scf = dev_env("CLDN_HIP_NO_FAST_COLS") != nullptr;  // A/B switch
        const bool scf = !no_scf && P.n_adaptive == 1u && L.slice_rec != nullptr && L.slices_done != nullptr;
        if (scf && P.adaptive[0].bpv <= 4u && L.dv_hint != 1u) {
          // round 6: a DeltaVarint section by the point decoder's machinery, one workgroup of 16 waves per chunk
          // (stage1_decode_dv.h); chunks it hands back (long tokens, other modes) go on to the kernels below.
          // (dv_hint 1: the codec's last calls had no such section -- the launch would find nothing to do)
          hipLaunchKernelGGL(k_section_dv_w, dim3(L.n_chunks), dim3(kDvWaves * 64u), (DvwLds::kTotal), L.stream, P, L.streams,
                             reinterpret_cast<const DecChunk*>(L.chunks), L.cols[0], (const uint32_t*)L.reg_end_pre, L.sec_cols,
                             (const uint32_t*)L.slices_done, L.status);
          if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_section_dv_w");
        }
        if (scf && !(L.dv_hint == 2u && P.adaptive[0].bpv <= 4u)) {  // (dv_hint 2: k_section_dv_w took every chunk of the last calls)
          // workgroups per chunk: one when the batch has chunks enough to fill the chip (C3, 512 chunks: 0.404 / 0.400 /
          // 0.402 / 0.404 ms with 1 / 2 / 4 / 8; workgroups that find nothing to share cost C4 about 20 us per
          // 1024 of them), more for a single cloud's few chunks
          static const int parts_env = dev_env_int("CLDN_HIP_SCF_PARTS", 0);  // A/B switch
          uint32_t parts = parts_env > 0 ? (uint32_t)parts_env : (512u + L.n_chunks - 1u) / L.n_chunks;
          parts = std::min<uint32_t>(std::max<uint32_t>(parts, 1u), kScfMaxParts);
          hipLaunchKernelGGL(k_sections_cols_fast, dim3(L.n_chunks * parts), dim3(kScfThreads), 0, L.stream, P, L.streams,
                             reinterpret_cast<const DecChunk*>(L.chunks), L.cols[0], L.reg_end_pre, L.sec_cols, L.slices_done,
                             L.slice_rec, L.slice_epoch, parts);
          if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_cols_fast");
        }
        hipLaunchKernelGGL(k_decode_sections_cols, dim3(L.n_chunks), dim3(kDvThreads), (DecSecLds::kTotal), L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), P.n_ops, L.cols[0], L.cols[1], L.reg_end_pre, L.sec_cols,
                           scf ? 1u : 0u);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_sections_cols");
      }
      const uint8_t* c0 = cols ? L.cols[0] : nullptr;
      const uint8_t* c1 = cols ? L.cols[1] : nullptr;
      const uint8_t* sc = (cols || many) ? L.sec_cols : nullptr;
      const uint32_t fill_zero = L.fill_zero ? 1u : 0u;
#define LAUNCH_POINTS(NOPS_, NF_)                                                                                         \
  hipLaunchKernelGGL((k_decode_points<NOPS_, NF_>), dim3(L.n_chunks), dim3(kFpThreads), (FpLds<NOPS_, NF_>::kTotal), L.stream, \
                     P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.sec_done, L.uses_v5,  \
                     L.status, c0, c1, L.reg_end_pre, sc, fill_zero)
#ifdef CLDN_DEV  // (CLDN_HIP_POINT_KERNEL=tiles: the round-3 tile kernel, an A/B reference of the development build)
#define LAUNCH_POINTS_TILES(NOPS_, NF_) if (pk == 0) LAUNCH_POINTS(NOPS_, NF_); else
#else
#define LAUNCH_POINTS_TILES(NOPS_, NF_)
#endif
#define LAUNCH_POINTS_W(NOPS_, NF_, NW_, WPE_)                                                                            \
  hipLaunchKernelGGL((k_decode_points_w<NOPS_, NF_, NW_, WPE_>), dim3(L.n_chunks), dim3(NW_ * 64), (WpLds<NOPS_, NF_, NW_>::kTotal), \
                     L.stream, P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.sec_done,   \
                     L.uses_v5, L.status, c0, c1, L.reg_end_pre, sc, fill_zero, dcols, WpSplit{})
#define LAUNCH_POINTS_W_SM(NOPS_, NF_, SM_)                                                                               \
  hipLaunchKernelGGL((k_decode_points_w<NOPS_, NF_, 16, 8, SM_>), dim3(L.n_chunks), dim3(16 * 64), (WpLds<NOPS_, NF_, 16>::kTotal), \
                     L.stream, P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.sec_done,   \
                     L.uses_v5, L.status, c0, c1, L.reg_end_pre, sc, fill_zero, dcols, WpSplit{})
// SPLIT launches (small batches): counts, PASS 1, carries, PASS 2
#define LAUNCH_POINTS_SPLIT(NOPS_, NF_, SM_)                                                                              \
  {                                                                                                                      \
    hipLaunchKernelGGL(k_wp_counts, dim3(L.n_chunks), dim3(1024), 0, L.stream, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), \
                       wsp, (uint32_t)(NOPS_ + 1));                                                                      \
    hipLaunchKernelGGL((k_decode_points_w<NOPS_, NF_, 16, 8, SM_, 1>), dim3(L.n_chunks, split_parts), dim3(16 * 64),      \
                       (WpLds<NOPS_, NF_, 16>::kTotal), L.stream, P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), \
                       L.out, L.reg_end, L.sec_done, L.uses_v5, L.status, c0, c1, L.reg_end_pre, sc, fill_zero, dcols, wsp); \
    hipLaunchKernelGGL(k_wp_carry<NOPS_>, dim3(L.n_chunks), dim3(64), 0, L.stream, L.streams,                            \
                       reinterpret_cast<const DecChunk*>(L.chunks), wsp);                                                \
    hipLaunchKernelGGL((k_decode_points_w<NOPS_, NF_, 16, 8, SM_, 2>), dim3(L.n_chunks, split_parts), dim3(16 * 64),      \
                       (WpLds<NOPS_, NF_, 16>::kTotal), L.stream, P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), \
                       L.out, L.reg_end, L.sec_done, L.uses_v5, L.status, c0, c1, L.reg_end_pre, sc, fill_zero, dcols, wsp); \
  }
#define LAUNCH_POINTS_ANY(NOPS_, NF_)                      \
  {                                                        \
    LAUNCH_POINTS_TILES(NOPS_, NF_)                        \
    LAUNCH_POINTS_W(NOPS_, NF_, 16, 8);                    \
  }
      // store-mode instantiations of the two headline layouts (XYZ, XYZ + one 16-bit field): the layout facts the kernel
      // otherwise keeps as uniform flags are checked here. CLDN_HIP_NO_STORE_MODES=1: A/B switch
      static const bool no_sm = dev_env("CLDN_HIP_NO_STORE_MODES") != nullptr;
      int sm = 0;
      if (!no_sm && pk == 16 && P.n_ops == 3u && nf <= 1u) {
        bool ok = ((P.point_step | P.ops[0].offset) & 3u) == 0u && P.ops[0].offset != 0xffffffffu &&
                  P.ops[1].offset == P.ops[0].offset + 4u && P.ops[2].offset == P.ops[0].offset + 8u;
        if (nf == 1u) ok = ok && P.adaptive[0].bpv == 2u && ((P.adaptive[0].offset | P.point_step) & 1u) == 0u;
        if (ok) {
          sm = 1;
          if (nf == 1u && fill_zero && P.point_step == 16u && P.ops[0].offset == 0u && P.adaptive[0].offset == 12u &&
              ((uintptr_t)L.out & 15u) == 0u)
            sm = 2;
        }
      }
      // SPLIT launches for batches that do not fill the chip (L.wp_parts: wp_split_parts(n_chunks), or what the test hook
      // cldn_hip_debug_decode_split asked for)
      const uint32_t split_parts = L.wp_split == nullptr || pk != 16 ? 1u : std::max(1u, L.wp_parts);
      WpSplit wsp = {};
      if (split_parts > 1u) {
        uint8_t* w = (uint8_t*)L.wp_split;
        wsp.maxp = L.wp_maxp;
        wsp.flags = (uint32_t*)w;
        w += ((size_t)L.n_chunks * 16u + 255u) & ~size_t(255);
        wsp.t0 = (uint32_t*)w;
        w += (size_t)L.n_chunks * L.wp_maxp * 4u;
        wsp.agg = (int32_t*)w;
        w += (size_t)L.n_chunks * L.wp_maxp * 20u;
        wsp.carry = (int32_t*)w;
      }
      ev_before();
      if (split_parts > 1u && P.n_ops == 3u && sm == 1) {
        if (nf == 0u) LAUNCH_POINTS_SPLIT(3, 0, 1)
        else LAUNCH_POINTS_SPLIT(3, 1, 1)
      } else if (split_parts > 1u && P.n_ops == 3u && sm == 2) {
        LAUNCH_POINTS_SPLIT(3, 1, 2)
      } else if (split_parts > 1u && P.n_ops == 3u && sm == 0) {
        if (nf == 0u) LAUNCH_POINTS_SPLIT(3, 0, 0)
        else if (nf == 1u) LAUNCH_POINTS_SPLIT(3, 1, 0)
        else if (nf == 2u) LAUNCH_POINTS_SPLIT(3, 2, 0)
        else LAUNCH_POINTS_SPLIT(3, 8, 0)
      } else if (split_parts > 1u && P.n_ops == 4u && sm == 0) {
        if (nf == 0u) LAUNCH_POINTS_SPLIT(4, 0, 0)
        else if (nf == 1u) LAUNCH_POINTS_SPLIT(4, 1, 0)
        else if (nf == 2u) LAUNCH_POINTS_SPLIT(4, 2, 0)
        else LAUNCH_POINTS_SPLIT(4, 8, 0)
      } else if (P.n_ops == 3u && sm == 1) {
        if (nf == 0u) LAUNCH_POINTS_W_SM(3, 0, 1);
        else LAUNCH_POINTS_W_SM(3, 1, 1);
      } else if (P.n_ops == 3u && sm == 2) {
        LAUNCH_POINTS_W_SM(3, 1, 2);
      } else if (P.n_ops == 3u) {
        if (nf == 0u) LAUNCH_POINTS_ANY(3, 0)
        else if (nf == 1u) LAUNCH_POINTS_ANY(3, 1)
        else if (nf == 2u) LAUNCH_POINTS_ANY(3, 2)
        else LAUNCH_POINTS_W(3, 8, 16, 8);
      } else {
        if (nf == 0u) LAUNCH_POINTS_ANY(4, 0)
        else if (nf == 1u) LAUNCH_POINTS_ANY(4, 1)
        else if (nf == 2u) LAUNCH_POINTS_ANY(4, 2)
        else LAUNCH_POINTS_W(4, 8, 16, 8);
      }
#undef LAUNCH_POINTS_ANY
#undef LAUNCH_POINTS_TILES
#undef LAUNCH_POINTS_SPLIT
#undef LAUNCH_POINTS_W_SM
#undef LAUNCH_POINTS_W
#undef LAUNCH_POINTS
      ev_after();
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_points");
    }
    // Behind k_decode_points, for plans whose sections it can fold, the rest is normally idle: one launch covers it
    // (plans with more adaptive fields keep the separate kernels: their Palette chunks really run k_decode_sections_small,
    // which wants its own, smaller LDS footprint)
    static const bool no_tail = dev_env("CLDN_HIP_NO_DECODE_TAIL") != nullptr;  // A/B switch
    const bool sections_any = L.uses_v5 && P.n_adaptive > 0u;
    if (points_kernel && !no_tail && (!sections_any || P.n_adaptive <= kFastPalFields)) {
      const uint32_t lds = sections_any ? std::max<uint32_t>(std::max<uint32_t>((uint32_t)Dv2Lds<4, false, 16>::kTotal, kSmallSecLds), (uint32_t)DecSecLds::kTotal)
                                        : (uint32_t)Dv2Lds<4, false, 16>::kTotal;
      hipLaunchKernelGGL(k_decode_tail, dim3(L.n_chunks), dim3(kDvThreads), lds, L.stream, P, L.streams,
                         reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.sec_done, L.status, L.uses_v5,
                         sections_any ? 1u : 0u);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_tail");
      return CLDN_HIP_OK;
    }
    // regular streams with raw (FieldEncoderCopy) fields between the varints: k_mark_token_ends lays out where the tokens
    // end, the 64-bit token kernel takes the ends from there (the plan says whether the stream has that form)
    static const bool no_mixed = dev_env("CLDN_HIP_NO_MIXED_DECODE") != nullptr;  // A/B switch
    const bool mixed = !fast && !no_fast && !no_mixed && P.varint_and_raw != 0u && L.token_ends != nullptr;
    if (mixed) {
      static const bool no_stream_mixed = dev_env("CLDN_HIP_NO_STREAM_KERNEL") != nullptr;  // A/B switch
      const bool stream_ok = !no_stream_mixed && P.n_ops <= kSwMaxOps && P.max_regular_bytes <= kSwMaxPointBytes;
      bool all_raw = true;  // points of a fixed size: the stream kernel needs no bitmap
      for (uint32_t k = 0; k < P.n_ops; ++k) all_raw = all_raw && (P.ops[k].kind == OP_COPY || P.ops[k].kind == OP_XOR32 || P.ops[k].kind == OP_XOR64);
      // layouts with varints AND raw fields: the stream kernel finds the points from their form (FORM instantiation);
      // CLDN_HIP_STREAM_BITMAP=1 keeps k_mark_token_ends' bitmap in front of it (A/B switch)
      static const bool force_bitmap = dev_env("CLDN_HIP_STREAM_BITMAP") != nullptr;
      // round 5: forms of at most 16 states get their token ends from k_mark_ends_automaton (stage1_decode_automaton.h) and
      // the stream kernel's bitmap mode; CLDN_HIP_FORM_KERNEL=1 keeps the round-4 FORM instantiation (A/B switch)
      static const bool form_kernel = dev_env("CLDN_HIP_FORM_KERNEL") != nullptr;
      const bool automaton = stream_ok && !all_raw && !force_bitmap && !form_kernel && automaton_states(P) != 0u && L.token_ends != nullptr;
      const bool form = stream_ok && !all_raw && !force_bitmap && !automaton;
      // streams of fixed-size tokens only (lossless floats, raw copies; <= 8 fields): nothing to find, k_decode_fixed.
      // CLDN_HIP_NO_FIXED_DECODE=1: the stream kernel (A/B switch)
      static const bool no_fixed_dec = dev_env("CLDN_HIP_NO_FIXED_DECODE") != nullptr;
      uint32_t fixed_bytes = 0u;
      if (!no_fixed_dec && all_raw && P.n_ops <= kFxMaxOps && P.n_gorilla == 0u)
        for (uint32_t k = 0; k < P.n_ops; ++k) fixed_bytes += P.ops[k].size;
      const bool bitmap = !(stream_ok && all_raw) && !form && fixed_bytes == 0u;
      if (bitmap && automaton) {
        if (automaton_states(P) <= 8u)
          hipLaunchKernelGGL(k_mark_ends_automaton<uint32_t>, dim3(L.n_chunks), dim3(kMaWaves * 64u), 0, L.stream, P, L.streams,
                             reinterpret_cast<const DecChunk*>(L.chunks), L.token_ends, L.reg_end);
        else
          hipLaunchKernelGGL(k_mark_ends_automaton<uint64_t>, dim3(L.n_chunks), dim3(kMaWaves * 64u), 0, L.stream, P, L.streams,
                             reinterpret_cast<const DecChunk*>(L.chunks), L.token_ends, L.reg_end);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_mark_ends_automaton");
      } else if (bitmap) {
        hipLaunchKernelGGL(k_mark_token_ends, dim3(L.n_chunks), dim3(kMtThreads), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), L.token_ends, L.reg_end);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_mark_token_ends");
      }
      ev_before();
      if (fixed_bytes != 0u) {
        hipLaunchKernelGGL(k_decode_fixed, dim3(L.n_chunks), dim3(kFxThreads), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status, fixed_bytes);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_fixed");
      } else if (form) {
        hipLaunchKernelGGL((k_decode_stream_w<12, 1>), dim3(L.n_chunks), dim3(12 * 64), (SwLds<12, 1>::kTotal), L.stream, P,
                           L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status, (const uint32_t*)nullptr, 0u, DecColumns{}, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint8_t*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w (form)");
      } else if (stream_ok) {
        // round 4: the barrier-free stream kernel reads the token ends from the bitmap (chunks it finds irregular go to the
        // serial decoder, like the chunks k_mark_token_ends gave up on)
        hipLaunchKernelGGL((k_decode_stream_w<16, false>), dim3(L.n_chunks), dim3(16 * 64), (SwLds<16, false>::kTotal), L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status,
                           bitmap ? (const uint32_t*)L.token_ends : (const uint32_t*)nullptr, 0u, DecColumns{}, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint8_t*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w (mixed)");
      } else {
        hipLaunchKernelGGL((k_decode_varint<8, true>), dim3(L.n_chunks), dim3(kDvThreads), (Dv2Lds<8, true, 8>::kTotal),
                           L.stream, P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status, 0u,
                           (const uint32_t*)L.token_ends);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_varint (mixed)");
      }
      ev_after();
      fast = true;  // from here on like any stream the parallel kernels have taken
    } else if (fast) {
      // round 4: general streams of varint tokens go through the barrier-free stream kernel first (stage1_decode_stream.h);
      // the tile kernel behind it only redoes the chunks it hands back. CLDN_HIP_NO_STREAM_KERNEL=1: A/B switch
      static const bool no_stream = dev_env("CLDN_HIP_NO_STREAM_KERNEL") != nullptr;
      const bool stream_kernel = !no_stream && !points_kernel && P.n_ops <= kSwMaxOps && P.max_regular_bytes <= kSwMaxPointBytes;
      // round 4: the integer fields of such a stream (1..8 of 2 / 4 bytes) go to dense columns FIRST -- the sections are
      // found by counting token ends (k_locate_sections), sized and decoded side by side -- and the stream kernel stores them
      // with the points: every point is written once (DDS layout with 1 us stamps: the ring column behind the points cost
      // 0.17 of 0.74 ms). Chunks whose sections did not all arrive take the passes below. CLDN_HIP_NO_STREAM_COLS=1: A/B switch
      static const bool no_stream_cols = dev_env("CLDN_HIP_NO_STREAM_COLS") != nullptr;
      stream_cols = stream_kernel && !no_stream_cols && L.uses_v5 && P.n_adaptive >= 1u && P.n_adaptive <= kSoMaxFields && L.dsec != nullptr &&
                    L.sec_cols != nullptr && L.reg_end_pre != nullptr && L.slices_done != nullptr;
      for (uint32_t a = 0; a < P.n_adaptive && stream_cols; ++a) stream_cols = P.adaptive[a].bpv <= 4u && L.cols[a] != nullptr;
      if (stream_cols) {
        DecColumns dcols = {};
        for (uint32_t a = 0; a < 8u; ++a) dcols.p[a] = L.cols[a];
        hipLaunchKernelGGL(k_locate_sections<4>, dim3(L.n_chunks), dim3(256), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), P.n_ops, L.reg_end_pre, L.sec_cols, L.slices_done, 1u, 0u, L.status);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_locate_sections");
        DecChunk* dsec = reinterpret_cast<DecChunk*>(L.dsec);
        hipLaunchKernelGGL(k_section_offsets, dim3(L.n_chunks), dim3(kSoThreads), 0, L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), L.n_chunks, (const uint32_t*)L.reg_end_pre, dsec, L.secs_ok, L.done_cnt, (const uint8_t*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_section_offsets");
        hipLaunchKernelGGL(k_sections_w, dim3(L.n_chunks, P.n_adaptive), dim3(kSwsThreads), 0, L.stream, P, L.streams,
                           (const DecChunk*)dsec, L.n_chunks, L.out, L.done_cnt, dv_cols_kernel() == 2 ? 2u : 1u, dcols);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_w");
        if (dv_cols_kernel() == 2) {
        } else if (dv_cols_kernel() == 1) {
          hipLaunchKernelGGL(k_sections_dv_cols, dim3(L.n_chunks, P.n_adaptive), dim3(kScfThreads), 0, L.stream, P, L.streams,
                             (const DecChunk*)dsec, L.n_chunks, L.done_cnt, dcols);
          if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_dv_cols");
        } else {
          const DevPlan S = sections_plan(P, true);
          hipLaunchKernelGGL((k_decode_stream_w<16, 0>), dim3(L.n_chunks, P.n_adaptive), dim3(16 * 64), (SwLds<16, false>::kTotal), L.stream, S,
                             L.streams, (const DecChunk*)dsec, (uint8_t*)nullptr, L.done_cnt, L.status, (const uint32_t*)nullptr, L.n_chunks, dcols, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint8_t*)nullptr);
          if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w (sections)");
        }
        hipLaunchKernelGGL(k_sections_done, dim3((L.n_chunks + 255u) / 256u), dim3(256), 0, L.stream, L.n_chunks, P.n_adaptive,
                           (const uint8_t*)L.secs_ok, (const uint32_t*)L.done_cnt, L.sec_cols, L.status, 0u);
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_done");
        ev_before();
        hipLaunchKernelGGL((k_decode_stream_w<16, false>), dim3(L.n_chunks), dim3(16 * 64), (SwLds<16, false>::kTotal), L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status, (const uint32_t*)nullptr, 0u, dcols,
                           (const uint8_t*)L.sec_cols, (const uint32_t*)L.reg_end_pre, L.sec_done);
        ev_after();
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w");
      } else if (stream_kernel) {
        ev_before();
        hipLaunchKernelGGL((k_decode_stream_w<16, false>), dim3(L.n_chunks), dim3(16 * 64), (SwLds<16, false>::kTotal), L.stream, P, L.streams,
                           reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status, (const uint32_t*)nullptr, 0u, DecColumns{}, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint8_t*)nullptr);
        ev_after();
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w");
      }
      const uint32_t redo_only = (points_kernel || stream_kernel) ? 1u : 0u;
      if (all_qf32 && P.n_ops <= 4u)
        hipLaunchKernelGGL((k_decode_varint<4, false>), dim3(L.n_chunks), dim3(kDvThreads), (Dv2Lds<4, false, 16>::kTotal),
                           L.stream, P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status,
                           redo_only, (const uint32_t*)nullptr);
      else
        hipLaunchKernelGGL((k_decode_varint<8, true>), dim3(L.n_chunks), dim3(kDvThreads), (Dv2Lds<8, true, 8>::kTotal),
                           L.stream, P, L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status, redo_only,
                           (const uint32_t*)nullptr);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_varint");
    }
    // round 4: streams with ONE Gorilla-coded field (FLOAT64 without resolution, wire version >= 4: the reference's own DDS
    // sample layout) next to varints and raw fields: MODE 2 of the stream kernel; CLDN_HIP_NO_GORILLA_KERNEL=1: A/B switch
    static const bool no_gor = dev_env("CLDN_HIP_NO_GORILLA_KERNEL") != nullptr;
    if (!fast && !no_fast && !no_gor && P.n_gorilla >= 1u && P.n_ops >= 2u && P.n_ops <= kSwMaxOps && P.max_regular_bytes <= kSwMaxPointBytes) {
      bool ok = true;
      for (uint32_t k = 0; k < P.n_ops && ok; ++k) {
        const uint32_t kd = P.ops[k].kind, sz = P.ops[k].size;
        if (kd == OP_COPY || kd == OP_XOR32 || kd == OP_XOR64) ok = sz == 1u || sz == 2u || sz == 4u || sz == 8u;
        else ok = kd == OP_QF32 || kd == OP_LOSSY_F32 || kd == OP_LOSSY_F64 || kd == OP_INT || kd == OP_GORILLA64;
      }
      if (ok) {
        ev_before();
        hipLaunchKernelGGL((k_decode_stream_w<12, 2>), dim3(L.n_chunks), dim3(12 * 64), (SwLds<12, 2>::kTotal), L.stream, P,
                           L.streams, reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.reg_end, L.status, (const uint32_t*)nullptr, 0u, DecColumns{}, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint8_t*)nullptr);
        ev_after();
        if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w (gorilla)");
        fast = true;  // from here on like any stream the parallel kernels have taken
      }
    }
    const bool fast_sections = fast && L.uses_v5 && P.n_adaptive > 0u;
    // round 4: the sections of a chunk side by side (stage1_decode_sections_w.h): sized without decoding, then one
    // workgroup per (chunk, field); chunks it does not finish stay with the kernels below. CLDN_HIP_NO_SECTIONS_W=1: A/B switch
    static const bool no_sw = dev_env("CLDN_HIP_NO_SECTIONS_W") != nullptr;
    // (stream_cols: the sections went out with the points; what is left -- irregular chunks -- takes the two launches below)
    bool sections_w = fast_sections && !no_sw && !many_used && !stream_cols && L.dsec != nullptr && P.n_adaptive <= kSoMaxFields;
    for (uint32_t a = 0; a < P.n_adaptive && sections_w; ++a) sections_w = P.adaptive[a].bpv <= 4u;
    if (sections_w) {
      DecChunk* dsec = reinterpret_cast<DecChunk*>(L.dsec);
      hipLaunchKernelGGL(k_section_offsets, dim3(L.n_chunks), dim3(kSoThreads), 0, L.stream, P, L.streams,
                         reinterpret_cast<const DecChunk*>(L.chunks), L.n_chunks, (const uint32_t*)L.reg_end, dsec, L.secs_ok, L.done_cnt, stream_cols ? (const uint8_t*)L.sec_done : (const uint8_t*)nullptr);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_section_offsets");
      hipLaunchKernelGGL(k_sections_w, dim3(L.n_chunks, P.n_adaptive), dim3(kSwsThreads), 0, L.stream, P, L.streams,
                         (const DecChunk*)dsec, L.n_chunks, L.out, L.done_cnt, 0u, DecColumns{});
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_w");
      // DeltaVarint sections: streams of n tokens of one integer op -> the stream kernel, row a of the grid = field a
      const DevPlan S = sections_plan(P, false);
      hipLaunchKernelGGL((k_decode_stream_w<16, 0>), dim3(L.n_chunks, P.n_adaptive), dim3(16 * 64), (SwLds<16, false>::kTotal), L.stream, S,
                         L.streams, (const DecChunk*)dsec, L.out, L.done_cnt, L.status, (const uint32_t*)nullptr, L.n_chunks, DecColumns{}, (const uint8_t*)nullptr, (const uint32_t*)nullptr, (uint8_t*)nullptr);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_stream_w (sections)");
      hipLaunchKernelGGL(k_sections_done, dim3((L.n_chunks + 255u) / 256u), dim3(256), 0, L.stream, L.n_chunks, P.n_adaptive,
                         (const uint8_t*)L.secs_ok, (const uint32_t*)L.done_cnt, L.sec_done, L.status, stream_cols ? 2u : 1u);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_sections_done");
      hipLaunchKernelGGL(k_decode_sections, dim3(L.n_chunks), dim3(kDvThreads), (DecSecLds::kTotal), L.stream, P, L.streams,
                         reinterpret_cast<const DecChunk*>(L.chunks), L.out, (const uint32_t*)L.reg_end, L.sec_done, L.status);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_sections");
    } else if (fast_sections) {
      hipLaunchKernelGGL(k_decode_sections_small, dim3(L.n_chunks), dim3(kDvThreads), kSmallSecLds, L.stream, P, L.streams,
                         reinterpret_cast<const DecChunk*>(L.chunks), L.out, (const uint32_t*)L.reg_end, L.sec_done, L.status,
                         (points_kernel || stream_cols) ? 1u : 0u);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_sections_small");
      hipLaunchKernelGGL(k_decode_sections, dim3(L.n_chunks), dim3(kDvThreads), (DecSecLds::kTotal), L.stream, P, L.streams,
                         reinterpret_cast<const DecChunk*>(L.chunks), L.out, (const uint32_t*)L.reg_end, L.sec_done, L.status);
      if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_sections");
    }
    hipLaunchKernelGGL(k_decode_general, dim3(L.n_chunks), dim3(64), 0, L.stream, P, L.streams,
                       reinterpret_cast<const DecChunk*>(L.chunks), L.out, L.uses_v5, fast ? 1u : 0u,
                       (const uint32_t*)L.reg_end, fast_sections ? (const uint8_t*)L.sec_done : (const uint8_t*)nullptr,
                       L.status);
    if ((e = hipGetLastError()) != hipSuccess) return hip_fail(e, "k_decode_general");
  }
  return CLDN_HIP_OK;
}

}  // namespace cldn
