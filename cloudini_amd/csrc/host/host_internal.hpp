// Internal glue between the host translation units (not installed).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

#include "cloudini_lib/cloudini.hpp"

namespace Cloudini {
namespace amd_detail {

// cldn_hip_viz_preprocess on host buffers through a pooled codec; throws std::runtime_error on failure.
uint64_t vizPreprocessOnDevice(const uint8_t* points, size_t n_points, uint32_t point_step, uint32_t xyz_offset,
                               float resolution, uint8_t* out, size_t out_capacity);

// ---- building blocks of the batch transcoder (batch_transcoder.cpp) ----
// One batched stage-1 encode of n clouds that share `info`'s schema, each in its own host buffer: the framed streams
// ([u32 size][payload] per 32768-point chunk) land back to back in `stage1`; stream_offsets has n + 1 entries,
// chunk_sizes one payload size per chunk in batch order. Throws std::runtime_error.
// `grow(bytes)` is called once with the exact size of the batch's streams and returns where they go (page-locked memory
// makes the copy back fast).
void encodeStage1Batch(const Cloudini::EncodingInfo& info, const uint8_t* const* cloud_ptrs, const uint64_t* cloud_points,
                       uint32_t n_clouds, const std::function<uint8_t*(uint64_t)>& grow, std::vector<uint64_t>& stream_offsets,
                       std::vector<uint32_t>& chunk_sizes);
// detail::CompressChunk (src/codec_common.cpp:220-258) and its worst-case output size
uint32_t compressChunkTo(Cloudini::CompressionOption opt, const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap);
size_t compressedChunkBound(Cloudini::CompressionOption opt, size_t stage1_bytes);
// ---- the way back (decode direction of the batch transcoder) ----
struct ChunkRef {
  const uint8_t* src;  // stage-2 payload of the chunk (behind its [u32 size])
  uint32_t size;
};
// The chunk chain of one compressed cloud (header already removed), validated like PointcloudDecoder::decode does
// (src/cloudini.cpp:645-664; same error strings). `points` = width * height of the header.
void walkCompressedChunks(Cloudini::ConstBufferView data, uint64_t points, std::vector<ChunkRef>& refs);
// detail::DecompressChunk (src/codec_common.cpp:260-300): LZ4 block / ZSTD frame / plain copy -> stage-1 bytes
// LZ4 streams with stage 2 on the device (cldn_hip_codec_set_stage2): off unless CLOUDINI_AMD_DEVICE_LZ4=1 or the setter
bool deviceLz4();
void setDeviceLz4(bool on);
int deviceLz4Level();            // 0 host pool, 1 CLDN_HIP_STAGE2_LZ4, 2 CLDN_HIP_STAGE2_LZ4_FAST
void setDeviceLz4Level(int level);

uint32_t decompressChunkTo(Cloudini::CompressionOption opt, const uint8_t* src, size_t size, uint8_t* dst, size_t dst_cap);
// worst-case stage-1 bytes of one 32768-point chunk of this schema (without its [u32 size])
size_t stage1ChunkBound(const Cloudini::EncodingInfo& info);
// One batched stage-1 decode: n clouds of the same schema, framed stage-1 streams back to back in `streams` (offsets has
// n + 1 entries), decoded points back to back in `out` (cloud k: cloud_points[k] * point_step bytes). Bytes of a point that
// no field covers keep the content of `out`, or read 0 with out_is_zero (the caller does not need `out`'s content:
// CLDN_HIP_FILL_ZERO). Throws std::runtime_error.
void decodeStage1Batch(const Cloudini::EncodingInfo& info, const uint8_t* streams, const uint64_t* offsets,
                       const uint64_t* cloud_points, uint32_t n_clouds, uint8_t* out, uint64_t out_capacity,
                       bool out_is_zero = false);
// fn(i) for i in [0, n) on the bounded stage-2 pool (the caller takes part)
void runOnStage2Pool(size_t n, const std::function<void(size_t)>& fn);

// Stage-2 (LZ4/ZSTD) threads per encode()/decode() call, the caller included (bounded worker pool, cloudini.cpp).
unsigned stage2Threads();
void setStage2Threads(unsigned n);

}  // namespace amd_detail
}  // namespace Cloudini
