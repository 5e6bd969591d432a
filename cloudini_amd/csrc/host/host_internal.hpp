// Internal glue between the host translation units (not installed).
#pragma once
#include <cstddef>
#include <cstdint>

namespace Cloudini {
namespace amd_detail {

// cldn_hip_viz_preprocess on host buffers through a pooled codec; throws std::runtime_error on failure.
uint64_t vizPreprocessOnDevice(const uint8_t* points, size_t n_points, uint32_t point_step, uint32_t xyz_offset,
                               float resolution, uint8_t* out, size_t out_capacity);

// Stage-2 (LZ4/ZSTD) threads per encode()/decode() call, the caller included (bounded worker pool, cloudini.cpp).
unsigned stage2Threads();
void setStage2Threads(unsigned n);

}  // namespace amd_detail
}  // namespace Cloudini
