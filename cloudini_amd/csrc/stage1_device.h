// stage1_device.h -- structures shared by the host side of the C ABI (hip_abi.hip) and the kernels.
#pragma once

#include <stdint.h>
#include <stdlib.h>

#include <hip/hip_runtime.h>

namespace cldn {

// A/B and profiling switches (the CLDN_HIP_* environment variables the development rounds used) exist only in a build with
// -DCLDN_DEV (tools/dev/variants2.sh): the shipped library reads none of them -- every `dev_env(...)` below is a constant
// null pointer there, the branch behind it is folded away, and no environment variable can change what the kernels write.
#ifdef CLDN_DEV
inline const char* dev_env(const char* name) { return getenv(name); }
inline int dev_env_int(const char* name, int otherwise) {
  const char* e = getenv(name);
  return e ? atoi(e) : otherwise;
}
#else
constexpr const char* dev_env(const char*) { return nullptr; }
constexpr int dev_env_int(const char*, int otherwise) { return otherwise; }
#endif

constexpr uint32_t kPointsPerChunk = 32768;  // detail::kPointsPerChunk, src/codec_common.hpp:28
constexpr uint32_t kDecInlineClouds = 8;      // decode calls of at most this many clouds pass their per-cloud tables as a kernel argument
constexpr uint32_t kProbePoints = 4096;      // kAdaptiveModeProbePoints, src/v5_codec.cpp:76
// Limits of the launch-argument plan (DevPlan travels to the kernels by value: 64 ops * 32 B + 64 adaptive fields * 8 B plus
// the column pointer table stay below the 4 KB a launch may carry). Schemas beyond them -- the reference has no limits,
// src/codec_common.cpp:116-153, src/v5_codec.cpp:719-740 -- take the WIDE route (stage1_wide.h): the plan lives in device
// memory (WidePlan), one workgroup per chunk writes the chunk's payload in one piece, the decoder is the serial one.
constexpr int kMaxOps = 64;                  // regular tokens per point
constexpr int kMaxAdaptive = 64;             // V5 adaptive-int fields per schema
constexpr uint32_t kMaxPointStep = 1024;     // generic kernel: points wider than 256 bytes go in 64-point tiles (2 x 64 KiB of LDS)
constexpr uint32_t kWidePointStep = 256;     // up to here a 1024-thread tile of 16 KiB holds at least 64 points

// One regular token per point. A FieldEncoderFloatN_Lossy (3 or 4 fused floats) is flattened into 3-4 OP_QF32
// ops: its lanes are independent (src/field_encoder.cpp:42-91) and emit in lane order.
enum : uint8_t {
  OP_QF32 = 0,       // FloatN lane: float32 * m, round-half-even, int32 wrap-around delta, varint / NaN marker
  OP_LOSSY_F32 = 1,  // FieldEncoderFloat_Lossy<float>: std::round, int64 delta
  OP_LOSSY_F64 = 2,  // FieldEncoderFloat_Lossy<double>
  OP_INT = 3,        // FieldEncoderInt<T>: int64 delta varint
  OP_COPY = 4,       // FieldEncoderCopy: raw bytes
  OP_XOR32 = 5,      // FieldEncoderFloat_XOR<float>
  OP_XOR64 = 6,      // FieldEncoderFloat_XOR<double>
  OP_GORILLA64 = 7   // FieldEncoderFloat_Gorilla<double>: tokens precomputed by k_gorilla_tokens (DevOp::mult_f unused,
                     // DevOp::type = index of the op's token buffer)
};

constexpr int kMaxGorilla = 64;  // Gorilla-coded FLOAT64 fields per schema

struct DevOp {
  uint8_t kind;
  uint8_t type;       // Cloudini::FieldType (OP_INT)
  uint8_t size;       // field size in bytes
  uint8_t max_bytes;  // worst-case encoded bytes
  uint32_t offset;    // field offset inside the point; 0xFFFFFFFF on decode = kDecodeButSkipStore
  float mult_f;       // encode multiplier (float paths)
  float res_f;        // decode multiplier
  double mult_d;
  double res_d;
};

struct DevAdaptive {
  uint32_t offset;
  uint8_t type;
  uint8_t bpv;
  uint8_t pad[2];
};

struct DevPlan {
  uint32_t point_step;
  uint32_t n_ops;
  uint32_t n_adaptive;
  uint32_t max_regular_bytes;  // worst-case regular-stream bytes per point
  uint32_t min_regular_bytes;  // best case (decode sanity checks)
  uint32_t all_varint;         // every regular op is a varint/NaN token (decode can find token ends by MSB)
  uint32_t n_gorilla;          // OP_GORILLA64 ops
  uint32_t varint_and_raw;     // decode: every regular op is a varint token or a raw copy of 1/2/4/8 bytes, at least one of
                               // them raw, at most 8 ops and 256 bytes per point: k_mark_token_ends can lay out the token ends
  DevOp ops[kMaxOps];
  DevAdaptive adaptive[kMaxAdaptive];
};

// Plan of a schema beyond kMaxOps / kMaxAdaptive / kMaxPointStep: same entries, arrays in device memory (stage1_wide.h).
struct WidePlan {
  uint32_t point_step;
  uint32_t n_ops;
  uint32_t n_adaptive;
  uint32_t n_gorilla;
  uint32_t min_regular_bytes;
  uint32_t reserved;
  const DevOp* ops;             // device [n_ops]; OP_GORILLA64: DevOp::type is not used, the token buffer index is op_aux[k]
  const uint32_t* op_aux;       // device [n_ops]
  const DevAdaptive* adaptive;  // device [n_adaptive]
};

// One 32768-point chunk of one cloud of the batch.
struct ChunkDesc {
  uint64_t first_point;     // index into the batch's concatenated point array
  uint32_t n_points;        // 1..32768
  uint32_t cloud;           // cloud index in the batch
  uint32_t chunk_in_cloud;  // 0 = the chunk whose first 4096 points decide the adaptive modes
  uint32_t reserved;
};

// A piece of a chunk's payload, produced by one kernel into the chunk's slot; the compaction kernel
// concatenates a chunk's segments in index order behind the [u32 size] prefix.
struct Seg {
  uint32_t off;   // byte offset inside the chunk slot (multiple of 16)
  uint32_t size;  // bytes
};

struct PreTokenPtrs {
  const uint4* p[kMaxGorilla];  // per Gorilla op: {w0, w1, w2, len} for every point of the batch
};

struct ColumnPtrs {
  uint8_t* p[kMaxAdaptive];  // SoA copies of the adaptive-int fields: value i of the batch at p[a] + i*bpv
};

// Slot layout of one chunk (all offsets multiples of 256):
//   [0, reg_stride)                        regular stream
//   [reg_stride + a*sec_stride, ...)       section of adaptive field a; Palette uses two segments inside it
constexpr uint32_t kSectionStride = 409600;  // >= 5 + 32768*11 (DeltaRle worst case), and
                                             // >= align256(3 + 32768*8) + 32768*15/8 (Palette worst case)
constexpr uint32_t kPaletteIndexOffset = 262400;  // align256(3 + 32768*8)

// status word bits (device side, sticky)
enum : uint32_t {
  ST_OUT_OVERFLOW = 1u,    // compacted stream did not fit the output capacity
  ST_CORRUPT = 2u,         // decode: malformed input
  ST_PALETTE_FULL = 4u,    // internal: LDS palette table overflowed (handled by the global-table path)
  ST_FINISH_TIMEOUT = 8u   // internal: a workgroup of k_finish waited too long for a predecessor's chunk size
};

// ---- piece kernel (stage1_fused.h) ----
// One piece = the points one wave encodes: kPieceRows rows of 63 points of ONE chunk. Piece counts per chunk are
// padded to a multiple of 4 (one workgroup = 4 pieces of the same chunk); padding pieces have no points.
struct PieceDesc {
  uint64_t chunk_first_point;  // ChunkDesc::first_point of its chunk
  uint32_t chunk;              // index into the chunk table
  uint32_t cloud;              // ChunkDesc::cloud
  uint32_t n_chunk_points;     // ChunkDesc::n_points
  uint16_t p;                  // index of the piece inside its chunk
  uint16_t P;                  // pieces of the chunk (multiple of 4)
  uint32_t pad[2];
};
static_assert(sizeof(PieceDesc) == 32, "one aligned 32-byte load per piece");

}  // namespace cldn
