// stage1_launch.h -- host-callable launchers implemented next to the kernels (stage1_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "stage1_device.h"

namespace cldn {

struct EncodeLaunch {
  const DevPlan* plan;
  hipStream_t stream;
  const uint8_t* points;      // device, batch AoS
  const uint8_t* points_end;
  const ChunkDesc* chunks;    // device [n_chunks]
  uint32_t n_chunks;
  uint32_t n_clouds;
  const uint32_t* cloud_first_chunk;  // device [n_clouds + 1]
  uint8_t* slots;             // device [n_chunks * slot_stride]
  uint64_t slot_stride;
  uint64_t reg_stride;        // = subs * sub_stride
  uint32_t subs;              // sub-chunks (independent regular sub-streams) per chunk, power of two
  uint32_t sub_points;        // 32768 / subs
  uint32_t sub_stride;        // bytes reserved per sub-stream
  Seg* segs;                  // device [n_chunks * segs_per_chunk]
  uint32_t segs_per_chunk;
  ColumnPtrs cols;
  PreTokenPtrs pre;           // Gorilla token buffers (read side, by value into the kernels)
  uint4* const* pre_out;      // device array [n_gorilla] of the same buffers (write side of k_gorilla_tokens)
  uint16_t* ranks[kMaxAdaptive];
  uint32_t* chunk_payload;    // device [n_chunks]
  uint64_t* chunk_dst;        // device [n_chunks]
  uint64_t* stream_offsets;   // device [n_clouds + 1]
  uint8_t* modes;             // device [n_clouds * n_adaptive]
  bool modes_forced;          // modes were uploaded by the caller: no probe kernels
  // per adaptive field: bit m set = mode m may occur (modes of the previous call, or forced); 0xF = unknown.
  // Only a launch hint: a fast section kernel that is not launched leaves its chunks to k_encode_sections.
  uint8_t mode_hint[kMaxAdaptive];
  uint8_t* fallback_flags;    // device [n_chunks * n_adaptive], zeroed per call: 1 = section written by a fast path
  // chunk-table output: no framing -- the call ends with the chunks' payloads in their slots (segment table) and
  // k_chunk_sizes (payload sizes, contiguity flag)
  bool chunks_only;
  uint32_t* contiguous_flag;       // device word, 0 at launch: set to 1 when a chunk's payload is not one run of its slot
  // k_finish (stage1_finish.h)
  unsigned long long* fin_rec;     // device [n_chunks]: look-back records, tagged with fin_epoch
  unsigned long long* fin_rec2;    // device [n_chunks]
  unsigned long long* fin_anchor;  // device [n_chunks / 1024 + 1], zero at launch
  uint32_t fin_epoch;              // != 0, changes with every call
  uint32_t* fin_ticket;            // device, zero at launch
  uint32_t use_ticket;             // k_finish takes its workgroups' order from the ticket counter (retry after ST_FINISH_TIMEOUT)
  uint32_t test_timeout;           // test hook: see FinishArgs
  uint8_t* out;               // device, framed streams
  uint64_t out_capacity;
  uint32_t* status;           // device status word
  hipEvent_t* events;         // 5 events (start, before regular, after regular, after sections, end) or NULL
  // piece kernel (stage1_fused.h). pieces != NULL: the regular stream is encoded one wave per piece, every workgroup
  // leaves one segment in the chunk's slot
  const PieceDesc* pieces;    // device [n_pieces] or NULL
  uint32_t n_pieces;          // multiple of 4
  bool intra;                 // the piece kernel's workgroups place a chunk's regular stream contiguously (subs == 1)
  unsigned long long* wgrec;  // device [n_chunks * 32]: their look-back records, tagged with fin_epoch
  // WIDE route (stage1_wide.h): schemas beyond the launch-argument plan. wide != NULL: `plan` holds only the scalar members,
  // subs == 1 and segs_per_chunk == 1 (one segment per chunk), the slots take a chunk's whole payload.
  const WidePlan* wide;       // host copy of the descriptor (its arrays are device memory), or NULL
  const DevOp* wide_ops_host; // host copy of the regular ops (the Gorilla pre-pass is launched per group of them)
  uint8_t* wide_scratch;      // device [n_chunks * stage1_wide_scratch_bytes()]
  const uint4* const* wide_pre;  // device [n_gorilla] token buffers (same array as pre_out)
};
size_t stage1_wide_scratch_bytes();  // per chunk

uint32_t stage1_piece_points(const DevPlan& plan, const uint8_t* points);        // piece kernel applies: points per piece, else 0
uint32_t stage1_piece_slot_stride(const DevPlan& plan, const uint8_t* points);   // bytes a piece may produce, 256-aligned

struct DecodeLaunch {
  const DevPlan* plan;
  hipStream_t stream;
  uint32_t uses_v5;
  const uint8_t* streams;             // device: framed stage-1 streams of the batch
  const uint64_t* stream_offsets;     // device [n_clouds + 1]
  const uint64_t* cloud_first_point;  // device [n_clouds + 1]
  const uint32_t* cloud_first_chunk;  // device [n_clouds + 1]
  const uint64_t* h_stream_offsets;   // host copies of the three tables: calls of a few clouds pass them as a kernel argument
  const uint64_t* h_cloud_first_point;  // (then the device pointers above may be NULL)
  const uint32_t* h_cloud_first_chunk;
  uint32_t n_clouds;
  uint32_t n_chunks;
  uint32_t dv_hint;                   // what the codec's earlier calls saw: 1 = no chunk had a lone DeltaVarint section, 2 = every chunk had, 0 = unknown / mixed
  void* chunks;                       // device [n_chunks] DecChunk (48 bytes each)
  uint32_t* reg_end;                  // device [n_chunks]: end of the regular stream per chunk (fast path)
  uint8_t* sec_done;                  // device [n_chunks]: 1 = sections decoded by k_decode_sections
  uint8_t* cols[8];                   // device: dense columns of the first adaptive fields (n_points * bpv each), or NULL
  uint32_t* reg_end_pre;              // device [n_chunks]: k_decode_sections_cols: where the regular stream ends
  uint8_t* sec_cols;                  // device [n_chunks]: 1 = the columns hold the chunk's integer fields
  const uint32_t* chunk_sizes;        // device [n_chunks] or NULL: the payload sizes, if the caller knows them (no serial walk)
  uint32_t* token_ends;           // k_mark_token_ends: one bit per stream byte (+ a word per chunk), or NULL
  uint32_t fill_zero;             // CLDN_HIP_FILL_ZERO: bytes of a point that no field covers may be written as 0
  uint32_t* slices_done;          // [n_chunks] DeltaVarint slices of a chunk that k_sections_cols_fast finished
  unsigned long long* slice_rec;  // [n_chunks * 48 * 2] (count, sum) records of the slices, tagged with slice_epoch
  uint32_t slice_epoch;           // != 0, different from every earlier launch on slice_rec since it was cleared
  void* dsec;                     // [n_adaptive * n_chunks] DecChunk: the sections k_section_offsets sized (stage1_decode_sections_w.h), or NULL
  uint8_t* secs_ok;               // [n_chunks]
  uint32_t* done_cnt;             // [n_chunks]
  uint8_t* out;                       // device: decoded AoS points
  uint32_t* status;
  uint32_t palette_hint;              // the codec's last decode call folded every chunk's section as a small Palette found by
                                      // the point kernel's own guess: the kernels that locate sections and decode them into columns
                                      // are not launched (a chunk that is different after all goes to k_decode_tail's section decoders)
  hipEvent_t* events;                 // 4 events (start, before / behind the regular-stream kernel, end) or NULL
  // WIDE route: the serial decoder with the plan in device memory (schemas beyond the launch-argument plan)
  const WidePlan* wide;               // host copy of the descriptor, or NULL
  void* wide_state;                   // device [n_chunks * n_ops * 16]: the decoder's per-op state
  // SPLIT launches of the point kernel (small batches; stage1_decode_wave.h): NULL = never split
  void* wp_split;                     // device [wp_split_bytes(n_chunks, wp_maxp)]
  uint32_t wp_maxp;                   // pieces (992 bytes) a chunk's payload may have
  uint32_t wp_parts;                  // workgroups per chunk of the point decoder's launch (1 = chained; wp_split_parts or the test hook)
};
// bytes of the SPLIT workspace: per piece t0 (4) + aggregates (5 x 4) + carries (4 x 4), per chunk 4 flag words
inline size_t wp_split_bytes(uint32_t n_chunks, uint32_t maxp) { return (size_t)n_chunks * maxp * 40u + (size_t)n_chunks * 16u + 256u; }
// workgroups per chunk of a SPLIT launch (1 = the chained launch). Measured (device-resident decode, n x 1 M XYZI points /
// 130 k-point Velodyne clouds): a split launch costs about 1.5 x the arithmetic, three more launches and a prologue per
// workgroup -- it wins up to about 64 chunks (one cloud 0.093 -> 0.072 ms, one Velodyne cloud 0.153 -> 0.088) and loses
// from about 100 on (124 chunks 0.095 -> 0.111, 496 chunks 0.16 -> 0.39 ms)
inline uint32_t wp_split_parts(uint32_t n_chunks) {
  if (n_chunks == 0u || n_chunks > 64u) return 1u;
  const uint32_t parts = 256u / n_chunks;
  return parts > 16u ? 16u : (parts < 2u ? 2u : parts);
}

constexpr size_t kDecChunkBytes = 48;

// Launch errors go to the ABI's thread-local error string (cldn_hip_last_error), implemented in hip_abi.hip.
int launch_fail(hipError_t e, const char* what);

int stage1_configure_kernels();
int stage1_configure_decode();   // decode TU (stage1_decode.hip); called by stage1_configure_kernels
int stage1_launch_encode(const EncodeLaunch& L);
int stage1_launch_decode(const DecodeLaunch& L);
int stage1_launch_decode_unframed(const DevPlan& plan, hipStream_t stream, const uint8_t* payload, uint32_t size,
                                  uint32_t capacity_points, void* chunk_slot, uint8_t* out, uint32_t* status,
                                  const WidePlan* wide = nullptr, void* wide_state = nullptr);

// k_finish alone (no section work): frames the chunks of a batch -- one or more segments per chunk in per-chunk slots --
// as [u32 size][bytes] streams. Used for the chunks the device-side stage 2 leaves (lz4_kernels.hip).
struct FrameLaunch {
  hipStream_t stream;
  const ChunkDesc* chunks;
  uint32_t n_chunks;
  const uint32_t* cloud_first_chunk;
  uint32_t n_clouds;
  const uint8_t* slots;
  uint64_t slot_stride;
  const Seg* segs;
  uint32_t segs_per_chunk;
  unsigned long long* rec;     // [n_chunks], tagged with epoch
  unsigned long long* anchor;  // [n_chunks / 1024 + 1], zero at launch
  uint32_t epoch;
  uint32_t* ticket;            // zero at launch
  uint32_t use_ticket;
  uint32_t test_timeout;
  uint32_t* chunk_payload;     // out
  uint64_t* chunk_dst;         // out
  uint64_t* stream_offsets;    // out
  uint8_t* out;
  uint64_t out_capacity;
  uint32_t* status;
};
int stage1_launch_frame(const FrameLaunch& F);

// ---- stage 2 on the device: LZ4 block per chunk (lz4_kernels.hip; parameters shared with oracle/lz4_model.c) ----
constexpr uint32_t kLzSubBytes = 8192;    // a wave parses this much of a payload with its own hash table
constexpr uint32_t kLzHashBits = 11;
constexpr uint32_t kLzMaxMatches = 1024;  // per sub-range (the rest of it leaves as literals)
// CLDN_HIP_STAGE2_LZ4_FAST (round 5): sub-ranges of 4 KiB with a 1024-entry table and 512 matches -- 8.2 KB of LDS per wave
// instead of 16.4, twice the resident waves (round 4 measured 1.53 x the speed for 3 % of the ratio: 0.894 -> 0.922)
constexpr uint32_t kLzFastSubBytes = 4096;
constexpr uint32_t kLzFastHashBits = 10;
constexpr uint32_t kLzFastMaxMatches = 512;
struct LzMatch {
  uint32_t pos;  // in the chunk's payload
  uint16_t len;  // 4 .. kLzSubBytes
  uint16_t off;  // 1 .. kLzSubBytes - 1
};
struct Lz4Launch {
  hipStream_t stream;
  const uint8_t* stage1;          // framed stage-1 streams (the payload of chunk c starts at chunk_dst[c] + 4)
  const uint64_t* chunk_dst;
  const uint32_t* chunk_payload;
  uint32_t n_chunks;
  uint32_t fast;                  // 1 = the CLDN_HIP_STAGE2_LZ4_FAST parameters
  uint64_t max_subs;              // upper bound of the sub-ranges of the batch (payload bound / sub-range bytes + n_chunks)
  uint32_t* sub_first;            // [n_chunks + 1]: compact sub-range numbering
  LzMatch* matches;               // [max_subs * kLzMaxMatches] (fast: kLzFastMaxMatches)
  uint32_t* counts;               // [max_subs] and the arrays behind it: last_end, anchor_in, sub_size, sub_chunk, before, next_pos
  uint32_t* last_end;
  uint32_t* anchor_in;
  uint32_t* sub_size;
  uint32_t* sub_chunk;            // [max_subs]: chunk of a sub-range
  uint32_t* before;               // [max_subs]: sequence bytes of the chunk's earlier sub-ranges
  uint32_t* next_pos;             // [max_subs]: the chunk's next match behind the sub-range
  uint32_t* block_size;           // [n_chunks]: bytes of the chunk's LZ4 block
  // the blocks are written straight into the framed streams (round 5: no slots, no second framing pass)
  const uint32_t* cloud_first_chunk;  // [n_clouds + 1]
  uint32_t n_clouds;
  uint64_t* block_dst;            // [n_chunks]: where [u32 size][block] of a chunk begins in `out`
  uint32_t* block_sizes_out;      // [n_chunks]: what chunk_sizes reports
  uint64_t* stream_offsets;       // [n_clouds + 1]
  uint8_t* out;
  uint64_t out_capacity;
  uint32_t* status;
};
int lz4_launch(const Lz4Launch& L);

// applyVizLossyPreprocessing (viz_kernels.hip)
struct VizLaunch {
  hipStream_t stream;
  const uint8_t* points;          // device AoS
  uint64_t n_points;              // < 2^32
  uint32_t point_step;
  uint32_t xyz_offset;
  float inv_res;
  unsigned long long* keys;       // device [viz_table_capacity(n_points)]
  uint32_t* first;                // device [capacity]
  uint32_t* slot_of;              // device [n_points]
  uint32_t* block_count;          // device [ceil(n_points / 1024)]
  unsigned long long* total;      // device: surviving points
  uint8_t* out;                   // device [n_points * point_step]
};
uint64_t viz_table_capacity(uint64_t n_points);
int viz_launch(const VizLaunch& L);

}  // namespace cldn
