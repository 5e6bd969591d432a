/* cloudini_hip.h -- C ABI of the MI355X (gfx950) stage-1 codec: the drop-in boundary.
 *
 * What it replaces in the reference (paths under /root/reference/cloudini_lib):
 *   detail::EncodeV5Stage1(...)        src/v5_codec.hpp:31-34, called from src/cloudini.cpp:590-599
 *   detail::EncodeV4Stage1Chunk(...)   src/v4_codec.hpp:33-35, called from src/cloudini.cpp:608-614
 *   detail::DecodeV5Stage1Chunk(...)   src/v5_codec.hpp:40-42, called from src/cloudini.cpp:677-679
 *   detail::DecodeV4Stage1Chunk(...)   src/v4_codec.hpp:37-40, called from src/cloudini.cpp:680-683
 * i.e. everything between "a contiguous AoS point buffer + EncodingInfo" and "stage-1 bytes per
 * 32768-point chunk", in both directions. Header (YAML), [u32 size] framing of *compressed* chunks and
 * LZ4/ZSTD stay on the host (cloudini_amd/csrc/host/).
 *
 * Conventions follow the reference's existing C ABI (include/cloudini_lib/wasm_functions.h:30-93):
 * plain pointers and sizes, caller-allocated outputs, no exceptions across the boundary. Errors are
 * negative return codes plus cldn_hip_last_error() (thread-local string).
 *
 * Wire format produced/consumed: the *framed stage-1 stream* of one cloud,
 *     [u32 LE payload_size][payload] per chunk of <= 32768 points,
 * byte-identical to what PointcloudEncoder::encode writes after its header when
 * compression_opt == NONE (src/chunk_writer.cpp:32-39). For LZ4/ZSTD the host compresses each payload.
 *
 * Memory: every data pointer is tagged CLDN_HIP_HOST or CLDN_HIP_DEVICE. With DEVICE pointers a call only
 * enqueues work on the codec's HIP stream (no synchronisation, outputs valid after the stream drains);
 * with HOST outputs the call returns after the results are in host memory.
 */
#ifndef CLOUDINI_HIP_H
#define CLOUDINI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLDN_HIP_ABI_VERSION 1
#define CLDN_HIP_POINTS_PER_CHUNK 32768u /* detail::kPointsPerChunk, src/codec_common.hpp:28 */
#define CLDN_HIP_PROBE_POINTS 4096u      /* kAdaptiveModeProbePoints, src/v5_codec.cpp:76 */

/* Schemas: every schema the reference's factory accepts (CreateCompatibleEncoder, src/codec_common.cpp:116-153;
 * buildV5Plan, src/v5_codec.cpp:719-740) is accepted here -- no limit on fields, tokens or point_step. Schemas of at
 * most 64 per-point tokens (a fused FloatN group counts 3 or 4), 64 adaptive integer fields and 1024-byte points take the
 * ordinary kernels (their plan is a launch argument); anything larger takes the WIDE route (csrc/stage1_wide.h: plan in
 * device memory, one workgroup per chunk on the way in, the serial decoder on the way back), byte-exact and slow. */

/* Return codes. */
enum {
  CLDN_HIP_OK = 0,
  CLDN_HIP_ERR_ARG = -1,          /* invalid argument / schema */
  CLDN_HIP_ERR_CAPACITY = -2,     /* output buffer smaller than the worst-case bound (cloudini.cpp:531-534) */
  CLDN_HIP_ERR_UNSUPPORTED = -3,  /* a call this build cannot serve (e.g. more than 2^32 - 2 points for the viz pre-filter); no schema is refused */
  CLDN_HIP_ERR_DEVICE = -4,       /* HIP runtime error */
  CLDN_HIP_ERR_NO_DEVICE = -5,    /* no usable GPU */
  CLDN_HIP_ERR_CORRUPT = -6,      /* decode: malformed stream (truncated, bad mode byte, trailing bytes, ...) */
  CLDN_HIP_ERR_NOMEM = -7
};

enum { CLDN_HIP_HOST = 0, CLDN_HIP_DEVICE = 1 };

/* One Cloudini::PointField (include/cloudini_lib/basic_types.hpp:47-67) without its name. `type` is a
 * Cloudini::FieldType value (1..10, == sensor_msgs/PointField datatype for 1..8). */
typedef struct cldn_hip_field {
  uint32_t offset;
  uint8_t type;
  uint8_t has_resolution;
  uint8_t reserved[2];
  float resolution;
} cldn_hip_field_t;

typedef struct cldn_hip_plan cldn_hip_plan_t;   /* immutable: schema -> regular ops + adaptive-int fields */
typedef struct cldn_hip_codec cldn_hip_codec_t; /* execution context: device, stream, workspace. One call at
                                                   a time per codec (like a PointcloudEncoder instance). */

const char* cldn_hip_last_error(void);
int cldn_hip_abi_version(void);
int cldn_hip_device_count(void); /* >= 0, or a negative error */
int cldn_hip_current_device(void); /* the calling thread's current HIP device (>= 0), or a negative error */
int cldn_hip_set_current_device(int device); /* hipSetDevice for the calling thread (a worker thread per GPU) */

/* Page-locked host memory for buffers handed to the HOST-tagged entry points: copies from / to it run at PCIe speed and
 * asynchronously, pageable memory goes through the driver's bounce buffers (measured 1.4 GB/s against 25 GB/s for the
 * messages of a bag). NULL on failure. */
void* cldn_hip_host_alloc(size_t bytes);
void cldn_hip_host_free(void* p);

/* Plan = the encoder/decoder selection of BuildV4Encoders (src/v4_codec.cpp:26-40), buildV5Plan
 * (src/v5_codec.cpp:719-740) and CreateCompatibleEncoder (src/codec_common.cpp:116-153) for
 * EncodingInfo{fields, point_step, version, encoding_opt}. encoding_opt: 0 NONE, 1 LOSSY, 2 LOSSLESS. */
int cldn_hip_plan_create(const cldn_hip_field_t* fields, uint32_t n_fields, uint32_t point_step,
                         uint8_t version, uint8_t encoding_opt, cldn_hip_plan_t** out);
void cldn_hip_plan_destroy(cldn_hip_plan_t* plan);
int cldn_hip_plan_uses_v5(const cldn_hip_plan_t* plan);                 /* detail::UsesV5Codec */
uint32_t cldn_hip_plan_adaptive_fields(const cldn_hip_plan_t* plan);    /* number of V5 adaptive-int fields */
uint32_t cldn_hip_plan_max_point_bytes(const cldn_hip_plan_t* plan);    /* detail::MaxSerializedPointSize */
/* MaxCompressedSize(info, n_points, include_header=false) for CompressionOption::NONE
 * (src/cloudini.cpp:249-292): the capacity the framed stage-1 stream of one cloud must be given. */
uint64_t cldn_hip_stage1_bound(const cldn_hip_plan_t* plan, uint64_t n_points);

/* device < 0: current device. hip_stream: a hipStream_t (NULL = the codec creates its own stream).
 * A codec works on its own device whatever the caller's current device is; every entry point restores the calling
 * thread's current device before it returns. */
int cldn_hip_codec_create(const cldn_hip_plan_t* plan, int device, void* hip_stream, cldn_hip_codec_t** out);
void cldn_hip_codec_destroy(cldn_hip_codec_t* codec);
int cldn_hip_codec_synchronize(cldn_hip_codec_t* codec);
void* cldn_hip_codec_stream(cldn_hip_codec_t* codec);
int cldn_hip_codec_device(const cldn_hip_codec_t* codec); /* device the codec was created on */

/* Encode a batch of clouds that share the plan's schema.
 *   points         n_total_points * point_step bytes, clouds back to back (cloud k has cloud_points[k] points)
 *   cloud_points   HOST array [n_clouds]
 *   out            receives the framed stage-1 streams of all clouds, back to back (compact)
 *   out_capacity   must be >= sum_k cldn_hip_stage1_bound(plan, cloud_points[k])
 *   stream_offsets [n_clouds + 1] byte offsets of each cloud's stream inside `out` (last = total size)
 *   chunk_sizes    [total chunks] payload size of every chunk, batch order (optional, may be NULL)
 *   modes          [n_clouds * adaptive_fields] committed V5 adaptive-int mode per cloud and field
 *                  (0 DeltaVarint, 1 Palette, 2 Rle, 3 DeltaRle; src/v5_codec.cpp:33-38) (optional)
 * stream_offsets / chunk_sizes / modes live where `out` lives (out_loc).
 * Alignment: `points` and `out` may have ANY byte alignment, host or device (a stream often follows a header of odd length;
 * tests/test_gpu_encode.py::test_device_buffers_at_any_address). Device-resident stream_offsets (8-byte aligned) and chunk_sizes
 * (4-byte aligned) are written by the kernels in place; at other alignments they are filled by a copy behind the kernels. */
int cldn_hip_encode_stage1(cldn_hip_codec_t* codec, const void* points, int points_loc,
                           const uint64_t* cloud_points, uint32_t n_clouds, void* out, uint64_t out_capacity,
                           int out_loc, uint64_t* stream_offsets, uint32_t* chunk_sizes, uint8_t* modes);

/* Chunk-table output: stage 1 WITHOUT the framing. The reference's own boundary between stage 1 and stage 2 is a buffer
 * per chunk (EncodeV5Stage1 / EncodeV4Stage1Chunk write one, WriteStage1Chunk compresses or copies it into the stream:
 * src/cloudini.cpp:590-614, src/chunk_writer.cpp:27-48); this call stops there. The payload of chunk c (batch order) is
 * the concatenation of its non-empty segments: segments[c * segments_per_chunk + k] = {offset inside the chunk's slot
 * payload_base + c * chunk_stride, size}; chunk_sizes[c] = their sum. Schemas with at most one adaptive integer field
 * on the piece-kernel path (XYZ, XYZI, XYZ + rgba, XYZI + ring ...) get every payload as ONE run of its slot
 * (*not_contiguous stays 0): nothing is moved a second time, which is what a device-side stage 2 or any other consumer
 * on the GPU wants. All pointers are DEVICE pointers into the codec's workspace: valid until the codec's next call, to
 * be read behind the codec's stream. modes_device: DEVICE array [n_clouds * adaptive_fields] or NULL.
 * cldn_hip_frame_chunks turns the table of the codec's last cldn_hip_encode_stage1_chunks call into the framed streams
 * (one straight copy per chunk) exactly as cldn_hip_encode_stage1 would have written them. */
typedef struct cldn_hip_segment {
  uint32_t offset;
  uint32_t size;
} cldn_hip_segment_t;
typedef struct cldn_hip_chunk_table {
  const uint8_t* payload_base;
  uint64_t chunk_stride;
  const cldn_hip_segment_t* segments;
  uint32_t segments_per_chunk;
  uint32_t n_chunks;
  const uint32_t* chunk_sizes;
  const uint32_t* not_contiguous; /* one word: 0 = every chunk's payload is one run, starting at its first non-empty segment */
} cldn_hip_chunk_table_t;
int cldn_hip_encode_stage1_chunks(cldn_hip_codec_t* codec, const void* points, int points_loc,
                                  const uint64_t* cloud_points, uint32_t n_clouds, cldn_hip_chunk_table_t* table,
                                  uint8_t* modes_device);
int cldn_hip_frame_chunks(cldn_hip_codec_t* codec, void* out, uint64_t out_capacity, int out_loc,
                          uint64_t* stream_offsets, uint32_t* chunk_sizes);

/* Two-step host output, for callers that do not want to provide the worst-case bound in host memory (50 bytes per point
 * for XYZI+ring against 8 produced): cldn_hip_encode_stage1 / _gather with out == NULL and out_loc == CLDN_HIP_HOST
 * encode into the codec's own device buffer and return the sizes (stream_offsets, chunk_sizes, modes in host memory);
 * cldn_hip_codec_fetch_output then copies the stream_offsets[n_clouds] bytes that were produced. */
int cldn_hip_codec_fetch_output(cldn_hip_codec_t* codec, void* out, uint64_t out_capacity);

/* The same for a batch whose clouds sit in SEPARATE host buffers (the messages of a bag): cloud k is read from
 * cloud_ptrs[k] (HOST array of HOST pointers, cloud_points[k] * point_step bytes each) and copied straight to its place
 * in the device batch -- no gathering copy on the host. Everything else as in cldn_hip_encode_stage1. */
int cldn_hip_encode_stage1_gather(cldn_hip_codec_t* codec, const void* const* cloud_ptrs, const uint64_t* cloud_points,
                                  uint32_t n_clouds, void* out, uint64_t out_capacity, int out_loc,
                                  uint64_t* stream_offsets, uint32_t* chunk_sizes, uint8_t* modes);

/* Stage 2 on the device (what it replaces: CompressChunk with LZ4_compress_default,
 * src/codec_common.cpp:220-258, fed by WriteStage1Chunk, src/chunk_writer.cpp:27-48). With CLDN_HIP_STAGE2_LZ4 the encode
 * calls of this codec write, for every chunk, [u32 LE block size][LZ4 block of the chunk's stage-1 payload] -- the bytes
 * a stream with compression_opt == LZ4 holds behind its header; chunk_sizes then reports the block sizes and the
 * capacity `out` must offer is sum_k cldn_hip_stage2_bound(plan, cloud_points[k], CLDN_HIP_STAGE2_LZ4)
 * (= MaxCompressedSize for LZ4, src/cloudini.cpp:249-292). The blocks are valid LZ4 blocks that LZ4_decompress_safe
 * (the reference's DecompressChunk, src/codec_common.cpp:260-299) turns back into the exact stage-1 payloads; they are
 * NOT the bytes lz4's own compressor would write (a different, data-parallel parse: cloudini_amd/csrc/lz4_kernels.hip,
 * restated serially in oracle/lz4_model.c). The setting stays until changed. Default: CLDN_HIP_STAGE2_NONE. */
enum { CLDN_HIP_STAGE2_NONE = 0, CLDN_HIP_STAGE2_LZ4 = 1,
       CLDN_HIP_STAGE2_LZ4_FAST = 2 /* round 5: the same parser on 4 KiB sub-ranges (1024-entry table): twice the resident
                                       waves, ~1.5 x the speed, blocks ~3 % larger; same bound, same decoder */ };
int cldn_hip_codec_set_stage2(cldn_hip_codec_t* codec, int stage2);
uint64_t cldn_hip_stage2_bound(const cldn_hip_plan_t* plan, uint64_t n_points, int stage2);

/* Continuation of one cloud across several calls / devices. The reference commits the adaptive-int modes once per
 * encode() call, on the first <= 4096 points of the cloud (src/v5_codec.cpp:934-949), and resets every other
 * state at each 32768-point chunk (:910-915). A range of whole chunks of a cloud can therefore be encoded on
 * its own -- on another GPU -- once the modes are known: pass the `modes` that the cloud's first chunk produced
 * (cldn_hip_encode_stage1 of its first min(n, 4096) points) and encode the range as if it were a cloud; the
 * framed chunks are byte-identical to that range of the whole cloud's stream.
 *   modes   HOST array [adaptive_fields] with values 0..3; NULL / n_modes = 0 returns to probing.
 * The setting stays until changed and applies to every cloud of the following encode calls. */
int cldn_hip_codec_force_modes(cldn_hip_codec_t* codec, const uint8_t* modes, uint32_t n_modes);

/* Encoder pipelines (both produce identical bytes; for A/B runs and tests):
 *   1  tile kernel + slots   every schema: workgroup tiles with barriers (k_encode_floatn / k_encode_regular)
 *   2  piece kernel + slots  schemas whose per-point stream is one fused FloatN encoder (3 or 4 leading lossy FLOAT32
 *                            fields), optionally followed by one more per-point encoder: one wave per 504/378-point
 *                            piece, barrier-free (cloudini_amd/csrc/stage1_fused.h)
 *   0  automatic (default)   2 where the schema allows it, else 1
 * Either way the streams are left in per-chunk slots and k_finish (sections, chunk sizes, placement) packs them.
 * Returns the pipeline the next encode call of this codec takes for inputs at `points` (device pointer, or NULL for
 * host inputs), or a negative error. */
int cldn_hip_codec_pipeline(cldn_hip_codec_t* codec, int mode, const void* points);

/* Decode a batch of framed stage-1 streams (inverse of the above).
 *   streams        the streams back to back; cloud k occupies [stream_offsets[k], stream_offsets[k+1])
 *   stream_offsets HOST array [n_clouds + 1]
 *   cloud_points   HOST array [n_clouds] (width*height of each cloud)
 *   points_out     sum_k cloud_points[k] * point_step bytes; bytes not covered by a field keep their
 *                  previous content (src/field_decoder.cpp:72-76)
 * `streams` and `points_out` may have any byte alignment, host or device.
 * Returns CLDN_HIP_ERR_CORRUPT for malformed input when out_loc == HOST; with DEVICE outputs the status is
 * reported by the next cldn_hip_codec_status() call. */
int cldn_hip_decode_stage1(cldn_hip_codec_t* codec, const void* streams, int streams_loc,
                           const uint64_t* stream_offsets, const uint64_t* cloud_points, uint32_t n_clouds,
                           void* points_out, uint64_t out_capacity, int out_loc);

/* The same with the payload sizes of the chunks given (chunk_sizes: [total chunks], batch order, HOST or DEVICE per
 * chunk_sizes_loc; e.g. what cldn_hip_encode_stage1 reported, or the [u32] prefixes a host caller has read anyway): the
 * chunk table is then built in parallel instead of following the prefixes one dependent read after the other (one
 * 10 M-point cloud has 306 of them). Every size is still checked against its prefix; a mismatch is
 * CLDN_HIP_ERR_CORRUPT. chunk_sizes == NULL: exactly cldn_hip_decode_stage1. */
int cldn_hip_decode_stage1_sized(cldn_hip_codec_t* codec, const void* streams, int streams_loc,
                                 const uint64_t* stream_offsets, const uint64_t* cloud_points, uint32_t n_clouds,
                                 const uint32_t* chunk_sizes, int chunk_sizes_loc, void* points_out, uint64_t out_capacity,
                                 int out_loc);

/* Wire version 2 (streams written before the chunked format; the reference still reads them, src/cloudini.cpp:665-667):
 * the whole stage-1 payload is ONE unframed run of points without [u32 size] prefixes and without state resets, decoded
 * until it is empty (DecodeV4Stage1Chunk with expected_points = 0, src/v4_codec.cpp:108-115). The output capacity bounds
 * the point count ("Output buffer is too small to hold the decoded data" = CLDN_HIP_ERR_CORRUPT here); points behind the
 * last decoded one keep their content. One lane decodes: this is a compatibility path, not a fast one. */
int cldn_hip_decode_stage1_unframed(cldn_hip_codec_t* codec, const void* payload, uint64_t size, int payload_loc,
                                    void* points_out, uint64_t out_capacity, int out_loc);

/* cloudini_ros::applyVizLossyPreprocessing, data path (include/cloudini_lib/ros_msg_utils.hpp:175-221,
 * src/ros_msg_utils.cpp:249-341) -- the step right in front of the encoder in the rosbag converter
 * (tools/src/mcap_converter.cpp:195-197): points with a non-finite x, y or z (three float32 at xyz_offset, +4, +8) are
 * dropped; of the points that fall into the same voxel (lround(v * (1.0f / resolution)) per axis) only the first
 * one survives; survivors keep their order and all their bytes. `out` must hold n_points * point_step bytes;
 * *kept_points (HOST) receives the survivor count. The call synchronises the codec's stream. The codec only lends
 * its device, stream and workspace: its plan is not used. */
int cldn_hip_viz_preprocess(cldn_hip_codec_t* codec, const void* points, int points_loc, uint64_t n_points,
                            uint32_t point_step, uint32_t xyz_offset, float resolution, void* out,
                            uint64_t out_capacity, int out_loc, uint64_t* kept_points);

/* What a decode call may do to the bytes of a point that no field covers. CLDN_HIP_FILL_KEEP (default): they keep the
 * content of points_out (src/field_decoder.cpp:72-76 writes fields only) -- for a HOST buffer of a layout with such bytes
 * that means bringing the buffer to the device first. CLDN_HIP_FILL_ZERO: the caller hands over a buffer whose content
 * it does not need (a freshly resized vector, like PointcloudDecoder::decode(info, data, std::vector&) of the reference
 * on an empty vector): those bytes read 0 afterwards in a HOST buffer, and are 0 or untouched in a DEVICE buffer. Besides
 * the saved upload, the two common padded layouts (XYZ f32 + a 16-bit field in 16-byte points, XYZ f32 + a 32-bit field at
 * offset 16 in 32-byte points) then leave the decoder as whole 16-byte stores: 15-18 % faster than stores around the
 * padding (32 x 1 M XYZI: 0.33 -> 0.28 ms). */
#define CLDN_HIP_FILL_KEEP 0
#define CLDN_HIP_FILL_ZERO 1
int cldn_hip_codec_set_decode_fill(cldn_hip_codec_t* codec, int fill);

/* Which kernels the last cldn_hip_decode_stage1 call used, in chunks (synchronises):
 *   stats[0] regular stream by the parallel decoder      stats[1] V5 sections by the parallel decoder
 *   stats[2] whole chunks by the serial decoder          stats[3] only the sections by the serial decoder */
int cldn_hip_codec_decode_stats(cldn_hip_codec_t* codec, uint32_t stats[4]);

/* Synchronise and return the status word of the last asynchronous call (0 or a negative error). */
int cldn_hip_codec_status(cldn_hip_codec_t* codec);

/* Forward progress of the framing kernel (k_finish). Its workgroups wait for the size records of the workgroups of LOWER
 * index, which the hardware has started earlier when it hands a grid out in index order -- an observation about gfx950, not
 * a guarantee of the programming model (another device, a shared or pre-empted GPU). The wait is bounded (~1 s); when it
 * runs out the launch reports ST_FINISH_TIMEOUT. A call with HOST outputs is then redone ONCE with the order taken from a
 * ticket counter (an atomic per workgroup, 11 us per 1000 chunks: independent of the dispatch order), and the codec keeps
 * the ticket order from then on; a call with DEVICE outputs cannot be redone by the library: cldn_hip_codec_status returns
 * CLDN_HIP_ERR_DEVICE, the codec switches to tickets, the caller repeats the call. The ticket order is not the default
 * because it is not free: measured on the 32 x 1 M-point batch (round 6, same box, profiles/r06_d_finish_ticket.txt) k_finish
 * takes 0.137 instead of 0.127 ms with it, the framed step 0.292 instead of 0.282 ms. (The development build, -DCLDN_DEV,
 * selects it from the start with CLDN_HIP_FINISH_TICKET=1; the shipped library reads no such variable.) Returns the number of
 * calls this codec has redone. */
uint32_t cldn_hip_codec_finish_retries(const cldn_hip_codec_t* codec);

/* Optional instrumentation for bench.py's roofline line: with n_slots > 0 every encode call records HIP
 * events on the codec's stream around its kernels into slot (call_index % n_slots); n_slots = 0 turns it off.
 * cldn_hip_codec_kernel_ms waits for that slot's last event and returns milliseconds:
 *   ms[0] k_encode_regular, ms[1] section kernels (probe + sections), ms[2] offsets + compaction, ms[3] all. */
int cldn_hip_codec_enable_timing(cldn_hip_codec_t* codec, uint32_t n_slots);
int cldn_hip_codec_kernel_ms(cldn_hip_codec_t* codec, uint32_t slot, float ms[4]);
/* The same for decode calls (round 5): with timing enabled, the LAST cldn_hip_decode_stage1* call's events:
 *   ms[0] the kernel that decodes the regular streams (k_decode_points_w / k_decode_stream_w / k_decode_fixed; 0 when the
 *   call took another route), ms[1] everything the call launched. */
int cldn_hip_codec_decode_ms(cldn_hip_codec_t* codec, float ms[2]);

#ifdef __cplusplus
}
#endif
#endif /* CLOUDINI_HIP_H */
