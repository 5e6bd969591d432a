/* cloudini_amd_c.h -- flat C access to the C++ host API (Cloudini::PointcloudEncoder / PointcloudDecoder / header
 * functions / ROS message converters) for language bindings and for the Python test-suite. Every function returns
 * a byte count (>= 0) or -1 with the std::runtime_error text available from cldn_amd_last_error(). */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cldn_amd_field {
  const char* name;
  uint32_t offset;
  uint8_t type;           /* Cloudini::FieldType */
  uint8_t has_resolution;
  uint8_t reserved[2];
  float resolution;
} cldn_amd_field_t;

/* Cloudini::EncodingInfo without its std::vector */
typedef struct cldn_amd_info {
  const cldn_amd_field_t* fields;
  uint32_t n_fields;
  uint32_t width;
  uint32_t height;
  uint32_t point_step;
  uint8_t encoding_opt;
  uint8_t compression_opt;
  uint8_t version;
  uint8_t use_threads;
} cldn_amd_info_t;

const char* cldn_amd_last_error(void);

/* MaxCompressedSize(info, n_points, include_header) */
int64_t cldn_amd_max_compressed_size(const cldn_amd_info_t* info, uint64_t n_points, int include_header);
/* EncodeHeader(info, out, binary ? BINARY : YAML) */
int64_t cldn_amd_encode_header(const cldn_amd_info_t* info, int binary, uint8_t* out, uint64_t capacity);
/* PointcloudEncoder(info).encode(data, out, write_header) */
int64_t cldn_amd_encode(const cldn_amd_info_t* info, const uint8_t* data, uint64_t size, uint8_t* out,
                        uint64_t capacity, int write_header);
/* DecodeHeader + PointcloudDecoder::decode of a full stream; yaml_out (optional) receives EncodingInfoToYAML of the
 * decoded header, version_out (optional) the wire version */
int64_t cldn_amd_decode(const uint8_t* stream, uint64_t size, uint8_t* out, uint64_t capacity, char* yaml_out,
                        uint64_t yaml_capacity, uint8_t* version_out);
/* PointcloudDecoder::decode(info, data (no header), out) */
int64_t cldn_amd_decode_noheader(const cldn_amd_info_t* info, const uint8_t* data, uint64_t size, uint8_t* out,
                                 uint64_t capacity);
/* the same for an output buffer that is all zeros on entry (PointcloudDecoder::decodeInto(..., true): what
 * decode(info, data, std::vector&) does for a vector that arrives empty) */
int64_t cldn_amd_decode_noheader_zeroed(const cldn_amd_info_t* info, const uint8_t* data, uint64_t size, uint8_t* out,
                                        uint64_t capacity);
/* getDeserializedPointCloudMessage + applyResolutionProfile({}, fields, resolution) + toEncodingInfo (compression
 * as given) + convertPointCloud2ToCompressedCloud */
int64_t cldn_amd_ros_compress(const uint8_t* dds, uint64_t size, float resolution, uint8_t compression_opt,
                              uint8_t* out, uint64_t capacity);
/* getDeserializedPointCloudMessage + convertCompressedCloudToPointCloud2 */
int64_t cldn_amd_ros_decompress(const uint8_t* dds, uint64_t size, uint8_t* out, uint64_t capacity);

/* cloudini_ros::applyVizLossyPreprocessing on a bare point buffer (fields / point_step of `info`; width, height are
 * derived from size): out receives the surviving points (capacity >= size), res_out[i] the resolution of field i
 * afterwards (NaN = none), *width_out / *height_out the new shape. Returns the surviving byte count. */
int64_t cldn_amd_viz_preprocess(const cldn_amd_info_t* info, const uint8_t* data, uint64_t size, uint8_t* out,
                                uint64_t capacity, float* res_out, uint32_t* width_out, uint32_t* height_out);

/* Batch transcoder (include/cloudini_amd/batch_transcoder.hpp): every file of in_dir (one CDR sensor_msgs/PointCloud2
 * per file, lexicographic order) becomes a CompressedPointCloud2 file of the same name in out_dir, byte-identical to
 * what cloudini_ros::convertPointCloud2ToCompressedCloud writes for it with resolution `resolution` for the FLOAT32
 * fields and stage 2 `compression_opt`. stats_out (optional, 8 doubles): messages, points, input bytes, output bytes,
 * GPU batches, seconds total, seconds in GPU calls, seconds in stage 2 + wrapping. Returns the message count or -1. */
int64_t cldn_amd_transcode_directory(const char* in_dir, const char* out_dir, float resolution, uint8_t compression_opt,
                                     int viz_lossy, uint32_t batch_messages, double* stats_out);

/* The way back (McapConverter::decodePointClouds, cloudini_lib/tools/src/mcap_converter.cpp:240-300): every file of
 * in_dir is a CDR CompressedPointCloud2, the file of the same name in out_dir the sensor_msgs/PointCloud2 that
 * cloudini_ros::convertCompressedCloudToPointCloud2 writes for it -- stage 2 undone on the host pool, one batched GPU
 * decode per run of messages with the same schema. stats_out as above. Returns the message count or -1. */
int64_t cldn_amd_decode_directory(const char* in_dir, const char* out_dir, uint32_t batch_messages, double* stats_out);

/* The same two with a device list (cloudini_amd::TranscodeOptions::devices): the batches are spread over one GPU stage
 * per entry (a device may be listed twice), the output order and bytes are those of the single-device run.
 * devices == NULL: the calling thread's current device. */
int64_t cldn_amd_transcode_directory_on(const char* in_dir, const char* out_dir, float resolution, uint8_t compression_opt,
                                        int viz_lossy, uint32_t batch_messages, const int32_t* devices, uint32_t n_devices,
                                        double* stats_out);
int64_t cldn_amd_decode_directory_on(const char* in_dir, const char* out_dir, uint32_t batch_messages, const int32_t* devices,
                                     uint32_t n_devices, double* stats_out);

/* Stage-2 (LZ4 / ZSTD) threads a single encode()/decode() call with use_threads may occupy, the caller included.
 * The reference's flag means one extra worker (cloudini_lib/src/cloudini.cpp:453-499); here the pool is bounded:
 * default min(4, hardware threads), overridden by the environment variable CLOUDINI_AMD_STAGE2_THREADS (read once)
 * or by this setter (1 = the calling thread only). Returns the value in effect. */
uint32_t cldn_amd_stage2_threads(void);
/* Stage 2 of LZ4 streams on the GPU (include/cloudini_hip.h, cldn_hip_codec_set_stage2): PointcloudEncoder::encode with
 * compression_opt == LZ4 then receives [u32 size][LZ4 block] per chunk from the device -- valid LZ4 blocks that the
 * reference's decoder reads, not the bytes lz4's own compressor writes. Off by default; the environment variable
 * CLOUDINI_AMD_DEVICE_LZ4=1 (read once) or this setter turns it on; 2 selects CLDN_HIP_STAGE2_LZ4_FAST (4 KiB windows: about 1.7 x
 * the speed, blocks about 3 % larger). Both return the level in effect (0, 1, 2). */
int cldn_amd_device_lz4(void);
int cldn_amd_set_device_lz4(int on);
uint32_t cldn_amd_set_stage2_threads(uint32_t n);

#ifdef __cplusplus
}
#endif
