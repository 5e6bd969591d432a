// Cloudini public API, MI355X edition: source-compatible with the reference's cloudini_lib/cloudini.hpp
// (EncodingInfo, header functions, MaxCompressedSize, PointcloudEncoder, PointcloudDecoder), but stage 1 --
// the per-point quantise / delta / varint work and the V5 integer sections -- runs in hand-written HIP kernels
// behind the C ABI of include/cloudini_hip.h. Header, chunk framing and LZ4/ZSTD (stage 2) stay on the host.
//
// Differences a caller can observe:
//   * a GPU is required; every error (including "no device") is a std::runtime_error like the reference's own;
//   * every schema the reference accepts is accepted (round 5: very wide ones -- more than 64 per-point tokens or adaptive
//     integer fields, points beyond 1024 bytes -- run on a slow route of their own, see cloudini_hip.h);
//   * the classes hold an opaque implementation pointer instead of the reference's private members.
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "cloudini_lib/encoding_utils.hpp"

namespace Cloudini {

// stage 1: field-aware encoding
enum class EncodingOptions : uint8_t { NONE = 0, LOSSY = 1, LOSSLESS = 2 };
// stage 2: general purpose compression of each chunk
enum class CompressionOption : uint8_t { NONE = 0, LZ4 = 1, ZSTD = 2 };

const char* ToString(const EncodingOptions& opt);
const char* ToString(const CompressionOption& opt);
const char* ToString(const FieldType& type);
EncodingOptions EncodingOptionsFromString(std::string_view str);
CompressionOption CompressionOptionFromString(std::string_view str);
FieldType FieldTypeFromString(std::string_view str);

constexpr const uint8_t kEncodingVersion = 5;

struct EncodingInfo {
  std::vector<PointField> fields;
  uint32_t width = 0;   // number of points when height == 1
  uint32_t height = 1;  // > 1 for organised clouds
  uint32_t point_step = 0;
  EncodingOptions encoding_opt = EncodingOptions::LOSSY;
  std::string encoding_config;  // free-form, travels in the header
  CompressionOption compression_opt = CompressionOption::ZSTD;
  bool use_threads = true;  // stage 2 may use worker threads (ignored for NONE)
  uint8_t version = kEncodingVersion;

  bool operator==(const EncodingInfo& o) const {
    return fields == o.fields && width == o.width && height == o.height && point_step == o.point_step &&
           encoding_opt == o.encoding_opt && compression_opt == o.compression_opt;
  }
  bool operator!=(const EncodingInfo& o) const { return !(*this == o); }
};

constexpr const char* kMagicHeader = "CLOUDINI_V";
constexpr int kMagicHeaderLength = 10;

enum class HeaderEncoding { BINARY, YAML };

// "CLOUDINI_V" + two version digits + '\n' + YAML + '\0' (or the legacy binary layout)
void EncodeHeader(const EncodingInfo& header, std::vector<uint8_t>& output,
                  HeaderEncoding encoding = HeaderEncoding::YAML);
// Parses and consumes the header at the front of `input`.
EncodingInfo DecodeHeader(ConstBufferView& input);
std::string EncodingInfoToYAML(const EncodingInfo& info);
EncodingInfo EncodingInfoFromYAML(std::string_view yaml);

// Worst-case size of an encoded cloud: header, one u32 per chunk, and each chunk's worst-case stage-1 size passed
// through the stage-2 bound.
size_t MaxCompressedSize(const EncodingInfo& info, size_t points_count, bool include_header = true);

class PointcloudEncoder {
 public:
  PointcloudEncoder(const EncodingInfo& info);  // (not explicit: the reference's is not, include/cloudini_lib/cloudini.hpp:156)
  ~PointcloudEncoder();
  PointcloudEncoder(const PointcloudEncoder&) = delete;
  PointcloudEncoder& operator=(const PointcloudEncoder&) = delete;

  // Encodes cloud_data (n * point_step bytes) into `output` (resized to fit); returns the encoded size.
  size_t encode(ConstBufferView cloud_data, std::vector<uint8_t>& output);
  // No-allocation variant: `output` must offer MaxCompressedSize bytes (+ the header when write_header).
  size_t encode(ConstBufferView cloud_data, BufferView& output, bool write_header);

  const EncodingInfo& getEncodingInfo() const { return info_; }
  const std::vector<uint8_t>& getHeader() const { return header_; }

 private:
  struct Impl;
  // chunk-group pipeline of encode(): stage 2 of group g next to the GPU's work on group g + 1 (host/cloudini.cpp)
  size_t encodePipelined(ConstBufferView cloud_data, uint64_t points, size_t n_chunks, unsigned workers, uint8_t* dst,
                         size_t dst_cap, std::vector<uint8_t>& stage1, std::vector<uint8_t>& stage2);
  EncodingInfo info_;
  std::vector<uint8_t> header_;
  std::unique_ptr<Impl> impl_;
};

class PointcloudDecoder {
 public:
  PointcloudDecoder();
  ~PointcloudDecoder();
  PointcloudDecoder(const PointcloudDecoder&) = delete;
  PointcloudDecoder& operator=(const PointcloudDecoder&) = delete;

  // compressed_data must NOT start with the header (use DecodeHeader first); output holds width*height*point_step
  // bytes. Bytes of a point that no field covers are left untouched.
  void decode(const EncodingInfo& info, ConstBufferView compressed_data, BufferView output);
  void decode(const EncodingInfo& info, ConstBufferView compressed_data, std::vector<uint8_t>& output) {
    // a vector that arrives empty is all zeros after the resize: nothing of it has to travel to the GPU for the bytes
    // of a point that no field covers (they read 0 afterwards, exactly as in the reference)
    const bool fresh = output.empty();
    output.resize(static_cast<size_t>(info.width) * info.height * info.point_step);
    decodeInto(info, compressed_data, BufferView(output.data(), output.size()), fresh);
  }

  // Not in the reference: decode() for a caller that knows every byte of `output` is 0 on entry (a buffer it has just
  // value-initialised). Same result as decode(); a host buffer of a layout with uncovered bytes then needs no trip to
  // the GPU before the decode.
  void decodeInto(const EncodingInfo& info, ConstBufferView compressed_data, BufferView output, bool output_is_zero);

 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace Cloudini
