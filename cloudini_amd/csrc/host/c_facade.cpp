// C entry points over the C++ host API: the reference's own `cldn_*` ABI (include/cloudini_lib/wasm_functions.h)
// and the flat helpers of include/cloudini_amd_c.h used by the Python bindings and tests.
#include <cstring>
#include <limits>
#include <string>

#include "cloudini_amd_c.h"
#include "cloudini_lib/cloudini.hpp"
#include "cloudini_lib/ros_msg_utils.hpp"
#include "cloudini_lib/wasm_functions.h"
#include "cloudini_amd/batch_transcoder.hpp"
#include "host_internal.hpp"

#define CLDN_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_error;

Cloudini::EncodingInfo toInfo(const cldn_amd_info_t* in) {
  Cloudini::EncodingInfo info;
  for (uint32_t i = 0; i < in->n_fields; ++i) {
    Cloudini::PointField f;
    f.name = in->fields[i].name ? in->fields[i].name : "";
    f.offset = in->fields[i].offset;
    f.type = static_cast<Cloudini::FieldType>(in->fields[i].type);
    if (in->fields[i].has_resolution) f.resolution = in->fields[i].resolution;
    info.fields.push_back(std::move(f));
  }
  info.width = in->width;
  info.height = in->height;
  info.point_step = in->point_step;
  info.encoding_opt = static_cast<Cloudini::EncodingOptions>(in->encoding_opt);
  info.compression_opt = static_cast<Cloudini::CompressionOption>(in->compression_opt);
  info.version = in->version;
  info.use_threads = in->use_threads != 0;
  return info;
}

template <typename Fn>
int64_t guarded(Fn&& fn) {
  try {
    g_error.clear();
    return fn();
  } catch (const std::exception& e) {
    g_error = e.what();
    return -1;
  }
}

template <typename Fn>
uint32_t guarded0(Fn&& fn) {  // the reference's "0 on failure" convention
  const int64_t r = guarded(fn);
  return r < 0 ? 0u : static_cast<uint32_t>(r);
}

int64_t copyOut(const std::vector<uint8_t>& v, uint8_t* out, uint64_t capacity) {
  if (v.size() > capacity) throw std::runtime_error("output buffer too small");
  std::memcpy(out, v.data(), v.size());
  return static_cast<int64_t>(v.size());
}

void setFloatResolution(Cloudini::EncodingInfo& info, float resolution) {
  for (auto& f : info.fields)
    if (f.type == Cloudini::FieldType::FLOAT32) f.resolution = resolution;
}

}  // namespace

CLDN_EXPORT const char* cldn_amd_last_error(void) { return g_error.c_str(); }
CLDN_EXPORT const char* cldn_LastError(void) { return g_error.c_str(); }

CLDN_EXPORT int64_t cldn_amd_max_compressed_size(const cldn_amd_info_t* info, uint64_t n_points, int include_header) {
  return guarded([&] { return (int64_t)Cloudini::MaxCompressedSize(toInfo(info), n_points, include_header != 0); });
}

CLDN_EXPORT int64_t cldn_amd_encode_header(const cldn_amd_info_t* info, int binary, uint8_t* out, uint64_t capacity) {
  return guarded([&] {
    std::vector<uint8_t> hdr;
    Cloudini::EncodeHeader(toInfo(info), hdr, binary ? Cloudini::HeaderEncoding::BINARY : Cloudini::HeaderEncoding::YAML);
    return copyOut(hdr, out, capacity);
  });
}

CLDN_EXPORT int64_t cldn_amd_encode(const cldn_amd_info_t* info, const uint8_t* data, uint64_t size, uint8_t* out,
                                    uint64_t capacity, int write_header) {
  return guarded([&] {
    Cloudini::PointcloudEncoder encoder(toInfo(info));
    Cloudini::BufferView view(out, capacity);
    return (int64_t)encoder.encode(Cloudini::ConstBufferView(data, size), view, write_header != 0);
  });
}

CLDN_EXPORT int64_t cldn_amd_transcode_directory_on(const char* in_dir, const char* out_dir, float resolution,
                                                    uint8_t compression_opt, int viz_lossy, uint32_t batch_messages,
                                                    const int32_t* devices, uint32_t n_devices, double* stats_out) {
  return guarded([&] {
    cloudini_amd::DirectorySource source(in_dir);
    cloudini_amd::DirectorySink sink(out_dir);
    cloudini_amd::TranscodeOptions opt;
    opt.default_resolution = resolution;
    opt.compression = static_cast<Cloudini::CompressionOption>(compression_opt);
    opt.viz_lossy = viz_lossy != 0;
    if (batch_messages) opt.batch_messages = batch_messages;
    if (devices) opt.devices.assign(devices, devices + n_devices);
    const cloudini_amd::TranscodeStats st = cloudini_amd::transcodePointClouds(source, sink, opt);
    if (stats_out) {
      const double v[8] = {(double)st.messages,    (double)st.points,  (double)st.input_bytes, (double)st.output_bytes,
                           (double)st.gpu_batches, st.seconds_total, st.seconds_gpu,         st.seconds_stage2};
      for (int i = 0; i < 8; ++i) stats_out[i] = v[i];
    }
    return (int64_t)st.messages;
  });
}

CLDN_EXPORT int64_t cldn_amd_transcode_directory(const char* in_dir, const char* out_dir, float resolution,
                                                 uint8_t compression_opt, int viz_lossy, uint32_t batch_messages,
                                                 double* stats_out) {
  return cldn_amd_transcode_directory_on(in_dir, out_dir, resolution, compression_opt, viz_lossy, batch_messages, nullptr, 0,
                                         stats_out);
}

CLDN_EXPORT int64_t cldn_amd_decode_directory_on(const char* in_dir, const char* out_dir, uint32_t batch_messages,
                                                 const int32_t* devices, uint32_t n_devices, double* stats_out) {
  return guarded([&] {
    cloudini_amd::DirectorySource source(in_dir);
    cloudini_amd::DirectorySink sink(out_dir);
    cloudini_amd::TranscodeOptions opt;
    opt.decode = true;
    if (batch_messages) opt.batch_messages = batch_messages;
    if (devices) opt.devices.assign(devices, devices + n_devices);
    const cloudini_amd::TranscodeStats st = cloudini_amd::transcodePointClouds(source, sink, opt);
    if (stats_out) {
      const double v[8] = {(double)st.messages,    (double)st.points,  (double)st.input_bytes, (double)st.output_bytes,
                           (double)st.gpu_batches, st.seconds_total, st.seconds_gpu,         st.seconds_stage2};
      for (int i = 0; i < 8; ++i) stats_out[i] = v[i];
    }
    return (int64_t)st.messages;
  });
}

CLDN_EXPORT int64_t cldn_amd_decode_directory(const char* in_dir, const char* out_dir, uint32_t batch_messages,
                                              double* stats_out) {
  return cldn_amd_decode_directory_on(in_dir, out_dir, batch_messages, nullptr, 0, stats_out);
}

CLDN_EXPORT int cldn_amd_device_lz4(void) { return Cloudini::amd_detail::deviceLz4Level(); }
CLDN_EXPORT int cldn_amd_set_device_lz4(int on) {
  Cloudini::amd_detail::setDeviceLz4Level(on);
  return Cloudini::amd_detail::deviceLz4Level();
}

CLDN_EXPORT uint32_t cldn_amd_stage2_threads(void) { return Cloudini::amd_detail::stage2Threads(); }
CLDN_EXPORT uint32_t cldn_amd_set_stage2_threads(uint32_t n) {
  Cloudini::amd_detail::setStage2Threads(n);
  return Cloudini::amd_detail::stage2Threads();
}

CLDN_EXPORT int64_t cldn_amd_decode(const uint8_t* stream, uint64_t size, uint8_t* out, uint64_t capacity,
                                    char* yaml_out, uint64_t yaml_capacity, uint8_t* version_out) {
  return guarded([&] {
    Cloudini::ConstBufferView in(stream, size);
    const Cloudini::EncodingInfo info = Cloudini::DecodeHeader(in);
    if (yaml_out && yaml_capacity) {
      const std::string yaml = Cloudini::EncodingInfoToYAML(info);
      const size_t n = std::min<size_t>(yaml.size(), yaml_capacity - 1);
      std::memcpy(yaml_out, yaml.data(), n);
      yaml_out[n] = 0;
    }
    if (version_out) *version_out = info.version;
    const uint64_t need = (uint64_t)info.width * info.height * info.point_step;
    if (need > capacity) throw std::runtime_error("decode buffer too small");
    Cloudini::PointcloudDecoder decoder;
    decoder.decode(info, in, Cloudini::BufferView(out, need));
    return (int64_t)need;
  });
}

CLDN_EXPORT int64_t cldn_amd_decode_noheader(const cldn_amd_info_t* info_in, const uint8_t* data, uint64_t size,
                                             uint8_t* out, uint64_t capacity) {
  return guarded([&] {
    const Cloudini::EncodingInfo info = toInfo(info_in);
    const uint64_t need = (uint64_t)info.width * info.height * info.point_step;
    if (need > capacity) throw std::runtime_error("decode buffer too small");
    Cloudini::PointcloudDecoder decoder;
    decoder.decode(info, Cloudini::ConstBufferView(data, size), Cloudini::BufferView(out, need));
    return (int64_t)need;
  });
}

CLDN_EXPORT int64_t cldn_amd_decode_noheader_zeroed(const cldn_amd_info_t* info_in, const uint8_t* data, uint64_t size,
                                                    uint8_t* out, uint64_t capacity) {
  return guarded([&] {
    const Cloudini::EncodingInfo info = toInfo(info_in);
    const uint64_t need = (uint64_t)info.width * info.height * info.point_step;
    if (need > capacity) throw std::runtime_error("decode buffer too small");
    Cloudini::PointcloudDecoder decoder;
    decoder.decodeInto(info, Cloudini::ConstBufferView(data, size), Cloudini::BufferView(out, need), true);
    return (int64_t)need;
  });
}

CLDN_EXPORT int64_t cldn_amd_ros_compress(const uint8_t* dds, uint64_t size, float resolution, uint8_t compression_opt,
                                          uint8_t* out, uint64_t capacity) {
  return guarded([&] {
    auto pc = cloudini_ros::getDeserializedPointCloudMessage(Cloudini::ConstBufferView(dds, size));
    cloudini_ros::applyResolutionProfile({}, pc.fields, resolution);
    Cloudini::EncodingInfo info = cloudini_ros::toEncodingInfo(pc);
    info.compression_opt = static_cast<Cloudini::CompressionOption>(compression_opt);
    std::vector<uint8_t> msg;
    cloudini_ros::convertPointCloud2ToCompressedCloud(pc, info, msg);
    return copyOut(msg, out, capacity);
  });
}

CLDN_EXPORT int64_t cldn_amd_ros_decompress(const uint8_t* dds, uint64_t size, uint8_t* out, uint64_t capacity) {
  return guarded([&] {
    const auto pc = cloudini_ros::getDeserializedPointCloudMessage(Cloudini::ConstBufferView(dds, size));
    std::vector<uint8_t> msg;
    cloudini_ros::convertCompressedCloudToPointCloud2(pc, msg);
    return copyOut(msg, out, capacity);
  });
}

CLDN_EXPORT int64_t cldn_amd_viz_preprocess(const cldn_amd_info_t* info, const uint8_t* data, uint64_t size, uint8_t* out,
                                            uint64_t capacity, float* res_out, uint32_t* width_out, uint32_t* height_out) {
  return guarded([&] {
    cloudini_ros::RosPointCloud2 pc;
    for (uint32_t i = 0; i < info->n_fields; ++i) {
      Cloudini::PointField f;
      f.name = info->fields[i].name ? info->fields[i].name : "";
      f.offset = info->fields[i].offset;
      f.type = static_cast<Cloudini::FieldType>(info->fields[i].type);
      if (info->fields[i].has_resolution) f.resolution = info->fields[i].resolution;
      pc.fields.push_back(f);
    }
    pc.point_step = info->point_step;
    pc.width = info->point_step ? static_cast<uint32_t>(size / info->point_step) : 0;
    pc.height = 1;
    pc.row_step = pc.width * pc.point_step;
    pc.data = Cloudini::ConstBufferView(data, size);
    cloudini_ros::applyVizLossyPreprocessing(pc);
    if (pc.data.size() > capacity) throw std::runtime_error("viz_preprocess: output capacity");
    if (pc.data.size()) std::memcpy(out, pc.data.data(), pc.data.size());
    for (uint32_t i = 0; i < info->n_fields; ++i)
      res_out[i] = pc.fields[i].resolution ? *pc.fields[i].resolution : std::numeric_limits<float>::quiet_NaN();
    *width_out = pc.width;
    *height_out = pc.height;
    return static_cast<int64_t>(pc.data.size());
  });
}

// ---- the reference's C ABI (src/wasm_functions.cpp) ---------------------------------------------------------------

CLDN_EXPORT uint32_t cldn_GetHeaderAsYAML(uintptr_t encoded_data_ptr, uint32_t encoded_data_size, uintptr_t output_yaml_ptr) {
  return guarded0([&] {
    Cloudini::ConstBufferView in(reinterpret_cast<const uint8_t*>(encoded_data_ptr), encoded_data_size);
    const std::string yaml = Cloudini::EncodingInfoToYAML(Cloudini::DecodeHeader(in));
    std::memcpy(reinterpret_cast<char*>(output_yaml_ptr), yaml.data(), yaml.size());
    return (int64_t)yaml.size();
  });
}

CLDN_EXPORT uint32_t cldn_GetHeaderAsYAMLFromDDS(uintptr_t raw_dds_msg, uint32_t dds_msg_size, uintptr_t output_yaml_ptr) {
  return guarded0([&] {
    const auto pc = cloudini_ros::getDeserializedPointCloudMessage(
        Cloudini::ConstBufferView(reinterpret_cast<const uint8_t*>(raw_dds_msg), dds_msg_size));
    return (int64_t)cldn_GetHeaderAsYAML(reinterpret_cast<uintptr_t>(pc.data.data()), (uint32_t)pc.data.size(), output_yaml_ptr);
  });
}

CLDN_EXPORT uint32_t cldn_ComputeCompressedSize(uintptr_t dds_msg_ptr, uint32_t dds_msg_size, float resolution) {
  return guarded0([&] {
    const auto pc = cloudini_ros::getDeserializedPointCloudMessage(
        Cloudini::ConstBufferView(reinterpret_cast<const uint8_t*>(dds_msg_ptr), dds_msg_size));
    Cloudini::EncodingInfo info = cloudini_ros::toEncodingInfo(pc);
    setFloatResolution(info, resolution);
    if (pc.data.size() != (size_t)pc.width * pc.height * pc.point_step && (pc.width == 0 || pc.height == 0)) return (int64_t)0;
    Cloudini::PointcloudEncoder encoder(info);
    std::vector<uint8_t> encoded;
    return (int64_t)encoder.encode(pc.data, encoded);
  });
}

CLDN_EXPORT uint32_t cldn_GetDecompressedSize(uintptr_t encoded_msg_ptr, uint32_t encoded_msg_size) {
  return guarded0([&] {
    const auto pc = cloudini_ros::getDeserializedPointCloudMessage(
        Cloudini::ConstBufferView(reinterpret_cast<const uint8_t*>(encoded_msg_ptr), encoded_msg_size));
    return (int64_t)pc.height * pc.width * pc.point_step;
  });
}

CLDN_EXPORT uint32_t cldn_ConvertCompressedMsgToPointCloud2Msg(uintptr_t compressed_msg_ptr, uint32_t encoded_data_size,
                                                               uintptr_t output_msg_ptr) {
  return guarded0([&] {
    const auto pc = cloudini_ros::getDeserializedPointCloudMessage(
        Cloudini::ConstBufferView(reinterpret_cast<const uint8_t*>(compressed_msg_ptr), encoded_data_size));
    std::vector<uint8_t> msg;
    cloudini_ros::convertCompressedCloudToPointCloud2(pc, msg);
    std::memcpy(reinterpret_cast<void*>(output_msg_ptr), msg.data(), msg.size());
    return (int64_t)msg.size();
  });
}

CLDN_EXPORT uint32_t cldn_DecodeCompressedData(uintptr_t encoded_data_ptr, uint32_t encoded_data_size, uintptr_t output_data) {
  return guarded0([&] {
    Cloudini::ConstBufferView in(reinterpret_cast<const uint8_t*>(encoded_data_ptr), encoded_data_size);
    const Cloudini::EncodingInfo info = Cloudini::DecodeHeader(in);
    const size_t bytes = (size_t)info.width * info.height * info.point_step;
    Cloudini::PointcloudDecoder decoder;
    decoder.decode(info, in, Cloudini::BufferView(reinterpret_cast<uint8_t*>(output_data), bytes));
    return (int64_t)bytes;
  });
}

CLDN_EXPORT uint32_t cldn_DecodeCompressedMessage(uintptr_t compressed_msg_ptr, uint32_t msg_size, uintptr_t output_data_ptr) {
  return guarded0([&] {
    const auto pc = cloudini_ros::getDeserializedPointCloudMessage(
        Cloudini::ConstBufferView(reinterpret_cast<const uint8_t*>(compressed_msg_ptr), msg_size));
    return (int64_t)cldn_DecodeCompressedData(reinterpret_cast<uintptr_t>(pc.data.data()), (uint32_t)pc.data.size(), output_data_ptr);
  });
}

CLDN_EXPORT uint32_t cldn_EncodePointcloudMessage(const uintptr_t pointcloud_msg_ptr, uint32_t msg_size, float resolution,
                                                  uintptr_t output_data_ptr) {
  return guarded0([&] {
    const auto pc = cloudini_ros::getDeserializedPointCloudMessage(
        Cloudini::ConstBufferView(reinterpret_cast<const uint8_t*>(pointcloud_msg_ptr), msg_size));
    Cloudini::EncodingInfo info = cloudini_ros::toEncodingInfo(pc);
    setFloatResolution(info, resolution);
    if (pc.data.size() != (size_t)pc.width * pc.height * pc.point_step) return (int64_t)0;
    Cloudini::PointcloudEncoder encoder(info);
    std::vector<uint8_t> encoded;
    const size_t n = encoder.encode(pc.data, encoded);
    if (n > msg_size) throw std::runtime_error("Output buffer too small for encoded message");  // caller allocates msg_size
    std::memcpy(reinterpret_cast<void*>(output_data_ptr), encoded.data(), n);
    return (int64_t)n;
  });
}

CLDN_EXPORT uint32_t cldn_EncodePointcloudData(const char* header_as_yaml, const uintptr_t pc_data_ptr, uint32_t pc_data_size,
                                               uintptr_t output_data_ptr) {
  return guarded0([&] {
    const Cloudini::EncodingInfo info = Cloudini::EncodingInfoFromYAML(header_as_yaml);
    const uint64_t cells = (uint64_t)info.width * info.height;  // 64-bit products: the header text is untrusted
    if (cells > 0xffffffffull || cells * info.point_step != (uint64_t)pc_data_size) throw std::runtime_error("Data size mismatch");
    Cloudini::PointcloudEncoder encoder(info);
    std::vector<uint8_t> encoded;
    const size_t n = encoder.encode(Cloudini::ConstBufferView(reinterpret_cast<const uint8_t*>(pc_data_ptr), pc_data_size), encoded);
    if (n > pc_data_size) throw std::runtime_error("Output buffer too small for encoded data");  // caller allocates pc_data_size
    std::memcpy(reinterpret_cast<void*>(output_data_ptr), encoded.data(), n);
    return (int64_t)n;
  });
}
