// CDR message adapters around PointcloudEncoder / PointcloudDecoder. Follows the reference's
// cloudini_lib/src/ros_msg_utils.cpp (parse :54-97, header writer :99-121, toEncodingInfo :123-132,
// decompress :135-165, compress :167-213, resolution profiles :217-238, applyVizLossyPreprocessing :249-341 -- its
// data path runs on the GPU through cldn_hip_viz_preprocess).
#include "cloudini_lib/ros_msg_utils.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "host_internal.hpp"

namespace cloudini_ros {

RosPointCloud2 getDeserializedPointCloudMessage(Cloudini::ConstBufferView msg) {
  RosPointCloud2 pc;
  nanocdr::Decoder cdr(msg);
  pc.cdr_header = cdr.header();
  cdr.decode(pc.ros_header.stamp_sec);
  cdr.decode(pc.ros_header.stamp_nsec);
  cdr.decode(pc.ros_header.frame_id);
  cdr.decode(pc.height);
  cdr.decode(pc.width);
  uint32_t field_count = 0;
  cdr.decode(field_count);
  for (uint32_t i = 0; i < field_count; ++i) {
    Cloudini::PointField f;
    uint8_t datatype = 0;
    uint32_t count = 0;
    cdr.decode(f.name);
    cdr.decode(f.offset);
    cdr.decode(datatype);
    cdr.decode(count);
    f.type = static_cast<Cloudini::FieldType>(datatype);
    pc.fields.push_back(std::move(f));
  }
  bool big_endian = false;
  cdr.decode(big_endian);  // read and ignored, like the reference
  cdr.decode(pc.point_step);
  cdr.decode(pc.row_step);
  cdr.decode(pc.data);
  cdr.decode(pc.is_dense);
  return pc;
}

void writePointCloudHeader(nanocdr::Encoder& enc, const RosPointCloud2& pc) {
  enc.encode(pc.ros_header.stamp_sec);
  enc.encode(pc.ros_header.stamp_nsec);
  enc.encode(pc.ros_header.frame_id);
  enc.encode(pc.height);
  enc.encode(pc.width);
  enc.encode(static_cast<uint32_t>(pc.fields.size()));
  for (const auto& f : pc.fields) {
    enc.encode(f.name);
    enc.encode(f.offset);
    enc.encode(static_cast<uint8_t>(f.type));
    enc.encode(static_cast<uint32_t>(1));  // count
  }
  enc.encode(false);  // is_bigendian
  enc.encode(pc.point_step);
  enc.encode(static_cast<uint32_t>(pc.point_step * pc.width));  // row_step
}

Cloudini::EncodingInfo toEncodingInfo(const RosPointCloud2& pc) {
  Cloudini::EncodingInfo info;
  info.width = pc.width;
  info.height = pc.height;
  info.point_step = pc.point_step;
  info.fields = pc.fields;
  info.encoding_opt = Cloudini::EncodingOptions::LOSSY;
  info.compression_opt = Cloudini::CompressionOption::ZSTD;
  return info;
}

void convertCompressedCloudToPointCloud2(const RosPointCloud2& pc, std::vector<uint8_t>& out_msg) {
  const size_t cloud_bytes = static_cast<size_t>(pc.width) * pc.height * pc.point_step;
  out_msg.clear();
  nanocdr::Encoder enc(pc.cdr_header, out_msg);
  writePointCloudHeader(enc, pc);
  enc.encode(static_cast<uint32_t>(cloud_bytes));  // length of PointCloud2::data
  if (cloud_bytes == 0) {
    enc.encode(pc.is_dense);
    return;
  }
  const size_t at = out_msg.size();
  out_msg.resize(at + cloud_bytes);
  Cloudini::ConstBufferView stream = pc.data;
  const Cloudini::EncodingInfo info = Cloudini::DecodeHeader(stream);
  Cloudini::PointcloudDecoder decoder;
  // (the bytes just added by resize() are zero: nothing of them has to go to the GPU first)
  decoder.decodeInto(info, stream, Cloudini::BufferView(out_msg.data() + at, cloud_bytes), true);
  enc.encode(pc.is_dense);
}

void convertPointCloud2ToCompressedCloud(const RosPointCloud2& pc, const Cloudini::EncodingInfo& info,
                                         std::vector<uint8_t>& out_msg) {
  out_msg.clear();
  nanocdr::Encoder enc(pc.cdr_header, out_msg);
  writePointCloudHeader(enc, pc);
  const size_t length_at = out_msg.size();  // patched once the encoded size is known (already 4-byte aligned)
  enc.encode(static_cast<uint32_t>(0));
  const size_t payload_at = out_msg.size();
  if (pc.data.size() == 0) {
    enc.encode(pc.is_dense);
    enc.encode(std::string("cloudini"));
    return;
  }
  if (info.point_step == 0) throw std::runtime_error("convertPointCloud2ToCompressedCloud: point_step cannot be 0");
  // the point count comes from the bytes actually present, not from width*height of the (untrusted) message
  const size_t points = pc.data.size() / info.point_step;
  out_msg.resize(payload_at + Cloudini::MaxCompressedSize(info, points, true));
  Cloudini::BufferView room(out_msg.data() + payload_at, out_msg.size() - payload_at);
  Cloudini::PointcloudEncoder encoder(info);
  const size_t encoded = encoder.encode(pc.data, room, true);
  const uint32_t encoded32 = static_cast<uint32_t>(encoded);
  std::memcpy(out_msg.data() + length_at, &encoded32, 4);
  out_msg.resize(payload_at + encoded);
  enc.encode(pc.is_dense);
  enc.encode(std::string("cloudini"));  // CompressedPointCloud2::format
}

void applyResolutionProfile(const ResolutionProfile& profile, std::vector<Cloudini::PointField>& fields,
                            std::optional<float> default_resolution) {
  fields.erase(std::remove_if(fields.begin(), fields.end(),
                              [&](const Cloudini::PointField& f) {
                                const auto it = profile.find(f.name);
                                return it != profile.end() && it->second == 0;
                              }),
               fields.end());
  for (auto& f : fields) {
    const auto it = profile.find(f.name);
    if (it != profile.end()) f.resolution = it->second;
    else if (default_resolution && f.type == Cloudini::FieldType::FLOAT32) f.resolution = *default_resolution;
  }
}

void applyVizLossyPreprocessing(RosPointCloud2& pc_info) {
  // the gate of the reference, src/ros_msg_utils.cpp:250-279
  if (pc_info.fields.size() < 3 || pc_info.point_step == 0) return;
  const auto& f0 = pc_info.fields[0];
  const auto& f1 = pc_info.fields[1];
  const auto& f2 = pc_info.fields[2];
  const bool has_triple = f0.type == Cloudini::FieldType::FLOAT32 && f1.type == Cloudini::FieldType::FLOAT32 &&
                          f2.type == Cloudini::FieldType::FLOAT32 && f0.resolution.has_value() &&
                          f1.resolution.has_value() && f2.resolution.has_value() &&
                          f0.resolution.value() == f1.resolution.value() &&
                          f0.resolution.value() == f2.resolution.value() && f1.offset == f0.offset + 4u &&
                          f2.offset == f0.offset + 8u;
  if (!has_triple) return;
  const float xyz_res = f0.resolution.value();
  if (!(xyz_res > 0.0f) || !std::isfinite(xyz_res)) return;
  const size_t n_in = pc_info.data.size() == 0 ? 0 : pc_info.data.size() / pc_info.point_step;
  if (n_in == 0) return;

  std::vector<uint8_t> out(n_in * pc_info.point_step);
  const uint64_t kept = Cloudini::amd_detail::vizPreprocessOnDevice(pc_info.data.data(), n_in, pc_info.point_step,
                                                                   f0.offset, xyz_res, out.data(), out.size());
  out.resize(static_cast<size_t>(kept) * pc_info.point_step);
  pc_info.owned_data = std::move(out);
  pc_info.data = Cloudini::ConstBufferView(pc_info.owned_data.data(), pc_info.owned_data.size());
  pc_info.width = static_cast<uint32_t>(kept);
  pc_info.height = 1;
  pc_info.row_step = pc_info.point_step * pc_info.width;
  for (auto& f : pc_info.fields) {  // :336-340
    if (f.type == Cloudini::FieldType::FLOAT64 && !f.resolution.has_value()) f.resolution = 1e-6f;
  }
}

}  // namespace cloudini_ros
