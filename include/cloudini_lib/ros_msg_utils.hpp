// ROS-free handling of sensor_msgs/PointCloud2 and point_cloud_interfaces/CompressedPointCloud2 as raw CDR (DDS)
// byte buffers. Source-compatible with the reference's cloudini_lib/ros_msg_utils.hpp (namespace cloudini_ros).
// Message layout being walked:
//   std_msgs/Header header (stamp.sec i32, stamp.nanosec u32, frame_id string)
//   u32 height, u32 width, PointField[] fields {string name, u32 offset, u8 datatype, u32 count},
//   bool is_bigendian, u32 point_step, u32 row_step, u8[] data, bool is_dense [, string format]
#pragma once

#include <map>
#include <optional>
#include <string>
#include <vector>

#include "cloudini_lib/cloudini.hpp"
#include "cloudini_lib/contrib/nanocdr.hpp"

namespace cloudini_ros {

struct RosHeader {
  int32_t stamp_sec = 0;
  uint32_t stamp_nsec = 0;
  std::string frame_id;
};

struct RosPointCloud2 {
  nanocdr::CdrHeader cdr_header;
  RosHeader ros_header;
  uint32_t height = 1;
  uint32_t width = 0;
  std::vector<Cloudini::PointField> fields;
  uint32_t point_step = 0;
  uint32_t row_step = 0;
  bool is_bigendian = false;
  Cloudini::ConstBufferView data;  // normally a view into the DDS message it was parsed from
  bool is_dense = true;
  // When a preprocessing step rewrites the points it stores them here and points `data` at this vector; copies
  // and moves keep that link intact.
  std::vector<uint8_t> owned_data;

  RosPointCloud2() = default;
  RosPointCloud2(const RosPointCloud2& o) { assign(o); }
  RosPointCloud2(RosPointCloud2&& o) noexcept { assign(std::move(o)); }
  RosPointCloud2& operator=(const RosPointCloud2& o) {
    if (this != &o) assign(o);
    return *this;
  }
  RosPointCloud2& operator=(RosPointCloud2&& o) noexcept {
    if (this != &o) assign(std::move(o));
    return *this;
  }

 private:
  bool viewsOwnData() const {
    return !owned_data.empty() && data.data() == owned_data.data() && data.size() == owned_data.size();
  }
  void copyScalars(const RosPointCloud2& o) {
    cdr_header = o.cdr_header;
    height = o.height;
    width = o.width;
    point_step = o.point_step;
    row_step = o.row_step;
    is_bigendian = o.is_bigendian;
    is_dense = o.is_dense;
    data = o.data;
  }
  void assign(const RosPointCloud2& o) {
    const bool rebind = o.viewsOwnData();
    copyScalars(o);
    ros_header = o.ros_header;
    fields = o.fields;
    owned_data = o.owned_data;
    if (rebind) data = Cloudini::ConstBufferView(owned_data.data(), owned_data.size());
  }
  void assign(RosPointCloud2&& o) {
    const bool rebind = o.viewsOwnData();
    copyScalars(o);
    ros_header = std::move(o.ros_header);
    fields = std::move(o.fields);
    owned_data = std::move(o.owned_data);
    if (rebind) data = Cloudini::ConstBufferView(owned_data.data(), owned_data.size());
  }
};

// field name -> resolution; a resolution of 0 removes the field
using ResolutionProfile = std::map<std::string, float>;

void applyResolutionProfile(const ResolutionProfile& profile, std::vector<Cloudini::PointField>& fields,
                            std::optional<float> default_resolution = std::nullopt);

Cloudini::EncodingInfo toEncodingInfo(const RosPointCloud2& pc_info);

// Visualisation-oriented lossy pre-filter that runs in front of the encoder (reference ros_msg_utils.hpp:175-221):
// if the first three fields are FLOAT32 with one shared resolution and offsets {b, b+4, b+8}, points with a
// non-finite coordinate are dropped and of all points of one voxel (at that resolution) only the first is kept,
// order preserved -- on the GPU (cldn_hip_viz_preprocess). pc_info.owned_data receives the surviving points, data
// views it, width = survivors, height = 1, row_step = point_step * width; FLOAT64 fields without a resolution get
// 1e-6. No-op without such a triple, with a non-positive / non-finite resolution, or with an empty cloud.
void applyVizLossyPreprocessing(RosPointCloud2& pc_info);

void writePointCloudHeader(nanocdr::Encoder& encoder, const RosPointCloud2& pc_info);

// sensor_msgs/PointCloud2 or CompressedPointCloud2 (same layout up to `data`) -> RosPointCloud2
RosPointCloud2 getDeserializedPointCloudMessage(Cloudini::ConstBufferView pc2_dds_msg);

// PointCloud2 -> CompressedPointCloud2 (encodes pc_info.data with PointcloudEncoder on the GPU)
void convertPointCloud2ToCompressedCloud(const RosPointCloud2& pc_info, const Cloudini::EncodingInfo& encoding_info,
                                         std::vector<uint8_t>& compressed_dds_msg);

// CompressedPointCloud2 (pc_info.data = header + chunks) -> PointCloud2
void convertCompressedCloudToPointCloud2(const RosPointCloud2& pc_info, std::vector<uint8_t>& pc2_dds_msg);

}  // namespace cloudini_ros
