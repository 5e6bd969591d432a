// A caller's view of the host mirror on a GPU box: the reference's README usage (PointcloudEncoder -> vector,
// DecodeHeader -> PointcloudDecoder) with the XYZI struct of test_field_encoders.cpp:695-769 and the tolerance the
// reference's own test applies (0.0011 at 1 mm), for every compression option; plus the pre-filter in front of it.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "cloudini_lib/cloudini.hpp"
#include "cloudini_lib/ros_msg_utils.hpp"

struct PointXYZI {
  float x, y, z, intensity;
};

int main() {
  using namespace Cloudini;
  const size_t n = 100000;
  std::vector<PointXYZI> cloud(n);
  for (size_t i = 0; i < n; ++i) {
    const float t = 0.001f * static_cast<float>(i);
    cloud[i] = {10.0f * std::cos(t), 10.0f * std::sin(t), 0.1f * t, static_cast<float>(i % 256)};
  }
  cloud[17].x = std::nanf("");
  int failures = 0;
  for (CompressionOption comp : {CompressionOption::NONE, CompressionOption::LZ4, CompressionOption::ZSTD}) {
    EncodingInfo info;
    info.width = static_cast<uint32_t>(n);
    info.height = 1;
    info.point_step = sizeof(PointXYZI);
    info.encoding_opt = EncodingOptions::LOSSY;
    info.compression_opt = comp;
    info.fields.push_back({"x", 0, FieldType::FLOAT32, 0.001f});
    info.fields.push_back({"y", 4, FieldType::FLOAT32, 0.001f});
    info.fields.push_back({"z", 8, FieldType::FLOAT32, 0.001f});
    info.fields.push_back({"intensity", 12, FieldType::FLOAT32, 0.001f});
    PointcloudEncoder encoder(info);
    std::vector<uint8_t> compressed;
    ConstBufferView in(reinterpret_cast<const uint8_t*>(cloud.data()), n * sizeof(PointXYZI));
    const size_t size = encoder.encode(in, compressed);
    if (size != compressed.size() || size >= n * sizeof(PointXYZI)) {
      std::printf("FAILED: unexpected size %zu\n", size);
      ++failures;
    }
    ConstBufferView stream(compressed.data(), compressed.size());
    const EncodingInfo header = DecodeHeader(stream);
    std::vector<PointXYZI> back(n);
    BufferView out(reinterpret_cast<uint8_t*>(back.data()), n * sizeof(PointXYZI));
    PointcloudDecoder decoder;
    decoder.decode(header, stream, out);
    for (size_t i = 0; i < n; ++i) {
      const float* a = &cloud[i].x;
      const float* b = &back[i].x;
      for (int k = 0; k < 4; ++k) {
        const bool ok = std::isnan(a[k]) ? std::isnan(b[k]) : std::fabs(a[k] - b[k]) <= 0.0011f;
        if (!ok && failures < 5) {
          std::printf("FAILED: point %zu lane %d: %g vs %g\n", i, k, a[k], b[k]);
          ++failures;
        }
      }
    }
    std::printf("compression %d: %zu -> %zu bytes\n", static_cast<int>(comp), n * sizeof(PointXYZI), size);
  }
  // pre-filter: NaN point dropped, coarse voxels collapse
  cloudini_ros::RosPointCloud2 pc;
  pc.fields.push_back({"x", 0, FieldType::FLOAT32, 0.5f});
  pc.fields.push_back({"y", 4, FieldType::FLOAT32, 0.5f});
  pc.fields.push_back({"z", 8, FieldType::FLOAT32, 0.5f});
  pc.fields.push_back({"intensity", 12, FieldType::FLOAT32, std::nullopt});
  pc.point_step = sizeof(PointXYZI);
  pc.width = static_cast<uint32_t>(n);
  pc.height = 1;
  pc.data = ConstBufferView(reinterpret_cast<const uint8_t*>(cloud.data()), n * sizeof(PointXYZI));
  cloudini_ros::applyVizLossyPreprocessing(pc);
  if (pc.width == 0 || pc.width >= n || pc.data.size() != size_t(pc.width) * sizeof(PointXYZI) || pc.height != 1) {
    std::printf("FAILED: pre-filter kept %u of %zu\n", pc.width, n);
    ++failures;
  }
  std::printf("pre-filter kept %u of %zu points\n", pc.width, n);
  std::printf(failures ? "%d failure(s)\n" : "all checks passed\n", failures);
  return failures ? 1 : 0;
}
