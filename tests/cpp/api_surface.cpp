// Source compatibility of the host mirror with the reference's public headers, CPU-only parts: compiled against
// include/cloudini_lib/*.hpp and linked with libcloudini_amd.so by tests/test_cpp_api_surface.py. Mirrors
// test_header.cpp (Header, HeaderTruncatedInput, HeaderMissingYamlTerminator), test_ros_msg.cpp
// (RosPointCloud2CopyRebindsOwnedDataView, :146-175) and the Span contract of contrib/span.hpp.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "cloudini_lib/cloudini.hpp"
#include "cloudini_lib/ros_msg_utils.hpp"

static int failures = 0;
#define CHECK(cond)                                                       \
  do {                                                                    \
    if (!(cond)) {                                                        \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);       \
      ++failures;                                                         \
    }                                                                     \
  } while (0)

template <typename F>
static bool throws(F&& f) {
  try {
    f();
  } catch (const std::exception&) {
    return true;
  }
  return false;
}

int main() {
  using namespace Cloudini;
  // ---- header round trip (test_header.cpp:24-105)
  EncodingInfo info;
  info.width = 10;
  info.height = 20;
  info.point_step = 16;
  info.encoding_opt = EncodingOptions::LOSSY;
  info.compression_opt = CompressionOption::ZSTD;
  info.fields.push_back({"x", 0, FieldType::FLOAT32, 0.01f});
  info.fields.push_back({"y", 4, FieldType::FLOAT32, 0.01f});
  info.fields.push_back({"z", 8, FieldType::FLOAT32, 0.01f});
  info.fields.push_back({"intensity", 12, FieldType::UINT16, std::nullopt});
  std::vector<uint8_t> header;
  EncodeHeader(info, header);
  CHECK(header.size() > 13 && std::memcmp(header.data(), "CLOUDINI_V05\n", 13) == 0);
  ConstBufferView view(header.data(), header.size());
  const EncodingInfo back = DecodeHeader(view);
  CHECK(view.empty());  // DecodeHeader consumes the header
  CHECK(back.width == 10 && back.height == 20 && back.point_step == 16);
  CHECK(back.encoding_opt == EncodingOptions::LOSSY && back.compression_opt == CompressionOption::ZSTD);
  CHECK(back.fields.size() == 4 && back.fields[3].name == "intensity" && back.fields[3].type == FieldType::UINT16);
  CHECK(back.fields[0].resolution.has_value() && !back.fields[3].resolution.has_value());
  CHECK(back.version == 5);
  const std::string yaml = EncodingInfoToYAML(info);
  const EncodingInfo from_yaml = EncodingInfoFromYAML(yaml);
  CHECK(from_yaml.fields.size() == 4 && from_yaml.point_step == 16);

  // ---- malformed headers (test_header.cpp:165-171, :243-262)
  CHECK(throws([&] {
    ConstBufferView v(header.data(), 5);
    DecodeHeader(v);
  }));
  {
    std::vector<uint8_t> no_nul(header.begin(), header.end() - 1);  // YAML terminator missing
    CHECK(throws([&] {
      ConstBufferView v(no_nul.data(), no_nul.size());
      DecodeHeader(v);
    }));
    std::vector<uint8_t> bad_magic = header;
    bad_magic[0] = 'X';
    CHECK(throws([&] {
      ConstBufferView v(bad_magic.data(), bad_magic.size());
      DecodeHeader(v);
    }));
  }

  // ---- capacity contract (cloudini.cpp:249-292): monotonic, header adds its size
  CHECK(MaxCompressedSize(info, 1000, true) > MaxCompressedSize(info, 1000, false));
  CHECK(MaxCompressedSize(info, 2000, false) > MaxCompressedSize(info, 1000, false));
  EncodingInfo zero_step = info;
  zero_step.point_step = 0;
  CHECK(throws([&] { MaxCompressedSize(zero_step, 10, false); }));

  // ---- Span semantics (contrib/span.hpp): trim_front throws past the end
  {
    uint8_t bytes[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    ConstBufferView s(bytes, 8);
    s.trim_front(3);
    CHECK(s.size() == 5 && s.data() == bytes + 3);
    CHECK(throws([&] { s.trim_front(6); }));
    BufferView w(bytes, 8);
    w.trim_front(8);
    CHECK(w.empty());
  }

  // ---- RosPointCloud2 copies rebind the view onto their own owned_data (test_ros_msg.cpp:146-175)
  {
    cloudini_ros::RosPointCloud2 a;
    a.owned_data = {1, 2, 3, 4, 5, 6};
    a.data = ConstBufferView(a.owned_data.data(), a.owned_data.size());
    a.point_step = 2;
    a.width = 3;
    cloudini_ros::RosPointCloud2 b = a;
    CHECK(b.data.data() == b.owned_data.data() && b.data.size() == 6 && b.data.data() != a.data.data());
    cloudini_ros::RosPointCloud2 c;
    c = a;
    CHECK(c.data.data() == c.owned_data.data() && c.data.size() == 6);
    cloudini_ros::RosPointCloud2 d = std::move(b);
    CHECK(d.data.data() == d.owned_data.data() && d.data.size() == 6);
    // a view onto foreign memory is copied as is
    uint8_t foreign[4] = {9, 9, 9, 9};
    cloudini_ros::RosPointCloud2 e;
    e.data = ConstBufferView(foreign, 4);
    cloudini_ros::RosPointCloud2 f = e;
    CHECK(f.data.data() == foreign && f.owned_data.empty());
  }

  // ---- resolution profiles and schema conversion (ros_msg_utils.cpp:123-132, :217-238)
  {
    cloudini_ros::RosPointCloud2 pc;
    pc.fields.push_back({"x", 0, FieldType::FLOAT32, std::nullopt});
    pc.fields.push_back({"y", 4, FieldType::FLOAT32, std::nullopt});
    pc.fields.push_back({"ring", 8, FieldType::UINT16, std::nullopt});
    pc.fields.push_back({"junk", 10, FieldType::UINT16, std::nullopt});
    pc.point_step = 12;
    pc.width = 7;
    pc.height = 1;
    cloudini_ros::applyResolutionProfile({{"y", 0.5f}, {"junk", 0.0f}}, pc.fields, 0.001f);
    CHECK(pc.fields.size() == 3);                                    // resolution 0 removes the field
    CHECK(pc.fields[0].resolution.value() == 0.001f);                // default for FLOAT32 without an entry
    CHECK(pc.fields[1].resolution.value() == 0.5f);
    CHECK(!pc.fields[2].resolution.has_value());
    const EncodingInfo ei = cloudini_ros::toEncodingInfo(pc);
    CHECK(ei.width == 7 && ei.height == 1 && ei.point_step == 12 && ei.fields.size() == 3);
    // no geometry triple: the pre-filter leaves the cloud alone (and needs no GPU for that)
    std::vector<uint8_t> bytes(12 * 7, 1);
    pc.data = ConstBufferView(bytes.data(), bytes.size());
    cloudini_ros::applyVizLossyPreprocessing(pc);
    CHECK(pc.data.data() == bytes.data() && pc.width == 7);
  }

  // ---- encoder argument checks that fire before any device work (cloudini.cpp:505-534)
  {
    EncodingInfo bad = info;
    bad.point_step = 0;
    CHECK(throws([&] { PointcloudEncoder enc(bad); }));
  }
  std::printf(failures ? "%d check(s) failed\n" : "all checks passed\n", failures);
  return failures ? 1 : 0;
}
