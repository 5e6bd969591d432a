// Schema vocabulary of the Cloudini API (drop-in for the reference's cloudini_lib/basic_types.hpp).
#pragma once

#include <cstdint>
#include <limits>
#include <optional>
#include <string>

namespace Cloudini {

// Values 1..8 are sensor_msgs/PointField datatypes; 9 and 10 extend them with 64-bit integers.
enum class FieldType : uint8_t {
  UNKNOWN = 0,
  INT8 = 1,
  UINT8 = 2,
  INT16 = 3,
  UINT16 = 4,
  INT32 = 5,
  UINT32 = 6,
  FLOAT32 = 7,
  FLOAT64 = 8,
  INT64 = 9,
  UINT64 = 10,
};

struct PointField {
  std::string name;
  uint32_t offset = 0;  // byte offset inside one point
  FieldType type = FieldType::UNKNOWN;
  // Quantisation step of lossy encoding; the reconstruction error is at most resolution / 2.
  std::optional<float> resolution;

  bool operator==(const PointField& o) const {
    return offset == o.offset && type == o.type && name == o.name && resolution == o.resolution;
  }
  bool operator!=(const PointField& o) const { return !(*this == o); }
};

// A decoder-side field with this offset is parsed but not written to the output point.
constexpr static uint32_t kDecodeButSkipStore = std::numeric_limits<uint32_t>::max();

constexpr int SizeOf(const FieldType& type) {
  switch (type) {
    case FieldType::INT8: case FieldType::UINT8: return 1;
    case FieldType::INT16: case FieldType::UINT16: return 2;
    case FieldType::INT32: case FieldType::UINT32: case FieldType::FLOAT32: return 4;
    case FieldType::FLOAT64: case FieldType::INT64: case FieldType::UINT64: return 8;
    default: return 0;
  }
}

}  // namespace Cloudini
