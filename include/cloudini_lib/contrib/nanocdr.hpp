// Minimal CDR (DDS serialisation) reader / writer: exactly what is needed to walk sensor_msgs/PointCloud2 and
// point_cloud_interfaces/CompressedPointCloud2 messages without ROS. Interface-compatible with the subset of the
// reference's contrib/nanocdr.hpp that ros_msg_utils uses (nanocdr::CdrHeader, Decoder, Encoder).
//
// Wire rules implemented: 4-byte encapsulation header {0, representation, 0, 0}; primitives are aligned to their
// size relative to the first byte after that header (8-byte types align to 4 under XCDR2); strings are a u32
// length (terminator included) + bytes + NUL; byte sequences are a u32 length + bytes.
#pragma once

#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "cloudini_lib/contrib/span.hpp"

namespace nanocdr {

using ConstBuffer = Span<const uint8_t>;

enum class CdrVersion : uint8_t { DDS_CDR = 1, XCDRv1 = 2, XCDRv2 = 3 };
enum class EncodingFlag : uint8_t { PLAIN_CDR = 0x0, PL_CDR = 0x2, PLAIN_CDR2 = 0x4 };
enum class Endianness : uint8_t { CDR_BIG_ENDIAN = 0x00, CDR_LITTLE_ENDIAN = 0x01 };

struct CdrHeader {
  Endianness endianness = Endianness::CDR_LITTLE_ENDIAN;
  EncodingFlag encoding = EncodingFlag::PLAIN_CDR;
  CdrVersion version = CdrVersion::DDS_CDR;
};

inline bool hostIsLittleEndian() {
  const uint16_t probe = 1;
  uint8_t first;
  std::memcpy(&first, &probe, 1);
  return first == 1;
}

template <typename T>
inline T byteSwapped(T v) {
  uint8_t b[sizeof(T)];
  std::memcpy(b, &v, sizeof(T));
  for (size_t i = 0; i < sizeof(T) / 2; ++i) std::swap(b[i], b[sizeof(T) - 1 - i]);
  std::memcpy(&v, b, sizeof(T));
  return v;
}

class Decoder {
 public:
  explicit Decoder(ConstBuffer message, CdrVersion default_version = CdrVersion::DDS_CDR) : rest_(message) {
    if (message.size() < 4) throw std::runtime_error("CDR message shorter than its encapsulation header");
    const uint8_t* h = message.data();
    if (h[0] != 0) throw std::runtime_error("Invalid CDR header: expected first byte to be 0");
    header_.endianness = static_cast<Endianness>(h[1] & 0x1);
    header_.encoding = static_cast<EncodingFlag>(h[1] & 0xFE);
    header_.version = default_version;
    const bool xcdr = default_version >= CdrVersion::XCDRv1;
    switch (header_.encoding) {
      case EncodingFlag::PLAIN_CDR:
        if (xcdr) header_.version = CdrVersion::XCDRv1;
        break;
      case EncodingFlag::PL_CDR:
        if (!xcdr) throw std::runtime_error("Unexpected encoding received.");
        header_.version = CdrVersion::XCDRv1;
        break;
      case EncodingFlag::PLAIN_CDR2:
        if (!xcdr) throw std::runtime_error("Unexpected encoding received.");
        header_.version = CdrVersion::XCDRv2;
        break;
      default:
        throw std::runtime_error("Unexpected encoding received.");
    }
    if (h[2] != 0 || h[3] != 0) throw std::runtime_error("Extended header not supported");
    rest_.trim_front(4);
    origin_ = rest_.data();
    wide_align_ = header_.version == CdrVersion::XCDRv2 ? 4 : 8;
  }

  const CdrHeader& header() const { return header_; }
  ConstBuffer currentBuffer() const { return rest_; }
  void jump(size_t bytes) { rest_.trim_front(bytes); }

  template <typename T>
  void decode(T& out) {
    static_assert(std::is_arithmetic_v<T>, "nanocdr::Decoder::decode: arithmetic types only");
    if (sizeof(T) > 1) skipPadding(sizeof(T));
    if (rest_.size() < sizeof(T)) throw std::runtime_error("Decode: not enough data to decode");
    std::memcpy(&out, rest_.data(), sizeof(T));
    rest_.trim_front(sizeof(T));
    if (sizeof(T) > 1 && (header_.endianness == Endianness::CDR_LITTLE_ENDIAN) != hostIsLittleEndian())
      out = byteSwapped(out);
  }
  void decode(std::string& out) {
    uint32_t len = 0;
    decode(len);
    if (rest_.size() < len) throw std::runtime_error("Decode: not enough data to decode (string). Size: " + std::to_string(len));
    const char* chars = reinterpret_cast<const char*>(rest_.data());
    out.assign(chars, (len && chars[len - 1] == '\0') ? len - 1 : len);
    rest_.trim_front(len);
  }
  void decode(ConstBuffer& out) {  // sequence<uint8>: a view into the message, no copy
    uint32_t len = 0;
    decode(len);
    if (rest_.size() < len) throw std::runtime_error("Decode: not enough data to decode (bytes). Size: " + std::to_string(len));
    out = ConstBuffer(rest_.data(), len);
    rest_.trim_front(len);
  }

 private:
  void skipPadding(size_t size) {
    const size_t a = size == 8 ? wide_align_ : size;
    const size_t pos = static_cast<size_t>(rest_.data() - origin_);
    rest_.trim_front((a - pos % a) % a);
  }
  ConstBuffer rest_;
  const uint8_t* origin_ = nullptr;
  CdrHeader header_;
  size_t wide_align_ = 8;
};

class Encoder {
 public:
  Encoder(CdrHeader header, std::vector<uint8_t>& storage) : header_(header), out_(&storage) {
    out_->clear();
    out_->reserve(1024);
    out_->push_back(0);
    out_->push_back(static_cast<uint8_t>(static_cast<uint8_t>(header.endianness) | static_cast<uint8_t>(header.encoding)));
    out_->push_back(0);
    out_->push_back(0);
    wide_align_ = header_.version == CdrVersion::XCDRv2 ? 4 : 8;
  }
  explicit Encoder(CdrHeader header) : Encoder(header, own_) {}
  // continue a message whose first part (encapsulation header included) is already in `storage`
  Encoder(CdrHeader header, std::vector<uint8_t>& storage, bool append) : header_(header), out_(&storage) {
    if (!append || storage.size() < 4) throw std::runtime_error("nanocdr::Encoder: nothing to append to");
    wide_align_ = header_.version == CdrVersion::XCDRv2 ? 4 : 8;
  }

  const CdrHeader& header() const { return header_; }
  ConstBuffer encodedBuffer() const { return ConstBuffer(out_->data(), out_->size()); }

  template <typename T>
  void encode(const T& in) {
    static_assert(std::is_arithmetic_v<T>, "nanocdr::Encoder::encode: arithmetic types only");
    if (sizeof(T) > 1) {
      const size_t a = sizeof(T) == 8 ? wide_align_ : sizeof(T);
      const size_t pos = out_->size() - 4;
      out_->resize(out_->size() + (a - pos % a) % a);
    }
    T v = in;
    if (sizeof(T) > 1 && (header_.endianness == Endianness::CDR_LITTLE_ENDIAN) != hostIsLittleEndian()) v = byteSwapped(v);
    const size_t at = out_->size();
    out_->resize(at + sizeof(T));
    std::memcpy(out_->data() + at, &v, sizeof(T));
  }
  void encode(const std::string& in) {
    encode(static_cast<uint32_t>(in.size() + 1));
    out_->insert(out_->end(), in.begin(), in.end());
    out_->push_back(0);
  }
  void encode(const ConstBuffer& bytes) {
    encode(static_cast<uint32_t>(bytes.size()));
    out_->insert(out_->end(), bytes.data(), bytes.data() + bytes.size());
  }

 private:
  CdrHeader header_;
  std::vector<uint8_t>* out_;
  std::vector<uint8_t> own_;
  size_t wide_align_ = 8;
};

}  // namespace nanocdr
