// Buffer views and the byte-level primitives of the wire format (drop-in for cloudini_lib/encoding_utils.hpp).
// The varint routines are the host-side twins of the device code in cloudini_amd/csrc/stage1_math.h; the
// library itself only needs them for headers, framing and tests.
#pragma once

#include <cstring>
#include <stdexcept>
#include <string>

#include "cloudini_lib/basic_types.hpp"
#include "cloudini_lib/contrib/span.hpp"

namespace Cloudini {

using ConstBufferView = Span<const uint8_t>;
using BufferView = Span<uint8_t>;

template <typename T>
inline void encode(const T& value, BufferView& out) {
  if (out.size() < sizeof(T)) throw std::runtime_error("encode: not enough output buffer space");
  std::memcpy(out.data(), &value, sizeof(T));
  out.trim_front(sizeof(T));
}
template <>
inline void encode(const std::string& text, BufferView& out) {  // u16 length + bytes
  const uint16_t n = static_cast<uint16_t>(text.size());
  encode(n, out);
  if (out.size() < n) throw std::runtime_error("encode(string): not enough output buffer space");
  std::memcpy(out.data(), text.data(), n);
  out.trim_front(n);
}

template <typename T>
inline void decode(ConstBufferView& in, T& value) {
  if (in.size() < sizeof(T)) throw std::runtime_error("decode: not enough input data");
  std::memcpy(&value, in.data(), sizeof(T));
  in.trim_front(sizeof(T));
}
template <>
inline void decode(ConstBufferView& in, std::string& text) {
  uint16_t n = 0;
  decode(in, n);
  if (in.size() < n) throw std::runtime_error("decode(string): not enough input data");
  text.assign(reinterpret_cast<const char*>(in.data()), n);
  in.trim_front(n);
}

// zig-zag, plus one (0 is the NaN marker), little-endian base-128
inline size_t encodeVarint64(int64_t value, uint8_t* dst) {
  uint64_t u = ((static_cast<uint64_t>(value) << 1) ^ static_cast<uint64_t>(value >> 63)) + 1;
  size_t n = 0;
  for (; u > 0x7F; u >>= 7) dst[n++] = static_cast<uint8_t>(u | 0x80);
  dst[n++] = static_cast<uint8_t>(u);
  return n;
}

template <typename T>
inline int64_t ToInt64(const uint8_t* src) {
  T tmp;
  std::memcpy(&tmp, src, sizeof(T));
  return static_cast<int64_t>(tmp);
}

inline size_t decodeVarint(const uint8_t* src, size_t available, int64_t& value) {
  if (available == 0) throw std::runtime_error("decodeVarint: empty input");
  uint64_t u = 0;
  size_t used = 0;
  for (unsigned shift = 0;; shift += 7) {
    if (used >= available) throw std::runtime_error("decodeVarint: truncated input");
    const uint8_t byte = src[used++];
    const uint64_t bits = byte & 0x7F;
    if (shift >= 64 || (shift == 63 && bits > 1)) throw std::runtime_error("decodeVarint: value overflow");
    u |= bits << shift;
    if (!(byte & 0x80)) break;
    if (shift >= 63) throw std::runtime_error("decodeVarint: value overflow");
  }
  if (u == 0) throw std::runtime_error("decodeVarint: unexpected NaN marker");
  --u;
  value = static_cast<int64_t>((u >> 1) ^ (~(u & 1) + 1));
  return used;
}

}  // namespace Cloudini
