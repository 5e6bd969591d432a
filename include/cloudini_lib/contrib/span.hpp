// Span<T>: non-owning (pointer, length) view used by every buffer-taking function of the Cloudini API.
// Drop-in for the reference's cloudini_lib/contrib/span.hpp: same operations, same throwing trim semantics
// (trim_front / trim_back throw std::runtime_error when asked to drop more than size()).
#pragma once

#include <array>
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <type_traits>
#include <vector>

template <typename T>
class Span {
  using Value = std::remove_const_t<T>;
  static constexpr bool kByteLike = sizeof(Value) == 1;
  static constexpr bool kConst = std::is_const_v<T>;

 public:
  Span() = default;
  Span(T* ptr, size_t count) : ptr_(ptr), count_(count) {}

  // untyped memory is accepted for byte-sized element types only
  Span(void* ptr, size_t count) : ptr_(static_cast<T*>(ptr)), count_(count) {
    static_assert(kByteLike, "Span(void*, n) needs a one-byte element type");
  }
  Span(const void* ptr, size_t count) : ptr_(static_cast<T*>(ptr)), count_(count) {
    static_assert(kByteLike && kConst, "Span(const void*, n) needs a const one-byte element type");
  }

  template <typename A, bool C = kConst, typename = std::enable_if_t<!C>>
  Span(std::vector<Value, A>& v) : ptr_(v.data()), count_(v.size()) {}
  template <typename A, bool C = kConst, typename = std::enable_if_t<C>>
  Span(const std::vector<Value, A>& v) : ptr_(v.data()), count_(v.size()) {}
  template <size_t N, bool C = kConst, typename = std::enable_if_t<!C>>
  Span(std::array<Value, N>& a) : ptr_(a.data()), count_(N) {}
  template <size_t N, bool C = kConst, typename = std::enable_if_t<C>>
  Span(const std::array<Value, N>& a) : ptr_(a.data()), count_(N) {}

  // a mutable view converts to a read-only one
  template <bool C = kConst, typename = std::enable_if_t<C>>
  Span(const Span<Value>& other) : ptr_(other.data()), count_(other.size()) {}

  T* data() const { return ptr_; }
  size_t size() const { return count_; }
  bool empty() const { return count_ == 0; }

  void trim_front(size_t n) {
    if (n > count_) throw std::runtime_error("Cannot trim more than the current size");
    ptr_ += n;
    count_ -= n;
  }
  void trim_back(size_t n) {
    if (n > count_) throw std::runtime_error("Cannot trim more than the current size");
    count_ -= n;
  }

 private:
  T* ptr_ = nullptr;
  size_t count_ = 0;
};
