// Host side of the Cloudini API (include/cloudini_lib/cloudini.hpp): header text, chunk framing, stage 2
// (LZ4 / ZSTD) and the calls into the HIP stage-1 codec (include/cloudini_hip.h).
//
// Behaviour follows the reference's codec driver, cloudini_lib/src/cloudini.cpp (header :165-230 / :294-428,
// MaxCompressedSize :249-292, PointcloudEncoder::encode :501-623, PointcloudDecoder::decode :635-684) and
// src/chunk_writer.cpp:27-48; the byte layout of everything written here is checked against the compiled reference
// in tests/test_host_api.py. There is deliberately no CPU implementation of stage 1 in this file.
#include "cloudini_lib/cloudini.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <memory>
#include <functional>
#include <cstring>
#include <limits>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <thread>

#include "cloudini_hip.h"
#include "host_internal.hpp"
#include "yaml_lite.hpp"

// stage-2 libraries: only these entry points are used (prototypes instead of the vendor headers so that the build
// needs nothing but the shared objects)
extern "C" {
int LZ4_compress_default(const char* src, char* dst, int srcSize, int dstCapacity);
int LZ4_decompress_safe(const char* src, char* dst, int compressedSize, int dstCapacity);
int LZ4_compressBound(int inputSize);
size_t ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
size_t ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize);
size_t ZSTD_compressBound(size_t srcSize);
unsigned ZSTD_isError(size_t code);
const char* ZSTD_getErrorName(size_t code);
}

namespace Cloudini {

namespace {

constexpr size_t kPointsPerChunk = CLDN_HIP_POINTS_PER_CHUNK;
constexpr size_t kStagingKeepBytes = size_t(64) << 20;  // per-thread staging buffers larger than this are released after the call

[[noreturn]] void throwHip(const char* what) {
  throw std::runtime_error(std::string(what) + ": " + cldn_hip_last_error());
}

template <typename Enum>
Enum enumFromNumber(std::string_view text, int lo, int hi, const char* what) {
  int value = 0;
  try {
    value = std::stoi(std::string(text));
  } catch (const std::exception&) {
    throw std::runtime_error(std::string("Invalid ") + what + " string: " + std::string(text));
  }
  if (value < lo || value > hi) throw std::runtime_error(std::string("Invalid ") + what + " string: " + std::string(text));
  return static_cast<Enum>(value);
}

// detail::MaxSerializedPointSize (src/codec_common.cpp:29-67)
size_t maxSerializedPointSize(const EncodingInfo& info) {
  size_t total = 0;
  const bool lossy = info.encoding_opt == EncodingOptions::LOSSY;
  for (const auto& f : info.fields) {
    switch (f.type) {
      case FieldType::INT16: case FieldType::UINT16: case FieldType::INT32: case FieldType::UINT32:
      case FieldType::INT64: case FieldType::UINT64:
        total += 10;
        break;
      case FieldType::FLOAT32:
        total += (lossy && f.resolution) ? 10 : 7;
        break;
      case FieldType::FLOAT64:
        total += (lossy && f.resolution) ? 10 : 11;
        break;
      case FieldType::INT8: case FieldType::UINT8:
        total += 1;
        break;
      default:
        throw std::runtime_error("Unsupported field type '" + f.name + "' (type=" +
                                 std::to_string(static_cast<int>(f.type)) + ") in MaxSerializedFieldSize");
    }
  }
  return total;
}

bool isAdaptiveInt(FieldType t) {
  return t == FieldType::INT16 || t == FieldType::UINT16 || t == FieldType::INT32 || t == FieldType::UINT32 ||
         t == FieldType::INT64 || t == FieldType::UINT64;
}

// detail::UsesV5Codec (src/v5_codec.cpp:883-892)
bool usesV5(const EncodingInfo& info) {
  if (info.version < 5 || info.encoding_opt != EncodingOptions::LOSSY) return false;
  size_t lead = 0;
  for (const auto& f : info.fields) {
    if (f.type != FieldType::FLOAT32 || !f.resolution) break;
    ++lead;
  }
  if (lead != 3 && lead != 4) lead = 0;
  for (size_t i = lead; i < info.fields.size(); ++i)
    if (isAdaptiveInt(info.fields[i].type)) return true;
  return false;
}

struct PlanHandle {
  cldn_hip_plan_t* plan = nullptr;
  explicit PlanHandle(const EncodingInfo& info) {
    std::vector<cldn_hip_field_t> fields(info.fields.size());
    for (size_t i = 0; i < fields.size(); ++i) {
      fields[i].offset = info.fields[i].offset;
      fields[i].type = static_cast<uint8_t>(info.fields[i].type);
      fields[i].has_resolution = info.fields[i].resolution ? 1 : 0;
      fields[i].reserved[0] = fields[i].reserved[1] = 0;
      fields[i].resolution = info.fields[i].resolution.value_or(0.0f);
    }
    if (cldn_hip_plan_create(fields.data(), static_cast<uint32_t>(fields.size()), info.point_step, info.version,
                             static_cast<uint8_t>(info.encoding_opt), &plan) != CLDN_HIP_OK)
      throwHip("Cloudini (HIP) cannot handle this schema");
  }
  ~PlanHandle() { cldn_hip_plan_destroy(plan); }
  PlanHandle(const PlanHandle&) = delete;
  PlanHandle& operator=(const PlanHandle&) = delete;
};

// Codecs own device workspace; creating one per message (as the ROS plugin does with encoders,
// cloudini_ros/src/cloudini_publisher_plugin.cpp:53-55) would mean a hipMalloc storm, so idle codecs are pooled by schema.
struct CodecPool {
  struct Entry {
    std::string key;
    cldn_hip_codec_t* codec;
  };
  std::mutex mutex;
  std::vector<Entry> idle;

  static std::string keyOf(const EncodingInfo& info) {
    std::ostringstream k;
    k << int(info.version) << '/' << int(info.encoding_opt) << '/' << info.point_step;
    for (const auto& f : info.fields) {
      k << '|' << f.offset << ':' << int(f.type) << ':';
      if (f.resolution) {
        uint32_t bits;
        std::memcpy(&bits, &*f.resolution, 4);
        k << bits;
      } else {
        k << 'n';
      }
    }
    return k.str();
  }

  // a codec is bound to the device it was created on: the caller's current device is part of the key
  static std::string keyOn(int device, const EncodingInfo& info) { return std::to_string(device) + '@' + keyOf(info); }

  cldn_hip_codec_t* acquire(const EncodingInfo& info, const PlanHandle& plan) {
    const int device = cldn_hip_current_device();
    if (device < 0) throwHip("Cloudini (HIP) cannot create a codec");
    const std::string key = keyOn(device, info);
    {
      std::lock_guard<std::mutex> lock(mutex);
      for (size_t i = 0; i < idle.size(); ++i) {
        if (idle[i].key == key) {
          cldn_hip_codec_t* c = idle[i].codec;
          idle.erase(idle.begin() + static_cast<long>(i));
          return c;
        }
      }
    }
    cldn_hip_codec_t* c = nullptr;
    if (cldn_hip_codec_create(plan.plan, device, nullptr, &c) != CLDN_HIP_OK) throwHip("Cloudini (HIP) cannot create a codec");
    return c;
  }

  void release(const EncodingInfo& info, cldn_hip_codec_t* c) {
    if (!c) return;
    std::lock_guard<std::mutex> lock(mutex);
    if (idle.size() >= 16) {
      cldn_hip_codec_destroy(idle.front().codec);
      idle.erase(idle.begin());
    }
    idle.push_back({keyOn(cldn_hip_codec_device(c), info), c});
  }

  ~CodecPool() {
    // process teardown: the HIP runtime may already be gone; leak the handles on purpose
  }
};

CodecPool& pool() {
  static CodecPool* p = new CodecPool();
  return *p;
}

// Stage-2 workers: a small persistent pool (threads are created once per process, not per encode() call).
// run(n, fn) calls fn(i) for i in [0, n) on the pool plus the calling thread and returns when all are done; the first
// exception is rethrown on the caller. Every run() owns its Job object: a worker that wakes late still holds the job
// it was woken for (a shared_ptr), whose tickets are exhausted, so it can never take a ticket of a newer job or touch
// the caller's stack after run() has returned. Several run() calls may be in flight at once (queue).
//
// Size: the reference's `use_threads` means ONE stage-2 worker next to the encoding thread
// (cloudini_lib/src/cloudini.cpp:453-499). Here chunks are independent after the GPU call, so a few workers pay off,
// but a process with several publishers must not oversubscribe the host: the pool is bounded, default
// min(4, hardware threads) including the caller, changed with CLOUDINI_AMD_STAGE2_THREADS (1 = caller only) before
// the first encode/decode, or with Cloudini::amd_detail::setStage2Threads().
class WorkerPool {
 public:
  explicit WorkerPool(unsigned n_threads) { grow(n_threads); }
  ~WorkerPool() {
    // process teardown: worker threads are detached on purpose (joining at exit can deadlock in shared libraries)
    for (auto& t : threads_) t.detach();
  }
  unsigned workers() {
    std::lock_guard<std::mutex> lock(mutex_);
    return static_cast<unsigned>(threads_.size());
  }
  void grow(unsigned n_threads) {  // never shrinks: idle workers cost nothing but a parked thread
    std::lock_guard<std::mutex> lock(mutex_);
    while (threads_.size() < n_threads) threads_.emplace_back([this] { loop(); });
  }
  // at most `max_helpers` pool threads join the caller
  template <typename Fn>
  void run(size_t n, unsigned max_helpers, Fn&& fn) {
    if (n == 0) return;
    auto job = std::make_shared<Job>();
    job->fn = [&fn](size_t i) { fn(i); };
    job->n = n;
    job->pending = n;
    const size_t helpers = std::min<size_t>(max_helpers, n > 0 ? n - 1 : 0);
    if (helpers) {
      {
        std::lock_guard<std::mutex> lock(mutex_);
        for (size_t h = 0; h < helpers; ++h) queue_.push_back(job);
      }
      if (helpers == 1) wake_.notify_one();
      else wake_.notify_all();
    }
    drain(*job);
    {
      std::unique_lock<std::mutex> lock(job->mutex);
      job->done.wait(lock, [&] { return job->pending == 0; });
    }
    {  // helpers that have not started yet must not find the job later: its fn refers to this stack frame
      std::lock_guard<std::mutex> lock(mutex_);
      for (auto it = queue_.begin(); it != queue_.end();) it = (*it == job) ? queue_.erase(it) : it + 1;
    }
    if (job->error) std::rethrow_exception(job->error);
  }

  struct Job;
  // Asynchronous form of run(): at most `helpers` pool threads start on the job right away, the caller goes on with
  // something else (the next chunk group's GPU call) and joins in finish(). `fn` is copied into the job; whatever it
  // refers to must stay alive until finish() has returned.
  std::shared_ptr<Job> submit(size_t n, unsigned helpers, std::function<void(size_t)> fn) {
    auto job = std::make_shared<Job>();
    job->fn = std::move(fn);
    job->n = n;
    job->pending = n;
    const size_t h = std::min<size_t>(helpers, n);
    if (h) {
      {
        std::lock_guard<std::mutex> lock(mutex_);
        for (size_t k = 0; k < h; ++k) queue_.push_back(job);
      }
      if (h == 1) wake_.notify_one();
      else wake_.notify_all();
    }
    return job;
  }
  // the caller takes part in what is left of the job, waits for the rest, and gets the first error (if any) back
  std::exception_ptr finish(const std::shared_ptr<Job>& job) {
    if (job->n) {
      drain(*job);
      std::unique_lock<std::mutex> lock(job->mutex);
      job->done.wait(lock, [&] { return job->pending == 0; });
    }
    std::lock_guard<std::mutex> lock(mutex_);
    for (auto it = queue_.begin(); it != queue_.end();) it = (*it == job) ? queue_.erase(it) : it + 1;
    return job->error;
  }

 public:
  struct Job {
    std::function<void(size_t)> fn;
    size_t n = 0;
    std::atomic<size_t> next{0};
    std::mutex mutex;
    std::condition_variable done;
    size_t pending = 0;
    std::exception_ptr error;
  };

 private:
  static void drain(Job& job) {
    for (;;) {
      const size_t i = job.next.fetch_add(1);
      if (i >= job.n) return;  // tickets exhausted: nothing of the job is touched any more
      std::exception_ptr err;
      try {
        job.fn(i);
      } catch (...) {
        err = std::current_exception();
      }
      std::lock_guard<std::mutex> lock(job.mutex);
      if (err && !job.error) job.error = err;
      if (--job.pending == 0) job.done.notify_all();
    }
  }
  void loop() {
    for (;;) {
      std::shared_ptr<Job> job;
      {
        std::unique_lock<std::mutex> lock(mutex_);
        wake_.wait(lock, [&] { return !queue_.empty(); });
        job = queue_.front();
        queue_.pop_front();
      }
      // A ticket is only valid while pending > 0, and run() cannot return before pending == 0: fn is alive whenever a
      // ticket below n is drawn.
      drain(*job);
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mutex_;
  std::condition_variable wake_;
  std::deque<std::shared_ptr<Job>> queue_;
};

std::atomic<unsigned> g_stage2_threads{0};  // 0 = not decided yet

unsigned stage2Threads() {
  unsigned n = g_stage2_threads.load();
  if (n) return n;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  n = std::min(4u, hw);
  if (const char* e = std::getenv("CLOUDINI_AMD_STAGE2_THREADS")) {
    const long v = std::strtol(e, nullptr, 10);
    if (v >= 1) n = static_cast<unsigned>(std::min<long>(v, 256));
  }
  g_stage2_threads.store(n);
  return n;
}

WorkerPool& stage2Pool() {
  static WorkerPool* p = new WorkerPool(0);
  const unsigned want = stage2Threads() - 1u;  // the caller is one of the threads
  if (p->workers() < want) p->grow(want);
  return *p;
}

uint32_t compressChunk(CompressionOption opt, const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap) {
  switch (opt) {  // detail::CompressChunk, src/codec_common.cpp:220-258
    case CompressionOption::LZ4: {
      if (src_size > size_t(std::numeric_limits<int>::max()) || dst_cap > size_t(std::numeric_limits<int>::max()))
        throw std::runtime_error("Chunk size too large for LZ4");
      const int n = LZ4_compress_default(reinterpret_cast<const char*>(src), reinterpret_cast<char*>(dst),
                                         static_cast<int>(src_size), static_cast<int>(dst_cap));
      if (n <= 0) throw std::runtime_error("LZ4 compression failed");
      return static_cast<uint32_t>(n);
    }
    case CompressionOption::ZSTD: {
      const size_t n = ZSTD_compress(dst, dst_cap, src, src_size, 1);
      if (ZSTD_isError(n)) throw std::runtime_error("ZSTD compression failed");
      if (n > std::numeric_limits<uint32_t>::max()) throw std::runtime_error("Compressed chunk too large");
      return static_cast<uint32_t>(n);
    }
    default:
      throw std::runtime_error("Unsupported compression option");
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// names
// ---------------------------------------------------------------------------------------------------------------

const char* ToString(const FieldType& type) {
  static const char* names[] = {"UNKNOWN", "INT8",  "UINT8",   "INT16",   "UINT16", "INT32",
                                "UINT32",  "FLOAT32", "FLOAT64", "INT64", "UINT64"};
  const auto i = static_cast<size_t>(type);
  return i < sizeof(names) / sizeof(names[0]) ? names[i] : "UNKNOWN";
}
const char* ToString(const EncodingOptions& opt) {
  switch (opt) {
    case EncodingOptions::NONE: return "NONE";
    case EncodingOptions::LOSSY: return "LOSSY";
    case EncodingOptions::LOSSLESS: return "LOSSLESS";
  }
  return "UNKNOWN";
}
const char* ToString(const CompressionOption& opt) {
  switch (opt) {
    case CompressionOption::NONE: return "NONE";
    case CompressionOption::LZ4: return "LZ4";
    case CompressionOption::ZSTD: return "ZSTD";
  }
  return "UNKNOWN";
}

EncodingOptions EncodingOptionsFromString(std::string_view str) {
  for (int i = 0; i <= 2; ++i)
    if (str == ToString(static_cast<EncodingOptions>(i))) return static_cast<EncodingOptions>(i);
  return enumFromNumber<EncodingOptions>(str, 0, 2, "EncodingOptions");
}
CompressionOption CompressionOptionFromString(std::string_view str) {
  for (int i = 0; i <= 2; ++i)
    if (str == ToString(static_cast<CompressionOption>(i))) return static_cast<CompressionOption>(i);
  return enumFromNumber<CompressionOption>(str, 0, 2, "CompressionOption");
}
FieldType FieldTypeFromString(std::string_view str) {
  for (int i = 1; i <= 10; ++i)
    if (str == ToString(static_cast<FieldType>(i))) return static_cast<FieldType>(i);
  return enumFromNumber<FieldType>(str, 0, 10, "FieldType");
}

// ---------------------------------------------------------------------------------------------------------------
// header
// ---------------------------------------------------------------------------------------------------------------

std::string EncodingInfoToYAML(const EncodingInfo& info) {
  std::ostringstream y;  // default ostream float formatting (6 significant digits), as the reference prints it
  y << "version: " << int(info.version) << '\n'
    << "width: " << info.width << '\n'
    << "height: " << info.height << '\n'
    << "point_step: " << info.point_step << '\n'
    << "encoding_opt: " << ToString(info.encoding_opt) << '\n'
    << "compression_opt: " << ToString(info.compression_opt) << '\n';
  if (!info.encoding_config.empty()) y << "encoding_config: " << info.encoding_config << '\n';
  y << "fields:\n";
  for (const auto& f : info.fields) {
    y << "  - name: " << f.name << '\n' << "    offset: " << f.offset << '\n' << "    type: " << ToString(f.type) << '\n';
    if (f.resolution) y << "    resolution: " << *f.resolution << '\n';
    else y << "    resolution: null\n";
  }
  return y.str();
}

EncodingInfo EncodingInfoFromYAML(std::string_view yaml) {
  const yaml_lite::Document doc = yaml_lite::parse(yaml);
  // header text is untrusted input: every integer is range-checked before it is narrowed
  auto ranged = [](long long v, long long hi, const char* what) -> long long {
    if (v < 0 || v > hi) throw std::runtime_error(std::string("YAML: '") + what + "' out of range: " + std::to_string(v));
    return v;
  };
  constexpr long long kU32 = 0xffffffffll;
  EncodingInfo info;
  info.version = static_cast<uint8_t>(ranged(doc.integer("version"), 255, "version"));
  info.width = static_cast<uint32_t>(ranged(doc.integer("width"), kU32, "width"));
  info.height = static_cast<uint32_t>(ranged(doc.integer("height"), kU32, "height"));
  info.point_step = static_cast<uint32_t>(ranged(doc.integer("point_step"), kU32, "point_step"));
  info.encoding_opt = EncodingOptionsFromString(doc.scalar("encoding_opt"));
  info.compression_opt = CompressionOptionFromString(doc.scalar("compression_opt"));
  if (doc.has("encoding_config")) info.encoding_config = doc.scalar("encoding_config");
  for (const auto& item : doc.items) {
    PointField f;
    f.name = item.scalar("name");
    f.offset = static_cast<uint32_t>(ranged(item.integer("offset"), kU32, "offset"));
    f.type = FieldTypeFromString(item.scalar("type"));
    const std::string res = item.scalar("resolution");
    if (res != "null") {
      size_t used = 0;
      float value = 0.0f;
      try {
        value = std::stof(res, &used);
      } catch (const std::exception&) {
        throw std::runtime_error("YAML: 'resolution' is not a number: " + res);
      }
      if (used != res.size() || !(value > 0.0f) || !(value <= std::numeric_limits<float>::max()))
        throw std::runtime_error("YAML: 'resolution' must be a positive finite number: " + res);
      f.resolution = value;
    }
    info.fields.push_back(std::move(f));
  }
  return info;
}

void EncodeHeader(const EncodingInfo& header, std::vector<uint8_t>& output, HeaderEncoding encoding) {
  output.clear();
  output.insert(output.end(), kMagicHeader, kMagicHeader + kMagicHeaderLength);
  output.push_back(static_cast<uint8_t>('0' + header.version / 10));
  output.push_back(static_cast<uint8_t>('0' + header.version % 10));
  if (encoding == HeaderEncoding::YAML) {
    const std::string yaml = EncodingInfoToYAML(header);
    output.push_back('\n');
    output.insert(output.end(), yaml.begin(), yaml.end());
    output.push_back('\0');
    return;
  }
  auto put = [&output](const void* p, size_t n) {
    const auto* b = static_cast<const uint8_t*>(p);
    output.insert(output.end(), b, b + n);
  };
  put(&header.width, 4);
  put(&header.height, 4);
  put(&header.point_step, 4);
  output.push_back(static_cast<uint8_t>(header.encoding_opt));
  output.push_back(static_cast<uint8_t>(header.compression_opt));
  const uint16_t count = static_cast<uint16_t>(header.fields.size());
  put(&count, 2);
  for (const auto& f : header.fields) {
    const uint16_t len = static_cast<uint16_t>(f.name.size());
    put(&len, 2);
    put(f.name.data(), len);
    put(&f.offset, 4);
    output.push_back(static_cast<uint8_t>(f.type));
    const float res = f.resolution.value_or(-1.0f);
    put(&res, 4);
  }
}

EncodingInfo DecodeHeader(ConstBufferView& input) {
  if (input.size() < size_t(kMagicHeaderLength + 2)) throw std::runtime_error("Input too small to contain Cloudini header");
  if (std::memcmp(input.data(), kMagicHeader, kMagicHeaderLength) != 0) {
    throw std::runtime_error("Invalid magic header. Expected 'CLOUDINI_V', got: " +
                             std::string(reinterpret_cast<const char*>(input.data()), kMagicHeaderLength));
  }
  input.trim_front(kMagicHeaderLength);
  auto digit = [](uint8_t c) -> uint8_t { return (c >= '0' && c <= '9') ? uint8_t(c - '0') : uint8_t(0); };
  const uint8_t version = uint8_t(digit(input.data()[0]) * 10 + digit(input.data()[1]));
  input.trim_front(2);
  if (version < 2 || version > kEncodingVersion) {
    throw std::runtime_error("Unsupported encoding version. Current is:" + std::to_string(kEncodingVersion) +
                             ", got: " + std::to_string(version));
  }
  // YAML form: '\n' then text then '\0'. The legacy binary form never starts with "\n" followed by a non-brace.
  if (input.size() >= 2 && input.data()[0] == '\n' && input.data()[1] != '{') {
    input.trim_front(1);
    const auto* text = reinterpret_cast<const char*>(input.data());
    const void* nul = std::memchr(text, 0, input.size());
    if (!nul) throw std::runtime_error("Malformed YAML header: missing null terminator");
    const size_t len = static_cast<size_t>(static_cast<const char*>(nul) - text);
    EncodingInfo info = EncodingInfoFromYAML(std::string_view(text, len));
    input.trim_front(len + 1);
    info.version = version;  // the magic string is authoritative
    return info;
  }
  EncodingInfo info;
  info.version = version;
  decode(input, info.width);
  decode(input, info.height);
  decode(input, info.point_step);
  uint8_t stage = 0;
  decode(input, stage);
  info.encoding_opt = static_cast<EncodingOptions>(stage);
  decode(input, stage);
  info.compression_opt = static_cast<CompressionOption>(stage);
  uint16_t count = 0;
  decode(input, count);
  for (uint16_t i = 0; i < count; ++i) {
    PointField f;
    decode(input, f.name);
    decode(input, f.offset);
    uint8_t type = 0;
    decode(input, type);
    f.type = static_cast<FieldType>(type);
    float res = 0.0f;
    decode(input, res);
    if (res > 0) f.resolution = res;
    info.fields.push_back(std::move(f));
  }
  return info;
}

size_t MaxCompressedSize(const EncodingInfo& info, size_t points_count, bool include_header) {
  if (info.point_step == 0) throw std::runtime_error("point_step cannot be 0");
  const size_t per_point = maxSerializedPointSize(info);
  const bool v5 = usesV5(info);
  size_t total = include_header ? (kMagicHeaderLength + 2 + 1 + EncodingInfoToYAML(info).size() + 1) : 0;
  for (size_t left = points_count; left > 0;) {
    const size_t in_chunk = std::min(left, kPointsPerChunk);
    left -= in_chunk;
    size_t stage1 = in_chunk * per_point;
    if (v5) stage1 += info.fields.size() * 32u + 1024u;  // mode bytes / section headers of the adaptive sections
    total += sizeof(uint32_t);
    switch (info.compression_opt) {
      case CompressionOption::NONE:
        total += stage1;
        break;
      case CompressionOption::LZ4:
        if (stage1 > size_t(std::numeric_limits<int>::max())) throw std::runtime_error("Chunk size too large for LZ4");
        total += static_cast<size_t>(LZ4_compressBound(static_cast<int>(stage1)));
        break;
      case CompressionOption::ZSTD:
        total += ZSTD_compressBound(stage1);
        break;
      default:
        throw std::runtime_error("Unsupported compression option in MaxCompressedSize");
    }
  }
  return total;
}

// ---------------------------------------------------------------------------------------------------------------
// encoder
// ---------------------------------------------------------------------------------------------------------------

struct PointcloudEncoder::Impl {
  std::unique_ptr<PlanHandle> plan;
  cldn_hip_codec_t* codec = nullptr;
  std::vector<uint32_t> chunk_sizes;  // payload size per chunk
};

PointcloudEncoder::PointcloudEncoder(const EncodingInfo& info) : info_(info), impl_(new Impl()) {
  EncodeHeader(info_, header_);
  impl_->plan = std::make_unique<PlanHandle>(info_);
  impl_->codec = pool().acquire(info_, *impl_->plan);
}

PointcloudEncoder::~PointcloudEncoder() {
  if (impl_ && impl_->codec) pool().release(info_, impl_->codec);
}

// Chunk-group pipeline of one encode() call (the reference double-buffers chunk k's compression against chunk k+1's
// encoding, src/cloudini.cpp:453-499, :572-588): the cloud is cut into groups of whole chunks; while the stage-2 pool
// compresses the chunks of group g, the calling thread runs group g+1 through the GPU (H2D, kernels, D2H). The
// adaptive-int modes are a property of the cloud's first <= 4096 points (src/v5_codec.cpp:934-949): group 0 decides
// them, the later groups are encoded with those modes forced (cldn_hip_codec_force_modes), which yields exactly the
// chunks of a whole-cloud call.
// [u32 size][compressed chunk] in chunk order (src/chunk_writer.cpp:41-47) from the per-chunk scratch slots; the copies
// run on the stage-2 threads (6 MB through one core is 0.3-0.5 ms of a 1.2 ms call)
static size_t layOutChunks(const std::vector<uint32_t>& packed_size, const uint8_t* scratch, size_t slot, uint8_t* dst,
                           size_t dst_cap, unsigned workers) {
  const size_t n_chunks = packed_size.size();
  std::vector<size_t> at(n_chunks + 1, 0);
  for (size_t c = 0; c < n_chunks; ++c) at[c + 1] = at[c] + 4u + packed_size[c];
  if (at[n_chunks] > dst_cap) throw std::runtime_error("Output buffer too small for compressed chunk");
  auto place = [&](size_t c) {
    const uint32_t n = packed_size[c];
    std::memcpy(dst + at[c], &n, 4);
    std::memcpy(dst + at[c] + 4, scratch + c * slot, n);
  };
  if (workers > 1u && at[n_chunks] > (size_t(1) << 20)) stage2Pool().run(n_chunks, workers - 1u, place);
  else
    for (size_t c = 0; c < n_chunks; ++c) place(c);
  return at[n_chunks];
}

constexpr size_t kPipelineMinChunks = 8;
constexpr size_t kPipelineMaxGroups = 8;

// Number of chunk groups: CLOUDINI_AMD_PIPELINE=n (2..8; 0 or 1 = never); without it, 2 groups when the call has at
// least 8 stage-2 threads and none below. Measured on MI355X (one 1 M-point XYZI cloud, LZ4, median ms per encode()):
//   stage-2 threads   off    2 groups   3 groups   4 groups
//          4          2.46     2.67       2.50       2.93
//         16          1.40     1.17       1.27       1.63
// Stage 1 is 0.5 ms of a call that stage 2 dominates, and every extra synchronous GPU call costs 0.1-0.15 ms (launch-bound
// kernels, another round of copies): the earlier start of stage 2 only pays when there are threads to use it.
static size_t pipelineGroups(unsigned workers) {
  static const long env = [] {
    const char* e = std::getenv("CLOUDINI_AMD_PIPELINE");
    return e ? std::strtol(e, nullptr, 10) : -1L;
  }();
  if (env >= 0) return size_t(env < 2 ? 0 : std::min<long>(env, long(kPipelineMaxGroups)));
  return workers >= 8u ? 2u : 0u;
}

size_t PointcloudEncoder::encodePipelined(ConstBufferView cloud_data, uint64_t points, size_t n_chunks, unsigned workers,
                                          uint8_t* dst, size_t dst_cap, std::vector<uint8_t>& stage1,
                                          std::vector<uint8_t>& stage2) {
  const size_t n_groups = std::min(pipelineGroups(workers), n_chunks / (kPipelineMinChunks / 2));
  const uint32_t na = cldn_hip_plan_adaptive_fields(impl_->plan->plan);
  const size_t step = info_.point_step;
  // staging: every group's framed stream behind the previous one; a worst-case slot per chunk for stage 2
  std::vector<size_t> g_chunk0(n_groups + 1);
  uint64_t s1_need = 0;
  for (size_t g = 0; g <= n_groups; ++g) g_chunk0[g] = n_chunks * g / n_groups;
  for (size_t g = 0; g < n_groups; ++g) {
    const uint64_t p0 = uint64_t(g_chunk0[g]) * kPointsPerChunk, p1 = std::min<uint64_t>(points, uint64_t(g_chunk0[g + 1]) * kPointsPerChunk);
    s1_need += cldn_hip_stage1_bound(impl_->plan->plan, p1 - p0);
  }
  if (stage1.size() < s1_need) stage1.resize(s1_need);
  const size_t chunk_bound = size_t(cldn_hip_stage1_bound(impl_->plan->plan, kPointsPerChunk));
  const size_t slot = amd_detail::compressedChunkBound(info_.compression_opt, chunk_bound);
  if (stage2.size() < slot * n_chunks) stage2.resize(slot * n_chunks);
  uint8_t* s1 = stage1.data();
  uint8_t* scratch = stage2.data();
  std::vector<uint32_t> packed_size(n_chunks);
  std::vector<size_t> src_off(n_chunks);
  std::vector<uint8_t> modes(std::max<uint32_t>(1u, na));

  struct Guard {  // whatever happens: no job refers to this frame afterwards, and the codec probes again next time
    WorkerPool& pool;
    cldn_hip_codec_t* codec;
    std::vector<std::shared_ptr<WorkerPool::Job>> jobs;
    std::exception_ptr finishAll() {
      std::exception_ptr first;
      for (auto& j : jobs) {
        std::exception_ptr e = pool.finish(j);
        if (e && !first) first = e;
      }
      jobs.clear();
      return first;
    }
    ~Guard() {
      (void)finishAll();
      (void)cldn_hip_codec_force_modes(codec, nullptr, 0);
    }
  } guard{stage2Pool(), impl_->codec, {}};

  const CompressionOption opt = info_.compression_opt;
  const uint32_t* chunk_sizes = impl_->chunk_sizes.data();
  size_t s1_pos = 0;
  for (size_t g = 0; g < n_groups; ++g) {
    const size_t c0 = g_chunk0[g], c1 = g_chunk0[g + 1];
    const uint64_t p0 = uint64_t(c0) * kPointsPerChunk, p1 = std::min<uint64_t>(points, uint64_t(c1) * kPointsPerChunk);
    const uint64_t gp = p1 - p0;
    if (g == 1 && na && cldn_hip_codec_force_modes(impl_->codec, modes.data(), na) != CLDN_HIP_OK)
      throw std::runtime_error(cldn_hip_last_error());
    uint64_t offsets[2] = {0, 0};
    if (cldn_hip_encode_stage1(impl_->codec, cloud_data.data() + p0 * step, CLDN_HIP_HOST, &gp, 1, s1 + s1_pos,
                               stage1.size() - s1_pos, CLDN_HIP_HOST, offsets, impl_->chunk_sizes.data() + c0,
                               g == 0 ? modes.data() : nullptr) != CLDN_HIP_OK)
      throw std::runtime_error(cldn_hip_last_error());
    size_t pos = s1_pos;
    for (size_t c = c0; c < c1; ++c) {
      src_off[c] = pos + 4;
      pos += 4 + chunk_sizes[c];
    }
    s1_pos += size_t(offsets[1]);
    // stage 2 of this group starts now; the last group is taken up by this thread as well (finish below)
    guard.jobs.push_back(guard.pool.submit(c1 - c0, workers - 1u, [=, &packed_size, &src_off](size_t i) {
      const size_t c = c0 + i;
      packed_size[c] = compressChunk(opt, s1 + src_off[c], chunk_sizes[c], scratch + c * slot, slot);
    }));
  }
  if (std::exception_ptr e = guard.finishAll()) std::rethrow_exception(e);
  return layOutChunks(packed_size, scratch, slot, dst, dst_cap, workers);
}

size_t PointcloudEncoder::encode(ConstBufferView cloud_data, std::vector<uint8_t>& output) {
  if (info_.point_step == 0) throw std::runtime_error("point_step cannot be 0");
  if (cloud_data.size() % info_.point_step != 0)
    throw std::runtime_error("Input cloud_data size is not a multiple of point_step");
  output.resize(MaxCompressedSize(info_, cloud_data.size() / info_.point_step, true));
  BufferView view(output.data(), output.size());
  const size_t size = encode(cloud_data, view, true);
  output.resize(size);
  return size;
}

size_t PointcloudEncoder::encode(ConstBufferView cloud_data, BufferView& output, bool write_header) {
  if (info_.point_step == 0) throw std::runtime_error("point_step cannot be 0");
  if (cloud_data.size() % info_.point_step != 0)
    throw std::runtime_error("Input cloud_data size is not a multiple of point_step");
  const uint64_t points = cloud_data.size() / info_.point_step;
  const size_t required = MaxCompressedSize(info_, points, false) + (write_header ? header_.size() : 0);
  if (output.size() < required) throw std::runtime_error("Output buffer too small for worst-case compressed size");

  uint8_t* dst = output.data();
  size_t written = 0;
  if (write_header) {
    std::memcpy(dst, header_.data(), header_.size());
    written = header_.size();
  }
  if (points == 0) return written;

  // stage 1 on the GPU: framed stream [u32 size][payload] per 32768-point chunk
  const size_t n_chunks = (points + kPointsPerChunk - 1) / kPointsPerChunk;
  const uint64_t bound = cldn_hip_stage1_bound(impl_->plan->plan, points);
  const bool direct = info_.compression_opt == CompressionOption::NONE;  // the framed stream IS the payload
  uint8_t* s1 = dst + written;
  uint64_t s1_cap = output.size() - written;
  // Staging buffers live per thread, not per encoder: callers like the ROS publisher plugin build a fresh encoder for
  // every message (cloudini_publisher_plugin.cpp:53-55) and a 20 MB zero-filled vector per message costs more than
  // the whole encode.
  static thread_local std::vector<uint8_t> tl_stage1, tl_stage2;
  // ... but one huge cloud must not pin its staging memory to the thread for ever: buffers above the cap are given
  // back when this call is over
  struct Trim {
    std::vector<uint8_t>&a, &b;
    ~Trim() {
      if (a.capacity() > kStagingKeepBytes) std::vector<uint8_t>().swap(a);
      if (b.capacity() > kStagingKeepBytes) std::vector<uint8_t>().swap(b);
    }
  } trim{tl_stage1, tl_stage2};
  impl_->chunk_sizes.resize(n_chunks);
  if (info_.compression_opt == CompressionOption::LZ4 && amd_detail::deviceLz4()) {
    // stage 2 on the device as well: the codec writes [u32 size][LZ4 block] per chunk straight
    // into the caller's buffer. Valid LZ4 blocks (the reference's decoder reads them), not lz4's own bytes.
    struct Reset {
      cldn_hip_codec_t* codec;
      ~Reset() { (void)cldn_hip_codec_set_stage2(codec, CLDN_HIP_STAGE2_NONE); }  // the codec goes back to the pool
    } reset{impl_->codec};
    const int lz_mode = amd_detail::deviceLz4Level() >= 2 ? CLDN_HIP_STAGE2_LZ4_FAST : CLDN_HIP_STAGE2_LZ4;
    if (cldn_hip_codec_set_stage2(impl_->codec, lz_mode) != CLDN_HIP_OK) throw std::runtime_error(cldn_hip_last_error());
    uint64_t offsets[2] = {0, 0};
    if (cldn_hip_encode_stage1(impl_->codec, cloud_data.data(), CLDN_HIP_HOST, &points, 1, dst + written, output.size() - written,
                               CLDN_HIP_HOST, offsets, impl_->chunk_sizes.data(), nullptr) != CLDN_HIP_OK)
      throw std::runtime_error(cldn_hip_last_error());
    return written + static_cast<size_t>(offsets[1]);
  }
  if (!direct) {  // (sized only here: the device-LZ4 branch above never stages stage-1 bytes on the host)
    if (tl_stage1.size() < bound) tl_stage1.resize(bound);
    s1 = tl_stage1.data();
    s1_cap = tl_stage1.size();
  }
  const unsigned workers = (info_.use_threads && n_chunks > 1) ? std::min<unsigned>(stage2Threads(), unsigned(n_chunks)) : 1u;
  if (!direct && workers > 1u && n_chunks >= kPipelineMinChunks && pipelineGroups(workers) >= 2u)
    return written + encodePipelined(cloud_data, points, n_chunks, workers, dst + written, output.size() - written,
                                     tl_stage1, tl_stage2);
  uint64_t offsets[2] = {0, 0};
  if (cldn_hip_encode_stage1(impl_->codec, cloud_data.data(), CLDN_HIP_HOST, &points, 1, s1, s1_cap, CLDN_HIP_HOST,
                             offsets, impl_->chunk_sizes.data(), nullptr) != CLDN_HIP_OK)
    throw std::runtime_error(cldn_hip_last_error());
  if (direct) return written + static_cast<size_t>(offsets[1]);

  // stage 2 on the host: [u32 compressed size][LZ4 block | ZSTD frame] per chunk (src/chunk_writer.cpp:41-47).
  // Chunks are independent; with use_threads they are compressed concurrently and laid out in order afterwards.
  std::vector<size_t> src_off(n_chunks);
  {
    size_t pos = 0;
    for (size_t c = 0; c < n_chunks; ++c) {
      src_off[c] = pos + 4;
      pos += 4 + impl_->chunk_sizes[c];
    }
  }
  if (workers <= 1) {
    for (size_t c = 0; c < n_chunks; ++c) {
      uint8_t* size_ptr = dst + written;
      const size_t cap = output.size() - written - 4;
      const uint32_t n = compressChunk(info_.compression_opt, s1 + src_off[c], impl_->chunk_sizes[c], size_ptr + 4, cap);
      std::memcpy(size_ptr, &n, 4);
      written += 4 + n;
    }
    return written;
  }
  // every chunk into its own worst-case slot of one scratch buffer (reused across calls), then laid out in order
  const size_t max_in = *std::max_element(impl_->chunk_sizes.begin(), impl_->chunk_sizes.end());
  if (info_.compression_opt == CompressionOption::LZ4 && max_in > size_t(std::numeric_limits<int>::max()))
    throw std::runtime_error("Chunk size too large for LZ4");
  const size_t slot = info_.compression_opt == CompressionOption::LZ4 ? size_t(LZ4_compressBound(int(max_in)))
                                                                     : ZSTD_compressBound(max_in);
  if (tl_stage2.size() < slot * n_chunks) tl_stage2.resize(slot * n_chunks);
  std::vector<uint32_t> packed_size(n_chunks);
  uint8_t* scratch = tl_stage2.data();
  static const bool timing = std::getenv("CLDN_HOST_TIMING") != nullptr;  // diagnostics: stage-2 wall time to stderr
  const auto t0 = std::chrono::steady_clock::now();
  stage2Pool().run(n_chunks, workers - 1u, [&](size_t c) {
    packed_size[c] = compressChunk(info_.compression_opt, s1 + src_off[c], impl_->chunk_sizes[c], scratch + c * slot, slot);
  });
  if (timing)
    std::fprintf(stderr, "[cloudini_amd] stage 2: %zu chunks, %.3f ms on up to %u threads\n", n_chunks,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), workers);
  return written + layOutChunks(packed_size, scratch, slot, dst + written, output.size() - written, workers);
}

// ---------------------------------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------------------------------

struct PointcloudDecoder::Impl {
  std::string key;
  EncodingInfo info;
  std::unique_ptr<PlanHandle> plan;
  cldn_hip_codec_t* codec = nullptr;

  void release() {
    if (codec) pool().release(info, codec);
    codec = nullptr;
    plan.reset();
  }
};

PointcloudDecoder::PointcloudDecoder() : impl_(new Impl()) {}
PointcloudDecoder::~PointcloudDecoder() {
  if (impl_) impl_->release();
}

void PointcloudDecoder::decode(const EncodingInfo& info, ConstBufferView compressed_data, BufferView output) {
  decodeInto(info, compressed_data, output, false);
}

void PointcloudDecoder::decodeInto(const EncodingInfo& info, ConstBufferView compressed_data, BufferView output, bool output_is_zero) {
  if (compressed_data.size() >= size_t(kMagicHeaderLength) &&
      std::memcmp(compressed_data.data(), kMagicHeader, kMagicHeaderLength) == 0)
    throw std::runtime_error("compressed_data contains the header. You should use DecodeHeader first");
  if (info.point_step == 0) throw std::runtime_error("point_step cannot be 0");

  const std::string key = CodecPool::keyOf(info);
  if (!impl_->codec || key != impl_->key) {
    impl_->release();
    impl_->info = info;
    impl_->key = key;
    impl_->plan = std::make_unique<PlanHandle>(info);
    impl_->codec = pool().acquire(info, *impl_->plan);
  }

  const uint64_t points = static_cast<uint64_t>(info.width) * info.height;
  const uint64_t out_bytes = points * info.point_step;
  if (info.version < 3) {
    // wire version 2 (src/cloudini.cpp:665-667): the whole payload is one chunk -- stage 2 undone in one piece with
    // width * height * point_step as its bound (:672-675), then points until the payload is empty (src/v4_codec.cpp:108-115)
    const uint8_t* s1 = compressed_data.data();
    size_t s1_size = compressed_data.size();
    std::vector<uint8_t> plain;
    if (info.compression_opt != CompressionOption::NONE) {
      if (info.compression_opt == CompressionOption::LZ4 &&
          (compressed_data.size() > size_t(std::numeric_limits<int>::max()) || out_bytes > uint64_t(std::numeric_limits<int>::max())))
        throw std::runtime_error("Chunk size too large for LZ4");
      plain.resize(static_cast<size_t>(out_bytes));
      s1_size = amd_detail::decompressChunkTo(info.compression_opt, compressed_data.data(), compressed_data.size(), plain.data(),
                                              plain.size());
      s1 = plain.data();
    }
    if (cldn_hip_decode_stage1_unframed(impl_->codec, s1, s1_size, CLDN_HIP_HOST, output.data(), output.size(), CLDN_HIP_HOST) !=
        CLDN_HIP_OK)
      throw std::runtime_error(cldn_hip_last_error());
    return;
  }
  if (output.size() < out_bytes) throw std::runtime_error("Output buffer is too small to hold the decoded data");

  // validate the chunk chain the way the reference does (src/cloudini.cpp:645-664) and undo stage 2
  const bool direct = info.compression_opt == CompressionOption::NONE;
  const uint8_t* s1 = compressed_data.data();
  uint64_t s1_size = compressed_data.size();
  static thread_local std::vector<uint8_t> tl_stage1;  // per thread, not per decoder (see PointcloudEncoder::encode)
  struct Trim {
    std::vector<uint8_t>& a;
    ~Trim() {
      if (a.capacity() > kStagingKeepBytes) std::vector<uint8_t>().swap(a);
    }
  } trim{tl_stage1};
  {
    struct ChunkRef {
      const uint8_t* src;
      uint32_t size;
      uint64_t in_chunk;
    };
    std::vector<ChunkRef> refs;
    ConstBufferView rest = compressed_data;
    uint64_t remaining = points;
    while (!rest.empty()) {
      if (remaining == 0) throw std::runtime_error("Encoded data contains more chunks than declared points");
      uint32_t chunk_size = 0;
      Cloudini::decode(rest, chunk_size);
      if (chunk_size > rest.size()) throw std::runtime_error("Invalid chunk size found while decoding");
      const uint64_t in_chunk = std::min<uint64_t>(remaining, kPointsPerChunk);
      if (!direct) refs.push_back({rest.data(), chunk_size, in_chunk});
      rest.trim_front(chunk_size);
      remaining -= in_chunk;
    }
    if (remaining != 0) throw std::runtime_error("Encoded data ended before all declared points were decoded");
    if (!direct && !refs.empty()) {
      // undo stage 2: every chunk into its own worst-case slot (concurrently when use_threads), then packed into the
      // framed stage-1 stream the device decoder takes
      const size_t slot = static_cast<size_t>(cldn_hip_stage1_bound(impl_->plan->plan, kPointsPerChunk)) + 64;
      if (tl_stage1.size() < slot * refs.size()) tl_stage1.resize(slot * refs.size());
      uint8_t* scratch = tl_stage1.data();
      std::vector<uint32_t> got(refs.size());
      auto undo = [&](size_t c) {
        uint8_t* dst = scratch + c * slot + 4;
        const size_t cap = slot - 4;
        if (info.compression_opt == CompressionOption::LZ4) {
          const int n = LZ4_decompress_safe(reinterpret_cast<const char*>(refs[c].src), reinterpret_cast<char*>(dst),
                                            static_cast<int>(refs[c].size), static_cast<int>(cap));
          if (n < 0) throw std::runtime_error("LZ4 decompression failed");
          got[c] = static_cast<uint32_t>(n);
        } else {
          const size_t n = ZSTD_decompress(dst, cap, refs[c].src, refs[c].size);
          if (ZSTD_isError(n)) throw std::runtime_error(std::string("ZSTD decompression failed: ") + ZSTD_getErrorName(n));
          got[c] = static_cast<uint32_t>(n);
        }
      };
      if (info.use_threads && refs.size() > 1 && stage2Threads() > 1u) {
        stage2Pool().run(refs.size(), stage2Threads() - 1u, undo);
      } else {
        for (size_t c = 0; c < refs.size(); ++c) undo(c);
      }
      size_t produced = 0;
      for (size_t c = 0; c < refs.size(); ++c) {  // slot c starts at or behind `produced`: forward memmove is safe
        std::memcpy(scratch + c * slot, &got[c], 4);
        if (produced != c * slot) std::memmove(scratch + produced, scratch + c * slot, 4 + size_t(got[c]));
        produced += 4 + size_t(got[c]);
      }
      s1 = scratch;
      s1_size = produced;
    } else if (!direct) {
      s1_size = 0;
    }
  }
  if (points == 0) return;

  const uint64_t offsets[2] = {0, s1_size};
  // (the codec comes from a pool: the fill mode is set for every call)
  cldn_hip_codec_set_decode_fill(impl_->codec, output_is_zero ? CLDN_HIP_FILL_ZERO : CLDN_HIP_FILL_KEEP);
  const int rc = cldn_hip_decode_stage1(impl_->codec, s1, CLDN_HIP_HOST, offsets, &points, 1, output.data(), out_bytes, CLDN_HIP_HOST);
  cldn_hip_codec_set_decode_fill(impl_->codec, CLDN_HIP_FILL_KEEP);
  if (rc != CLDN_HIP_OK) throw std::runtime_error(cldn_hip_last_error());
}

namespace amd_detail {

namespace {
std::atomic<int> g_device_lz4{-1};  // -1: not decided yet (environment)
}
// 0 = the host pool (liblz4's own bytes), 1 = CLDN_HIP_STAGE2_LZ4, 2 = CLDN_HIP_STAGE2_LZ4_FAST (4 KiB windows: about 1.7 x the
// speed, blocks about 3 % larger)
int deviceLz4Level() {
  int v = g_device_lz4.load();
  if (v < 0) {
    const char* e = std::getenv("CLOUDINI_AMD_DEVICE_LZ4");
    const int x = e ? std::atoi(e) : 0;
    v = x >= 2 ? 2 : (x == 1 ? 1 : 0);
    g_device_lz4.store(v);
  }
  return v;
}
bool deviceLz4() { return deviceLz4Level() != 0; }
void setDeviceLz4(bool on) { g_device_lz4.store(on ? 1 : 0); }
void setDeviceLz4Level(int level) { g_device_lz4.store(level >= 2 ? 2 : (level == 1 ? 1 : 0)); }

void encodeStage1Batch(const EncodingInfo& info, const uint8_t* const* cloud_ptrs, const uint64_t* cloud_points,
                       uint32_t n_clouds, const std::function<uint8_t*(uint64_t)>& grow, std::vector<uint64_t>& stream_offsets,
                       std::vector<uint32_t>& chunk_sizes) {
  PlanHandle plan(info);
  uint64_t n_chunks = 0;
  for (uint32_t k = 0; k < n_clouds; ++k) n_chunks += (cloud_points[k] + kPointsPerChunk - 1) / kPointsPerChunk;
  stream_offsets.assign((size_t)n_clouds + 1, 0);
  chunk_sizes.assign((size_t)std::max<uint64_t>(1, n_chunks), 0);
  cldn_hip_codec_t* codec = pool().acquire(info, plan);
  // two-step host output: sizes first, then exactly the bytes that were produced
  int rc = cldn_hip_encode_stage1_gather(codec, reinterpret_cast<const void* const*>(cloud_ptrs), cloud_points, n_clouds, nullptr,
                                         0, CLDN_HIP_HOST, stream_offsets.data(), chunk_sizes.data(), nullptr);
  if (rc == CLDN_HIP_OK) {
    const uint64_t total = stream_offsets[n_clouds];
    rc = cldn_hip_codec_fetch_output(codec, total ? grow(total) : nullptr, total);
  }
  const std::string err = rc != CLDN_HIP_OK ? cldn_hip_last_error() : "";
  pool().release(info, codec);
  if (rc != CLDN_HIP_OK) throw std::runtime_error(err);
  chunk_sizes.resize((size_t)n_chunks);
}

uint32_t compressChunkTo(CompressionOption opt, const uint8_t* src, size_t src_size, uint8_t* dst, size_t dst_cap) {
  return compressChunk(opt, src, src_size, dst, dst_cap);
}

size_t compressedChunkBound(CompressionOption opt, size_t stage1_bytes) {
  switch (opt) {
    case CompressionOption::LZ4:
      if (stage1_bytes > size_t(std::numeric_limits<int>::max())) throw std::runtime_error("Chunk size too large for LZ4");
      return static_cast<size_t>(LZ4_compressBound(static_cast<int>(stage1_bytes)));
    case CompressionOption::ZSTD:
      return ZSTD_compressBound(stage1_bytes);
    default:
      return stage1_bytes;
  }
}

void walkCompressedChunks(ConstBufferView data, uint64_t points, std::vector<ChunkRef>& refs) {
  refs.clear();
  ConstBufferView rest = data;
  uint64_t remaining = points;
  while (!rest.empty()) {
    if (remaining == 0) throw std::runtime_error("Encoded data contains more chunks than declared points");
    uint32_t chunk_size = 0;
    Cloudini::decode(rest, chunk_size);
    if (chunk_size > rest.size()) throw std::runtime_error("Invalid chunk size found while decoding");
    refs.push_back({rest.data(), chunk_size});
    rest.trim_front(chunk_size);
    remaining -= std::min<uint64_t>(remaining, kPointsPerChunk);
  }
  if (remaining != 0) throw std::runtime_error("Encoded data ended before all declared points were decoded");
}

uint32_t decompressChunkTo(CompressionOption opt, const uint8_t* src, size_t size, uint8_t* dst, size_t dst_cap) {
  switch (opt) {
    case CompressionOption::LZ4: {
      if (size > size_t(std::numeric_limits<int>::max()) || dst_cap > size_t(std::numeric_limits<int>::max()))
        throw std::runtime_error("Chunk size too large for LZ4");
      const int n = LZ4_decompress_safe(reinterpret_cast<const char*>(src), reinterpret_cast<char*>(dst), static_cast<int>(size),
                                        static_cast<int>(dst_cap));
      if (n < 0) throw std::runtime_error("LZ4 decompression failed");
      return static_cast<uint32_t>(n);
    }
    case CompressionOption::ZSTD: {
      const size_t n = ZSTD_decompress(dst, dst_cap, src, size);
      if (ZSTD_isError(n)) throw std::runtime_error(std::string("ZSTD decompression failed: ") + ZSTD_getErrorName(n));
      return static_cast<uint32_t>(n);
    }
    default:
      if (size > dst_cap) throw std::runtime_error("Invalid chunk size found while decoding");
      std::memcpy(dst, src, size);
      return static_cast<uint32_t>(size);
  }
}

size_t stage1ChunkBound(const EncodingInfo& info) {
  PlanHandle plan(info);
  return static_cast<size_t>(cldn_hip_stage1_bound(plan.plan, kPointsPerChunk));
}

void decodeStage1Batch(const EncodingInfo& info, const uint8_t* streams, const uint64_t* offsets, const uint64_t* cloud_points,
                       uint32_t n_clouds, uint8_t* out, uint64_t out_capacity, bool out_is_zero) {
  if (info.point_step == 0) throw std::runtime_error("point_step cannot be 0");
  // framed streams only: a wire-version-2 payload has no [u32 size] prefixes (PointcloudDecoder::decode takes it through
  // cldn_hip_decode_stage1_unframed); its first bytes must never be read as a chunk size here
  if (info.version < 3) throw std::runtime_error("decodeStage1Batch: framed streams (wire version >= 3) only");
  PlanHandle plan(info);
  cldn_hip_codec_t* codec = pool().acquire(info, plan);
  cldn_hip_codec_set_decode_fill(codec, out_is_zero ? CLDN_HIP_FILL_ZERO : CLDN_HIP_FILL_KEEP);
  const int rc = cldn_hip_decode_stage1(codec, streams, CLDN_HIP_HOST, offsets, cloud_points, n_clouds, out, out_capacity,
                                        CLDN_HIP_HOST);
  cldn_hip_codec_set_decode_fill(codec, CLDN_HIP_FILL_KEEP);
  const std::string err = rc != CLDN_HIP_OK ? cldn_hip_last_error() : "";
  pool().release(info, codec);
  if (rc != CLDN_HIP_OK) throw std::runtime_error(err);
}

void runOnStage2Pool(size_t n, const std::function<void(size_t)>& fn) {
  const unsigned threads = Cloudini::stage2Threads();
  if (n <= 1 || threads <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  stage2Pool().run(n, threads - 1u, fn);
}

unsigned stage2Threads() { return Cloudini::stage2Threads(); }
void setStage2Threads(unsigned n) { g_stage2_threads.store(std::max(1u, std::min(n, 256u))); }

uint64_t vizPreprocessOnDevice(const uint8_t* points, size_t n_points, uint32_t point_step, uint32_t xyz_offset,
                               float resolution, uint8_t* out, size_t out_capacity) {
  // any pooled codec can lend its device, stream and workspace; this schema is only the pool key
  EncodingInfo ctx;
  ctx.fields.push_back(PointField{"x", 0, FieldType::FLOAT32, 0.001f});
  ctx.fields.push_back(PointField{"y", 4, FieldType::FLOAT32, 0.001f});
  ctx.fields.push_back(PointField{"z", 8, FieldType::FLOAT32, 0.001f});
  ctx.point_step = 12;
  PlanHandle plan(ctx);
  cldn_hip_codec_t* codec = pool().acquire(ctx, plan);
  uint64_t kept = 0;
  const int rc = cldn_hip_viz_preprocess(codec, points, CLDN_HIP_HOST, n_points, point_step, xyz_offset, resolution, out,
                                         out_capacity, CLDN_HIP_HOST, &kept);
  pool().release(ctx, codec);
  if (rc != CLDN_HIP_OK) throwHip("applyVizLossyPreprocessing (HIP) failed");
  return kept;
}

}  // namespace amd_detail

}  // namespace Cloudini
