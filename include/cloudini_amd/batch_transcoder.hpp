// Batched PointCloud2 -> CompressedPointCloud2 transcoder on top of the batched HIP stage-1 encoder.
//
// What it replaces in the reference: the single-threaded per-message loop of the rosbag converter,
// cloudini_lib/tools/src/mcap_converter.cpp:170-222 (deserialise -> applyResolutionProfile -> optional
// applyVizLossyPreprocessing -> toEncodingInfo -> convertPointCloud2ToCompressedCloud -> write). Per message the output
// bytes are exactly what cloudini_ros::convertPointCloud2ToCompressedCloud (src/ros_msg_utils.cpp:167-213) produces; the
// difference is how the work is arranged:
//
//   reader thread   pulls messages from a MessageSource into batches (page-locked buffers, recycled)
//   GPU stage       one cldn_hip_encode_stage1_gather call per run of messages that share a schema: the clouds go from
//                   their message buffers straight to the device, the stage-1 streams come back into page-locked memory
//   stage-2 thread  LZ4 / ZSTD of ALL chunks of the batch on the bounded host pool, CDR wrapping
//   writer thread   hands the messages to a MessageSink in input order
// so the GPU works on batch k+1 while the host cores compress batch k and the sink writes batch k-1.
//
// Container I/O is pluggable (this image has no MCAP library): DirectorySource / DirectorySink read and write one CDR
// message per file; an MCAP reader / writer only has to implement the two interfaces.
#pragma once

#include <cstddef>
#include <cstdint>
#include <new>
#include <optional>
#include <string>
#include <vector>

#include "cloudini_hip.h"
#include "cloudini_lib/cloudini.hpp"
#include "cloudini_lib/ros_msg_utils.hpp"

namespace cloudini_amd {

// Page-locked host memory (cldn_hip_host_alloc): message buffers and staging areas the GPU copies from / to directly.
template <typename T>
struct PinnedAllocator {
  using value_type = T;
  PinnedAllocator() = default;
  template <typename U>
  PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(size_t n);
  void deallocate(T* p, size_t) noexcept;
  template <typename U>
  bool operator==(const PinnedAllocator<U>&) const { return true; }
  template <typename U>
  bool operator!=(const PinnedAllocator<U>&) const { return false; }
};
using PinnedBytes = std::vector<uint8_t, PinnedAllocator<uint8_t>>;

struct Message {
  std::string name;   // file name / channel + sequence: whatever identifies the message for the sink
  PinnedBytes bytes;  // CDR (DDS) bytes of a sensor_msgs/PointCloud2; sources reuse the capacity of the Message they are given
};

class MessageSource {
 public:
  virtual ~MessageSource() = default;
  virtual bool next(Message& out) = 0;  // false at the end
};

class MessageSink {
 public:
  virtual ~MessageSink() = default;
  virtual void write(const std::string& name, const uint8_t* data, size_t size) = 0;  // called in input order
};

// every regular file of a directory in lexicographic order / one file per message
class DirectorySource : public MessageSource {
 public:
  explicit DirectorySource(const std::string& dir);
  bool next(Message& out) override;

 private:
  std::vector<std::string> files_;
  std::string dir_;
  size_t at_ = 0;
};

class DirectorySink : public MessageSink {
 public:
  explicit DirectorySink(const std::string& dir);
  void write(const std::string& name, const uint8_t* data, size_t size) override;

 private:
  std::string dir_;
};

struct TranscodeOptions {
  cloudini_ros::ResolutionProfile profile;                      // applyResolutionProfile: field -> resolution, 0 removes
  std::optional<float> default_resolution = 0.001f;             // for FLOAT32 fields the profile does not name
  bool viz_lossy = false;                                       // applyVizLossyPreprocessing in front of the encoder
  Cloudini::CompressionOption compression = Cloudini::CompressionOption::ZSTD;  // toEncodingInfo's default
  size_t batch_messages = 32;                                   // messages per GPU batch
  // The way back (McapConverter::decodePointClouds, tools/src/mcap_converter.cpp:240-300): the messages are
  // CompressedPointCloud2, every output is the sensor_msgs/PointCloud2 that convertCompressedCloudToPointCloud2
  // (src/ros_msg_utils.cpp:135-165) writes. profile / default_resolution / viz_lossy / compression are not used.
  bool decode = false;
  // GPUs the batches are spread over: one GPU stage (thread + pooled codecs) per entry, shared reader, stage-2 pool and
  // ordered writer; batches go to whichever GPU stage is free. Empty = the calling thread's current device. The same
  // device may be listed more than once (two batches in flight on one GPU).
  std::vector<int> devices;
};

struct TranscodeStats {
  uint64_t messages = 0, points = 0, input_bytes = 0, output_bytes = 0, gpu_batches = 0;
  double seconds_total = 0, seconds_gpu = 0, seconds_stage2 = 0;  // seconds_gpu: summed over the GPU stages
  uint64_t gpu_workers = 0;
};

template <typename T>
T* PinnedAllocator<T>::allocate(size_t n) {
  void* p = cldn_hip_host_alloc(n * sizeof(T));
  if (!p) throw std::bad_alloc();
  return static_cast<T*>(p);
}
template <typename T>
void PinnedAllocator<T>::deallocate(T* p, size_t) noexcept {
  cldn_hip_host_free(p);
}

// One batch, in memory (also the unit the pipeline below runs): out[i] = CompressedPointCloud2 of in[i].
void transcodeBatch(const std::vector<Message>& in, const TranscodeOptions& options, std::vector<std::vector<uint8_t>>& out,
                    TranscodeStats* stats = nullptr);

// The whole pipeline: reader thread -> batches -> writer thread.
TranscodeStats transcodePointClouds(MessageSource& source, MessageSink& sink, const TranscodeOptions& options);

}  // namespace cloudini_amd
