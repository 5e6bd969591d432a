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
// Container I/O is pluggable: DirectorySource / DirectorySink read and write one CDR message per file; bags go through
// include/cloudini_amd/mcap_io.hpp (its own MCAP reader / writer: this image has no mcap library), which implements the two
// interfaces and copies every other message of the bag through.
#pragma once

#include <cstddef>
#include <cstdint>
#include <new>
#include <functional>
#include <optional>
#include <string>
#include <utility>
#include <vector>

#include "cloudini_hip.h"
#include "cloudini_lib/cloudini.hpp"
#include "cloudini_lib/ros_msg_utils.hpp"

namespace cloudini_amd {

// Page-locked host memory (cldn_hip_host_alloc): message buffers and staging areas the GPU copies from / to directly.
// Freed blocks are kept (up to 2 GiB per process) and handed out again: page-locking 370 MB of batch buffers costs a
// transcodePointClouds call 50 ms otherwise. releasePinnedCache() returns them to the driver.
void* pinnedAlloc(size_t bytes);
void pinnedFree(void* p) noexcept;
void releasePinnedCache() noexcept;

template <typename T>
struct PinnedAllocator {
  using value_type = T;
  PinnedAllocator() = default;
  template <typename U>
  PinnedAllocator(const PinnedAllocator<U>&) {}
  T* allocate(size_t n);
  void deallocate(T* p, size_t) noexcept;
  // resize() of a byte buffer that is about to be read into must not write it first (a 2.3 MB message would cross memory twice)
  template <typename U>
  void construct(U* p) noexcept {
    ::new (static_cast<void*>(p)) U;
  }
  template <typename U, typename... Args>
  void construct(U* p, Args&&... args) {
    ::new (static_cast<void*>(p)) U(std::forward<Args>(args)...);
  }
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
  // Optional: a source whose messages can be fetched independently of each other (files of a directory, records of an
  // indexed bag) lets the pipeline read a batch with several threads: claim() hands out the next message's ticket (one
  // caller at a time, input order), fetch() may then run concurrently for different tickets. next() == claim + fetch.
  virtual bool concurrent() const { return false; }
  virtual bool claim(uint64_t& ticket) {
    (void)ticket;
    return false;
  }
  virtual void fetch(uint64_t ticket, Message& out) {
    (void)ticket;
    (void)out;
  }
  // Optional: true right after a next() = "hand the batch on as it is" -- a source that holds other data back until the
  // messages it has delivered come out of the sink (the container converter: everything that lies between two point
  // clouds of a bag) asks for this before its next() waits for the sink.
  virtual bool submitNow() const { return false; }
  // Optional: asked after a next() that returned false. true = this is NOT the end: the source has read as much as it is
  // willing to hold back behind the messages it has delivered; the pipeline hands on what it has collected (also a batch
  // that is not full) and calls next() again, which may then wait for the sink.
  virtual bool more() const { return false; }
};

class MessageSink {
 public:
  virtual ~MessageSink() = default;
  virtual void write(const std::string& name, const uint8_t* data, size_t size) = 0;  // called in input order
  // Optional: a sink whose messages are independent (one file each) may be written by several threads at once; batches
  // still arrive in input order, the messages inside a batch in any order.
  virtual bool concurrent() const { return false; }
};

// every regular file of a directory in lexicographic order / one file per message
class DirectorySource : public MessageSource {
 public:
  explicit DirectorySource(const std::string& dir);
  bool next(Message& out) override;
  bool concurrent() const override { return true; }
  bool claim(uint64_t& ticket) override;
  void fetch(uint64_t ticket, Message& out) override;

 private:
  std::vector<std::string> files_;
  std::string dir_;
  size_t at_ = 0;
};

class DirectorySink : public MessageSink {
 public:
  explicit DirectorySink(const std::string& dir);
  void write(const std::string& name, const uint8_t* data, size_t size) override;
  bool concurrent() const override { return true; }

 private:
  std::string dir_;
};

struct TranscodeOptions {
  cloudini_ros::ResolutionProfile profile;                      // applyResolutionProfile: field -> resolution, 0 removes
  std::optional<float> default_resolution = 0.001f;             // for FLOAT32 fields the profile does not name
  bool viz_lossy = false;                                       // applyVizLossyPreprocessing in front of the encoder
  Cloudini::CompressionOption compression = Cloudini::CompressionOption::ZSTD;  // toEncodingInfo's default
  size_t batch_messages = 32;                                   // messages per GPU batch
  unsigned io_threads = 4;                                      // threads that read / write a batch of a concurrent() source / sink
  // The way back (McapConverter::decodePointClouds, tools/src/mcap_converter.cpp:240-300): the messages are
  // CompressedPointCloud2, every output is the sensor_msgs/PointCloud2 that convertCompressedCloudToPointCloud2
  // (src/ros_msg_utils.cpp:135-165) writes. profile / default_resolution / viz_lossy / compression are not used.
  bool decode = false;
  // GPUs the batches are spread over: one GPU stage (thread + pooled codecs) per entry, shared reader, stage-2 pool and
  // ordered writer; batches go to whichever GPU stage is free. Empty = the calling thread's current device. The same
  // device may be listed more than once (two batches in flight on one GPU).
  std::vector<int> devices;
  // Test hook (tests/cpp/transcoder_order.cpp, runs without a GPU): when set, `test_workers` stage threads call it instead of
  // the GPU stage and stage 2 passes the batch on untouched -- what remains is the pipeline itself: batches handed to
  // whichever stage is free, the writer putting them back into input order, an error on any stage stopping all of them.
  std::function<void(size_t worker, const std::vector<Message>& in, std::vector<std::vector<uint8_t>>& out)> test_stage;
  size_t test_workers = 0;
};

struct TranscodeStats {
  uint64_t messages = 0, points = 0, input_bytes = 0, output_bytes = 0, gpu_batches = 0;
  double seconds_total = 0, seconds_gpu = 0, seconds_stage2 = 0;  // seconds_gpu: summed over the GPU stages
  uint64_t gpu_workers = 0;
};

template <typename T>
T* PinnedAllocator<T>::allocate(size_t n) {
  void* p = pinnedAlloc(n * sizeof(T));
  if (!p) throw std::bad_alloc();
  return static_cast<T*>(p);
}
template <typename T>
void PinnedAllocator<T>::deallocate(T* p, size_t) noexcept {
  pinnedFree(p);
}

// One batch, in memory (also the unit the pipeline below runs): out[i] = CompressedPointCloud2 of in[i].
void transcodeBatch(const std::vector<Message>& in, const TranscodeOptions& options, std::vector<std::vector<uint8_t>>& out,
                    TranscodeStats* stats = nullptr);

// The whole pipeline: reader thread -> batches -> writer thread.
TranscodeStats transcodePointClouds(MessageSource& source, MessageSink& sink, const TranscodeOptions& options);

}  // namespace cloudini_amd
