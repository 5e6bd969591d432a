// include/cloudini_amd/batch_transcoder.hpp: batched replacement of the reference's per-message converter loop
// (cloudini_lib/tools/src/mcap_converter.cpp:170-222).
#include "cloudini_amd/batch_transcoder.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <dirent.h>
#include <fstream>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <sys/stat.h>
#include <thread>

#include "host_internal.hpp"

namespace cloudini_amd {

namespace {

using Clock = std::chrono::steady_clock;
double since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// what makes two messages batchable: same fields (name-independent), step, options
std::string schemaKey(const Cloudini::EncodingInfo& info) {
  std::ostringstream k;
  k << int(info.version) << '/' << int(info.encoding_opt) << '/' << int(info.compression_opt) << '/' << info.point_step;
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

struct Parsed {
  cloudini_ros::RosPointCloud2 pc;
  Cloudini::EncodingInfo info;
  std::string key;
  uint64_t points = 0;
};

template <typename T>
class BoundedQueue {
 public:
  explicit BoundedQueue(size_t cap) : cap_(cap) {}
  void push(T v) {
    std::unique_lock<std::mutex> lock(m_);
    not_full_.wait(lock, [&] { return q_.size() < cap_ || closed_; });
    q_.push_back(std::move(v));
    not_empty_.notify_one();
  }
  bool pop(T& out) {
    std::unique_lock<std::mutex> lock(m_);
    not_empty_.wait(lock, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    out = std::move(q_.front());
    q_.pop_front();
    not_full_.notify_one();
    return true;
  }
  void close() {
    std::lock_guard<std::mutex> lock(m_);
    closed_ = true;
    not_empty_.notify_all();
    not_full_.notify_all();
  }

 private:
  std::mutex m_;
  std::condition_variable not_full_, not_empty_;
  std::deque<T> q_;
  size_t cap_;
  bool closed_ = false;
};

}  // namespace

// ---- directory I/O --------------------------------------------------------------------------------------------

DirectorySource::DirectorySource(const std::string& dir) : dir_(dir) {
  DIR* d = opendir(dir.c_str());
  if (!d) throw std::runtime_error("cannot open directory " + dir);
  while (dirent* e = readdir(d)) {
    const std::string name = e->d_name;
    if (name == "." || name == "..") continue;
    struct stat st;
    if (stat((dir + "/" + name).c_str(), &st) == 0 && S_ISREG(st.st_mode)) files_.push_back(name);
  }
  closedir(d);
  std::sort(files_.begin(), files_.end());
}

bool DirectorySource::next(Message& out) {
  if (at_ >= files_.size()) return false;
  out.name = files_[at_++];
  std::ifstream f(dir_ + "/" + out.name, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot read " + dir_ + "/" + out.name);
  const std::streamsize size = f.tellg();
  f.seekg(0);
  out.bytes.resize(static_cast<size_t>(size));
  if (size && !f.read(reinterpret_cast<char*>(out.bytes.data()), size)) throw std::runtime_error("short read: " + out.name);
  return true;
}

DirectorySink::DirectorySink(const std::string& dir) : dir_(dir) {
  struct stat st;
  if (stat(dir.c_str(), &st) != 0 && mkdir(dir.c_str(), 0755) != 0) throw std::runtime_error("cannot create directory " + dir);
}

void DirectorySink::write(const std::string& name, const uint8_t* data, size_t size) {
  std::ofstream f(dir_ + "/" + name, std::ios::binary | std::ios::trunc);
  if (!f || (size && !f.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(size))))
    throw std::runtime_error("cannot write " + dir_ + "/" + name);
}

// ---- one batch ------------------------------------------------------------------------------------------------

void transcodeBatch(const std::vector<Message>& in, const TranscodeOptions& opt, std::vector<std::vector<uint8_t>>& out,
                    TranscodeStats* stats) {
  const size_t n = in.size();
  out.assign(n, {});
  std::vector<Parsed> parsed(n);
  for (size_t i = 0; i < n; ++i) {  // the front of the reference's loop, message by message (mcap_converter.cpp:187-203)
    Parsed& p = parsed[i];
    p.pc = cloudini_ros::getDeserializedPointCloudMessage(Cloudini::ConstBufferView(in[i].bytes.data(), in[i].bytes.size()));
    cloudini_ros::applyResolutionProfile(opt.profile, p.pc.fields, opt.default_resolution);
    if (opt.viz_lossy) cloudini_ros::applyVizLossyPreprocessing(p.pc);
    p.info = cloudini_ros::toEncodingInfo(p.pc);
    p.info.compression_opt = opt.compression;
    p.points = p.info.point_step ? p.pc.data.size() / p.info.point_step : 0;
    p.key = schemaKey(p.info);
    if (stats) {
      stats->messages += 1;
      stats->points += p.points;
      stats->input_bytes += in[i].bytes.size();
    }
  }

  std::vector<uint8_t> stage1;
  std::vector<uint64_t> offsets;
  std::vector<uint32_t> chunk_sizes;
  for (size_t r0 = 0; r0 < n;) {  // runs of messages that share a schema: one GPU call each
    size_t r1 = r0 + 1;
    while (r1 < n && parsed[r1].key == parsed[r0].key) ++r1;
    const uint32_t m = static_cast<uint32_t>(r1 - r0);
    const Cloudini::EncodingInfo& info0 = parsed[r0].info;
    if (info0.point_step == 0) throw std::runtime_error("convertPointCloud2ToCompressedCloud: point_step cannot be 0");

    std::vector<const uint8_t*> ptrs(m);
    std::vector<uint64_t> pts(m);
    for (uint32_t k = 0; k < m; ++k) {
      ptrs[k] = parsed[r0 + k].pc.data.data();
      pts[k] = parsed[r0 + k].points;
    }
    const auto t_gpu = Clock::now();
    Cloudini::amd_detail::encodeStage1Batch(info0, ptrs.data(), pts.data(), m, stage1, offsets, chunk_sizes);
    if (stats) {
      stats->seconds_gpu += since(t_gpu);
      stats->gpu_batches += 1;
    }

    // stage 2: every chunk of the run is an independent job; each message owns a worst-case slot range of its own
    // output vector, so the jobs write in place and only a compaction inside the message remains
    const auto t_s2 = Clock::now();
    struct Job {
      uint32_t msg;        // index in the run
      const uint8_t* src;  // stage-1 payload
      uint32_t src_size;
      size_t slot;         // offset of the job's worst-case slot inside the message's scratch area
      uint32_t packed = 0;
    };
    std::vector<Job> jobs;
    std::vector<size_t> first_job(m + 1, 0), scratch_at(m), payload_at(m), length_at(m);
    std::vector<std::vector<uint8_t>> header(m);
    const bool direct = opt.compression == Cloudini::CompressionOption::NONE;
    size_t chunk_index = 0;
    for (uint32_t k = 0; k < m; ++k) {
      const Parsed& p = parsed[r0 + k];
      std::vector<uint8_t>& msg = out[r0 + k];
      nanocdr::Encoder enc(p.pc.cdr_header, msg);
      cloudini_ros::writePointCloudHeader(enc, p.pc);
      length_at[k] = msg.size();
      enc.encode(static_cast<uint32_t>(0));
      payload_at[k] = msg.size();
      first_job[k] = jobs.size();
      if (p.pc.data.size() == 0) continue;  // empty cloud: no Cloudini header either (ros_msg_utils.cpp:178-183)
      Cloudini::EncodeHeader(p.info, header[k]);
      size_t need = header[k].size();
      const uint64_t n_chunks = (p.points + 32767) / 32768;
      const uint8_t* s = stage1.data() + offsets[k];
      size_t slot = 0;
      for (uint64_t c = 0; c < n_chunks; ++c) {
        const uint32_t size = chunk_sizes[chunk_index + c];
        Job j;
        j.msg = k;
        j.src = s + 4;
        j.src_size = size;
        j.slot = slot;
        jobs.push_back(j);
        slot += 4 + Cloudini::amd_detail::compressedChunkBound(opt.compression, size);
        s += 4 + size;
      }
      chunk_index += n_chunks;
      need += slot;
      scratch_at[k] = payload_at[k] + header[k].size();
      msg.resize(payload_at[k] + need);
      std::memcpy(msg.data() + payload_at[k], header[k].data(), header[k].size());
    }
    first_job[m] = jobs.size();
    Cloudini::amd_detail::runOnStage2Pool(jobs.size(), [&](size_t ji) {
      Job& j = jobs[ji];
      uint8_t* dst = out[r0 + j.msg].data() + scratch_at[j.msg] + j.slot;
      if (direct) {
        std::memcpy(dst + 4, j.src, j.src_size);
        j.packed = j.src_size;
      } else {
        j.packed = Cloudini::amd_detail::compressChunkTo(opt.compression, j.src, j.src_size, dst + 4,
                                                         Cloudini::amd_detail::compressedChunkBound(opt.compression, j.src_size));
      }
      std::memcpy(dst, &j.packed, 4);
    });
    // close the gaps between a message's chunks, patch the length, finish the CDR message
    for (uint32_t k = 0; k < m; ++k) {
      const Parsed& p = parsed[r0 + k];
      std::vector<uint8_t>& msg = out[r0 + k];
      size_t end = payload_at[k];
      if (p.pc.data.size() != 0) {
        end = scratch_at[k];
        for (size_t ji = first_job[k]; ji < first_job[k + 1]; ++ji) {
          const Job& j = jobs[ji];
          uint8_t* from = msg.data() + scratch_at[k] + j.slot;
          if (from != msg.data() + end) std::memmove(msg.data() + end, from, 4u + j.packed);
          end += 4u + j.packed;
        }
      }
      const uint32_t encoded32 = static_cast<uint32_t>(end - payload_at[k]);
      std::memcpy(msg.data() + length_at[k], &encoded32, 4);
      msg.resize(end);
      nanocdr::Encoder tail(p.pc.cdr_header, msg, /*append=*/true);
      tail.encode(p.pc.is_dense);
      tail.encode(std::string("cloudini"));  // CompressedPointCloud2::format
      if (stats) stats->output_bytes += msg.size();
    }
    if (stats) stats->seconds_stage2 += since(t_s2);
    r0 = r1;
  }
}

// ---- the pipeline ---------------------------------------------------------------------------------------------

TranscodeStats transcodePointClouds(MessageSource& source, MessageSink& sink, const TranscodeOptions& opt) {
  TranscodeStats stats;
  const auto t0 = Clock::now();
  struct Batch {
    std::vector<Message> in;
    std::vector<std::vector<uint8_t>> out;
  };
  BoundedQueue<Batch> to_encode(2), to_write(2);
  std::exception_ptr reader_error, writer_error;
  const size_t batch = std::max<size_t>(1, opt.batch_messages);

  std::thread reader([&] {
    try {
      Batch b;
      Message msg;
      while (source.next(msg)) {
        b.in.push_back(std::move(msg));
        if (b.in.size() == batch) {
          to_encode.push(std::move(b));
          b = Batch();
        }
      }
      if (!b.in.empty()) to_encode.push(std::move(b));
    } catch (...) {
      reader_error = std::current_exception();
    }
    to_encode.close();
  });
  std::thread writer([&] {
    Batch b;
    while (to_write.pop(b)) {
      if (writer_error) continue;  // keep draining so the encoder never blocks
      try {
        for (size_t i = 0; i < b.in.size(); ++i) sink.write(b.in[i].name, b.out[i].data(), b.out[i].size());
      } catch (...) {
        writer_error = std::current_exception();
      }
    }
  });

  std::exception_ptr encode_error;
  Batch b;
  while (to_encode.pop(b)) {
    if (encode_error) continue;
    try {
      transcodeBatch(b.in, opt, b.out, &stats);
      to_write.push(std::move(b));
      b = Batch();
    } catch (...) {
      encode_error = std::current_exception();
    }
  }
  to_write.close();
  reader.join();
  writer.join();
  if (encode_error) std::rethrow_exception(encode_error);
  if (reader_error) std::rethrow_exception(reader_error);
  if (writer_error) std::rethrow_exception(writer_error);
  stats.seconds_total = since(t0);
  return stats;
}

}  // namespace cloudini_amd
