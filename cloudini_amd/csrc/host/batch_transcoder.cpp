// include/cloudini_amd/batch_transcoder.hpp: batched replacement of the reference's per-message converter loop
// (cloudini_lib/tools/src/mcap_converter.cpp:170-222).
#include "cloudini_amd/batch_transcoder.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <cerrno>
#include <dirent.h>
#include <fcntl.h>
#include <unistd.h>
#include <fstream>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <sys/stat.h>
#include <thread>
#include <unordered_map>

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

// ---- page-locked blocks, kept across calls ------------------------------------------------------------------------
namespace {
struct PinnedCache {
  std::mutex mutex;
  std::unordered_map<void*, size_t> capacity;       // every live block (handed out or cached)
  std::multimap<size_t, void*> idle;                // cached blocks by capacity
  size_t idle_bytes = 0;
  static constexpr size_t kMaxIdleBytes = size_t(2) << 30;
};
PinnedCache& pinnedCache() {
  static PinnedCache* c = new PinnedCache();  // (never destroyed: blocks may be freed from static destructors)
  return *c;
}
}  // namespace

void releasePinnedCache() noexcept;

void* pinnedAlloc(size_t bytes) {
  if (bytes == 0) bytes = 1;
  PinnedCache& c = pinnedCache();
  {
    std::lock_guard<std::mutex> lock(c.mutex);
    auto it = c.idle.lower_bound(bytes);
    if (it != c.idle.end() && it->first <= 2 * bytes + 4096) {  // (a block much larger than asked for stays for a larger request)
      void* p = it->second;
      c.idle_bytes -= it->first;
      c.idle.erase(it);
      return p;
    }
  }
  static const bool no_device = cldn_hip_device_count() <= 0;  // (message buffers of a process without a GPU: ordinary memory)
  if (no_device) return std::malloc(bytes);
  void* p = cldn_hip_host_alloc(bytes);
  if (!p) {  // page-locked memory is short while idle blocks of other sizes sit in the cache: give them back, try once more
    releasePinnedCache();
    p = cldn_hip_host_alloc(bytes);
  }
  if (p) {
    std::lock_guard<std::mutex> lock(c.mutex);
    c.capacity[p] = bytes;
  }
  return p;
}

void pinnedFree(void* p) noexcept {
  if (!p) return;
  // decided per pointer, not per process: a block the cache knows is page-locked, anything else came from malloc (a process
  // without a GPU) -- two separately evaluated "is there a device" answers could disagree (runtime already torn down at exit)
  PinnedCache& c = pinnedCache();
  {
    std::lock_guard<std::mutex> lock(c.mutex);
    auto it = c.capacity.find(p);
    if (it == c.capacity.end()) {
      std::free(p);
      return;
    }
    if (c.idle_bytes + it->second <= PinnedCache::kMaxIdleBytes) {
      c.idle.emplace(it->second, p);
      c.idle_bytes += it->second;
      return;
    }
    c.capacity.erase(it);
  }
  cldn_hip_host_free(p);
}

void releasePinnedCache() noexcept {
  PinnedCache& c = pinnedCache();
  std::lock_guard<std::mutex> lock(c.mutex);
  for (auto& kv : c.idle) {
    c.capacity.erase(kv.second);
    cldn_hip_host_free(kv.second);
  }
  c.idle.clear();
  c.idle_bytes = 0;
}

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
  uint64_t ticket;
  if (!claim(ticket)) return false;
  fetch(ticket, out);
  return true;
}

bool DirectorySource::claim(uint64_t& ticket) {
  if (at_ >= files_.size()) return false;
  ticket = at_++;
  return true;
}

void DirectorySource::fetch(uint64_t ticket, Message& out) {
  out.name = files_[static_cast<size_t>(ticket)];
  const std::string path = dir_ + "/" + out.name;
  const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) throw std::runtime_error("cannot read " + path);
  struct stat st;
  if (::fstat(fd, &st) != 0 || st.st_size < 0) {
    ::close(fd);
    throw std::runtime_error("cannot size " + path);
  }
  out.bytes.resize(static_cast<size_t>(st.st_size));  // (PinnedAllocator::construct leaves the bytes alone)
  size_t got = 0;
  while (got < out.bytes.size()) {
    const ssize_t r = ::read(fd, out.bytes.data() + got, out.bytes.size() - got);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) break;
    got += static_cast<size_t>(r);
  }
  ::close(fd);
  if (got != out.bytes.size()) throw std::runtime_error("short read: " + out.name);
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

// ---- one batch: GPU phase, then stage-2 phase ---------------------------------------------------------------------

namespace {

struct Run {                       // messages [first, first + count) share a schema: one GPU call
  size_t first = 0;
  uint32_t count = 0;
  PinnedBytes stage1;              // framed stage-1 streams of the run, page-locked (the GPU copies into it)
  std::vector<uint64_t> offsets;   // count + 1
  std::vector<uint32_t> chunk_sizes;
  PinnedBytes decoded;             // decode direction: the points of the run's clouds, back to back
  std::vector<uint64_t> out_at;    // decode direction: where every message's points start in `decoded`
};

struct Batch {
  uint64_t seq = 0;                // position in the input (the writer restores this order)
  std::vector<Message> in;
  std::vector<Parsed> parsed;
  std::vector<Run> runs;
  std::vector<std::vector<uint8_t>> out;
};

// the front of the reference's loop message by message (mcap_converter.cpp:187-203), then one GPU call per schema run
void gpuPhase(Batch& b, const TranscodeOptions& opt, TranscodeStats* stats) {
  const size_t n = b.in.size();
  b.parsed.assign(n, Parsed());
  b.out.assign(n, {});
  for (size_t i = 0; i < n; ++i) {
    Parsed& p = b.parsed[i];
    p.pc = cloudini_ros::getDeserializedPointCloudMessage(Cloudini::ConstBufferView(b.in[i].bytes.data(), b.in[i].bytes.size()));
    cloudini_ros::applyResolutionProfile(opt.profile, p.pc.fields, opt.default_resolution);
    if (opt.viz_lossy) cloudini_ros::applyVizLossyPreprocessing(p.pc);
    p.info = cloudini_ros::toEncodingInfo(p.pc);
    p.info.compression_opt = opt.compression;
    // the reference's order, message by message (src/ros_msg_utils.cpp:178-190, src/cloudini.cpp:525-531): an empty
    // cloud becomes an empty message whatever its schema says; otherwise point_step 0 and a data size that is not a
    // multiple of point_step are errors (never a silently shortened cloud)
    p.points = 0;
    if (!p.pc.data.empty()) {
      if (p.info.point_step == 0) throw std::runtime_error("convertPointCloud2ToCompressedCloud: point_step cannot be 0");
      if (p.pc.data.size() % p.info.point_step != 0)
        throw std::runtime_error("Input cloud_data size is not a multiple of point_step");
      p.points = p.pc.data.size() / p.info.point_step;
    }
    p.key = p.pc.data.empty() ? std::string() : schemaKey(p.info);  // empty clouds: a run of their own, no GPU call
    if (stats) {
      stats->messages += 1;
      stats->points += p.points;
      stats->input_bytes += b.in[i].bytes.size();
    }
  }
  size_t n_runs = 0;
  for (size_t r0 = 0; r0 < n;) {
    size_t r1 = r0 + 1;
    while (r1 < n && b.parsed[r1].key == b.parsed[r0].key) ++r1;
    if (b.runs.size() <= n_runs) b.runs.emplace_back();  // recycled batches keep their page-locked staging
    Run& run = b.runs[n_runs++];
    run.first = r0;
    run.count = static_cast<uint32_t>(r1 - r0);
    const Cloudini::EncodingInfo& info0 = b.parsed[r0].info;
    if (b.parsed[r0].key.empty()) {  // a run of empty clouds: nothing to encode (their schema may not even be one)
      run.offsets.assign(run.count + 1, 0);
      run.chunk_sizes.clear();
      r0 = r1;
      continue;
    }
    std::vector<const uint8_t*> ptrs(run.count);
    std::vector<uint64_t> pts(run.count);
    for (uint32_t k = 0; k < run.count; ++k) {
      ptrs[k] = b.parsed[r0 + k].pc.data.data();
      pts[k] = b.parsed[r0 + k].points;
    }
    const auto t_gpu = Clock::now();
    Cloudini::amd_detail::encodeStage1Batch(
        info0, ptrs.data(), pts.data(), run.count,
        [&](uint64_t bytes) {  // the exact size, not the 50-bytes-per-point bound; recycled batches keep the capacity
          if (run.stage1.size() < bytes) run.stage1.resize(bytes + bytes / 8);
          return run.stage1.data();
        },
        run.offsets, run.chunk_sizes);
    if (stats) {
      stats->seconds_gpu += since(t_gpu);
      stats->gpu_batches += 1;
    }
    r0 = r1;
  }
  b.runs.resize(n_runs);
}

// stage 2 of every chunk of the batch on the host pool + CDR wrapping (src/ros_msg_utils.cpp:167-213)
void stage2Phase(Batch& b, const TranscodeOptions& opt, TranscodeStats* stats) {
  const auto t_s2 = Clock::now();
  const bool direct = opt.compression == Cloudini::CompressionOption::NONE;
  struct Job {
    size_t msg;          // index in the batch
    const uint8_t* src;  // stage-1 payload
    uint32_t src_size;
    size_t slot;         // offset of the job's worst-case slot inside the message's scratch area
    uint32_t packed = 0;
  };
  const size_t n = b.in.size();
  std::vector<Job> jobs;
  std::vector<size_t> first_job(n + 1, 0), scratch_at(n, 0), payload_at(n, 0), length_at(n, 0);
  std::vector<uint8_t> header;
  for (const Run& run : b.runs) {
    size_t chunk_index = 0;
    for (uint32_t k = 0; k < run.count; ++k) {
      const size_t i = run.first + k;
      const Parsed& p = b.parsed[i];
      std::vector<uint8_t>& msg = b.out[i];
      nanocdr::Encoder enc(p.pc.cdr_header, msg);
      cloudini_ros::writePointCloudHeader(enc, p.pc);
      length_at[i] = msg.size();  // patched once the encoded size is known
      enc.encode(static_cast<uint32_t>(0));
      payload_at[i] = msg.size();
      first_job[i] = jobs.size();
      if (p.pc.data.size() == 0) continue;  // empty cloud: no Cloudini header either (ros_msg_utils.cpp:178-183)
      Cloudini::EncodeHeader(p.info, header);
      const uint64_t n_chunks = (p.points + 32767) / 32768;
      const uint8_t* s = run.stage1.data() + run.offsets[k];
      size_t slot = 0;
      for (uint64_t c = 0; c < n_chunks; ++c) {
        const uint32_t size = run.chunk_sizes[chunk_index + c];
        Job j;
        j.msg = i;
        j.src = s + 4;
        j.src_size = size;
        j.slot = slot;
        jobs.push_back(j);
        slot += 4 + Cloudini::amd_detail::compressedChunkBound(opt.compression, size);
        s += 4 + size;
      }
      chunk_index += n_chunks;
      scratch_at[i] = payload_at[i] + header.size();
      msg.resize(scratch_at[i] + slot);
      std::memcpy(msg.data() + payload_at[i], header.data(), header.size());
    }
  }
  first_job[n] = jobs.size();  // messages were visited in batch order (runs are consecutive ranges): the table is monotone
  Cloudini::amd_detail::runOnStage2Pool(jobs.size(), [&](size_t ji) {
    Job& j = jobs[ji];
    uint8_t* dst = b.out[j.msg].data() + scratch_at[j.msg] + j.slot;
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
  for (size_t i = 0; i < n; ++i) {
    const Parsed& p = b.parsed[i];
    std::vector<uint8_t>& msg = b.out[i];
    size_t end = payload_at[i];
    if (p.pc.data.size() != 0) {
      end = scratch_at[i];
      for (size_t ji = first_job[i]; ji < first_job[i + 1]; ++ji) {
        const Job& j = jobs[ji];
        uint8_t* from = msg.data() + scratch_at[i] + j.slot;
        if (from != msg.data() + end) std::memmove(msg.data() + end, from, 4u + j.packed);
        end += 4u + j.packed;
      }
    }
    const uint32_t encoded32 = static_cast<uint32_t>(end - payload_at[i]);
    std::memcpy(msg.data() + length_at[i], &encoded32, 4);
    msg.resize(end);
    nanocdr::Encoder tail(p.pc.cdr_header, msg, /*append=*/true);
    tail.encode(p.pc.is_dense);
    tail.encode(std::string("cloudini"));  // CompressedPointCloud2::format
    if (stats) stats->output_bytes += msg.size();
  }
  if (stats) stats->seconds_stage2 += since(t_s2);
}

// width * height * point_step of a message, checked: a product that wraps, or a cloud beyond the u32 length field of
// the CDR byte sequence, is an error (the reference would try to allocate it, src/ros_msg_utils.cpp:150-153)
uint64_t messageCloudBytes(uint32_t width, uint32_t height, uint32_t point_step) {
  uint64_t bytes = 0;
  if (__builtin_mul_overflow(uint64_t(width) * height, uint64_t(point_step), &bytes) || bytes > 0xffffffffull)
    throw std::runtime_error("CompressedPointCloud2: width * height * point_step does not fit a PointCloud2 message");
  return bytes;
}

// ---- the way back: CompressedPointCloud2 -> PointCloud2 ---------------------------------------------------------------

// per message: CDR parse + Cloudini header; per schema run: stage 2 undone on the host pool, ONE batched GPU decode
void decodeGpuPhase(Batch& b, TranscodeStats* stats) {
  const size_t n = b.in.size();
  b.parsed.assign(n, Parsed());
  b.out.assign(n, {});
  std::vector<Cloudini::ConstBufferView> body(n);  // the chunks of every message, behind the Cloudini header
  for (size_t i = 0; i < n; ++i) {
    Parsed& p = b.parsed[i];
    p.pc = cloudini_ros::getDeserializedPointCloudMessage(Cloudini::ConstBufferView(b.in[i].bytes.data(), b.in[i].bytes.size()));
    // header values are not trusted: width * height * point_step must not wrap 64 bits, and the decoded cloud has to fit
    // the u32 length of the CDR byte sequence it goes into
    const uint64_t cloud_bytes = messageCloudBytes(p.pc.width, p.pc.height, p.pc.point_step);
    p.points = 0;
    p.key.clear();
    if (cloud_bytes != 0) {
      body[i] = p.pc.data;
      p.info = Cloudini::DecodeHeader(body[i]);
      const uint64_t pts = uint64_t(p.info.width) * p.info.height;
      uint64_t need = 0;
      // what the decoder would check (src/cloudini.cpp:632-635), against the size the message announces
      if (__builtin_mul_overflow(pts, uint64_t(p.info.point_step), &need) || need > cloud_bytes)
        throw std::runtime_error("Output buffer is too small to hold the decoded data");
      p.points = pts;
      // batchable only when the message's own geometry is the header's (always, for streams this library or the
      // reference wrote); anything else takes the single-message path in the wrap phase
      // (wire version 2 has no chunk framing -- one unframed payload, src/cloudini.cpp:665-667 -- and goes one by one too)
      if (pts * p.info.point_step == cloud_bytes && pts != 0 && p.info.version >= 3) p.key = schemaKey(p.info);
    }
    if (stats) {
      stats->messages += 1;
      stats->points += p.points;
      stats->input_bytes += b.in[i].bytes.size();
    }
  }
  size_t n_runs = 0;
  std::vector<Cloudini::amd_detail::ChunkRef> refs, all_refs;
  for (size_t r0 = 0; r0 < n;) {
    if (b.parsed[r0].key.empty()) {  // empty or odd message: no GPU work here
      ++r0;
      continue;
    }
    const Cloudini::EncodingInfo& info0 = b.parsed[r0].info;
    // stage 2 is undone chunk by chunk into worst-case slots, then packed into the framed stage-1 streams of the run; a run
    // ends where its slots would pass kDecodeScratchBytes (wide points: one chunk can be tens of megabytes)
    const size_t slot = 4u + Cloudini::amd_detail::stage1ChunkBound(info0) + 64u;
    constexpr size_t kDecodeScratchBytes = size_t(256) << 20;
    auto chunks_of = [&](size_t i) { return size_t((b.parsed[i].points + 32767) / 32768); };
    size_t r1 = r0 + 1, run_chunks = chunks_of(r0);
    while (r1 < n && b.parsed[r1].key == b.parsed[r0].key && (run_chunks + chunks_of(r1)) * slot <= kDecodeScratchBytes) {
      run_chunks += chunks_of(r1);
      ++r1;
    }
    if (b.runs.size() <= n_runs) b.runs.emplace_back();
    Run& run = b.runs[n_runs++];
    run.first = r0;
    run.count = static_cast<uint32_t>(r1 - r0);
    const auto t_gpu = Clock::now();
    all_refs.clear();
    std::vector<size_t> first_ref(run.count + 1, 0);
    for (uint32_t k = 0; k < run.count; ++k) {
      Cloudini::amd_detail::walkCompressedChunks(body[r0 + k], b.parsed[r0 + k].points, refs);
      first_ref[k] = all_refs.size();
      all_refs.insert(all_refs.end(), refs.begin(), refs.end());
    }
    first_ref[run.count] = all_refs.size();
    if (run.stage1.size() < slot * all_refs.size()) run.stage1.resize(slot * all_refs.size());
    uint8_t* scratch = run.stage1.data();
    std::vector<uint32_t> got(all_refs.size());
    const Cloudini::CompressionOption comp = info0.compression_opt;  // part of the schema key: the same for the whole run
    Cloudini::amd_detail::runOnStage2Pool(all_refs.size(), [&](size_t c) {
      got[c] = Cloudini::amd_detail::decompressChunkTo(comp, all_refs[c].src, all_refs[c].size, scratch + c * slot + 4, slot - 4);
    });
    run.offsets.assign(run.count + 1, 0);
    run.out_at.assign(run.count + 1, 0);
    std::vector<uint64_t> pts(run.count);
    size_t produced = 0;
    for (uint32_t k = 0; k < run.count; ++k) {
      run.offsets[k] = produced;
      for (size_t c = first_ref[k]; c < first_ref[k + 1]; ++c) {  // slot c starts at or behind `produced`
        std::memcpy(scratch + c * slot, &got[c], 4);
        if (produced != c * slot) std::memmove(scratch + produced, scratch + c * slot, 4u + size_t(got[c]));
        produced += 4u + size_t(got[c]);
      }
      pts[k] = b.parsed[r0 + k].points;
      run.out_at[k + 1] = run.out_at[k] + pts[k] * info0.point_step;
    }
    run.offsets[run.count] = produced;
    const uint64_t out_bytes = run.out_at[run.count];
    if (run.decoded.size() < out_bytes) run.decoded.resize(out_bytes);
    // bytes of a point that no field covers read zero, like in the freshly resized vector the reference decodes into
    // (src/ros_msg_utils.cpp:150-153): CLDN_HIP_FILL_ZERO -- the device buffer is cleared, every byte of `decoded` comes
    // back from it (no memset of 600 MB on this thread, no upload of zeros)
    Cloudini::amd_detail::decodeStage1Batch(info0, scratch, run.offsets.data(), pts.data(), run.count, run.decoded.data(),
                                            out_bytes, true);
    if (stats) {
      stats->seconds_gpu += since(t_gpu);
      stats->gpu_batches += 1;
    }
    r0 = r1;
  }
  b.runs.resize(n_runs);
}

// CDR wrapping of the decoded clouds (convertCompressedCloudToPointCloud2, src/ros_msg_utils.cpp:135-165)
void decodeWrapPhase(Batch& b, TranscodeStats* stats) {
  const auto t0 = Clock::now();
  const size_t n = b.in.size();
  std::vector<const uint8_t*> data_of(n, nullptr);
  for (const Run& run : b.runs)
    for (uint32_t k = 0; k < run.count; ++k) data_of[run.first + k] = run.decoded.data() + run.out_at[k];
  // (one message per job on the stage-2 pool: a 2.3 MB message is a resize and a copy, 0.25 ms on one thread)
  Cloudini::amd_detail::runOnStage2Pool(n, [&](size_t i) {
    const Parsed& p = b.parsed[i];
    std::vector<uint8_t>& msg = b.out[i];
    const size_t cloud_bytes = messageCloudBytes(p.pc.width, p.pc.height, p.pc.point_step);
    if (cloud_bytes != 0 && !data_of[i]) {  // geometry of the message and of its Cloudini header differ: one by one
      cloudini_ros::convertCompressedCloudToPointCloud2(p.pc, msg);
    } else {
      msg.clear();
      nanocdr::Encoder enc(p.pc.cdr_header, msg);
      cloudini_ros::writePointCloudHeader(enc, p.pc);
      enc.encode(static_cast<uint32_t>(cloud_bytes));
      if (cloud_bytes != 0) {
        const size_t at = msg.size();
        msg.resize(at + cloud_bytes);
        std::memcpy(msg.data() + at, data_of[i], cloud_bytes);
      }
      nanocdr::Encoder tail(p.pc.cdr_header, msg, /*append=*/true);
      tail.encode(p.pc.is_dense);
    }
  });
  if (stats) {
    for (size_t i = 0; i < n; ++i) stats->output_bytes += b.out[i].size();
    stats->seconds_stage2 += since(t0);
  }
}

}  // namespace

void transcodeBatch(const std::vector<Message>& in, const TranscodeOptions& opt, std::vector<std::vector<uint8_t>>& out,
                    TranscodeStats* stats) {
  Batch b;
  b.in.resize(in.size());
  for (size_t i = 0; i < in.size(); ++i) {
    b.in[i].name = in[i].name;
    b.in[i].bytes.assign(in[i].bytes.begin(), in[i].bytes.end());
  }
  if (opt.decode) {
    decodeGpuPhase(b, stats);
    decodeWrapPhase(b, stats);
  } else {
    gpuPhase(b, opt, stats);
    stage2Phase(b, opt, stats);
  }
  out = std::move(b.out);
}

// fn(i) for i in [0, n) on up to `threads` threads (the caller is one of them); the first exception is rethrown
template <typename Fn>
void parallelFor(size_t n, unsigned threads, Fn&& fn) {
  if (threads <= 1 || n <= 1) {
    for (size_t i = 0; i < n; ++i) fn(i);
    return;
  }
  std::atomic<size_t> next{0};
  std::exception_ptr error;
  std::mutex error_mutex;
  auto work = [&] {
    for (;;) {
      const size_t i = next.fetch_add(1);
      if (i >= n) return;
      try {
        fn(i);
      } catch (...) {
        std::lock_guard<std::mutex> lock(error_mutex);
        if (!error) error = std::current_exception();
        next.store(n);
      }
    }
  };
  std::vector<std::thread> team;
  for (unsigned t = 1; t < std::min<size_t>(threads, n); ++t) team.emplace_back(work);
  work();
  for (std::thread& t : team) t.join();
  if (error) std::rethrow_exception(error);
}

// ---- the pipeline: reader -> GPU (this thread) -> stage 2 -> writer; batches circulate (page-locked buffers are reused) ---

TranscodeStats transcodePointClouds(MessageSource& source, MessageSink& sink, const TranscodeOptions& opt) {
  TranscodeStats stats, stats2;
  const auto t0 = Clock::now();
  // GPU workers: one thread per entry of options.devices (the same device may appear more than once), each with its own
  // pooled codecs (the pool key carries the device, host/cloudini.cpp); batches go to whichever worker is free and are put
  // back into input order in front of the writer
  std::vector<int> devices = opt.devices;
  const bool fake = static_cast<bool>(opt.test_stage);  // test hook: the stages are the caller's function, no device is touched
  if (fake) {
    devices.assign(std::max<size_t>(1, opt.test_workers), 0);
  } else {
    if (devices.empty()) devices.push_back(-1);  // -1: the calling thread's current device
    const int caller_device = cldn_hip_current_device();
    for (int& d : devices)
      if (d < 0) d = caller_device;
    const int n_devices = cldn_hip_device_count();
    for (int d : devices)
      if (d < 0 || d >= n_devices) throw std::runtime_error("transcodePointClouds: device " + std::to_string(d) + " does not exist");
  }
  const size_t n_workers = devices.size();
  const size_t batches_in_flight = 2 * n_workers + 1;
  std::vector<std::unique_ptr<Batch>> storage;
  for (size_t i = 0; i < batches_in_flight; ++i) storage.emplace_back(new Batch());
  BoundedQueue<Batch*> free_q(batches_in_flight), to_gpu(batches_in_flight), to_stage2(batches_in_flight), to_write(batches_in_flight);
  for (auto& b : storage) free_q.push(b.get());
  // every error is written by one thread and rethrown behind the joins; `failed` is what the other threads look at: after
  // the first error nothing more is read, encoded or written (messages already written stay in the sink: the output of
  // a failed run is a prefix of the input)
  std::exception_ptr reader_error, stage2_error, writer_error;
  std::vector<std::exception_ptr> gpu_error(n_workers);
  std::atomic<bool> failed{false};
  const size_t batch = std::max<size_t>(1, opt.batch_messages);
  double seconds_read = 0, seconds_write = 0;  // each written by one thread (diagnostics, CLDN_HOST_TIMING)

  const unsigned io_threads = std::max(1u, opt.io_threads);
  std::thread reader([&] {
    try {
      uint64_t seq = 0;
      std::vector<uint64_t> tickets;
      for (;;) {
        Batch* b = nullptr;
        if (!free_q.pop(b)) break;
        if (failed.load()) {
          free_q.push(b);
          break;
        }
        size_t used = 0;
        bool exhausted = false;
        const auto t_read = Clock::now();
        if (source.concurrent() && io_threads > 1) {
          // tickets in input order by this thread, the messages themselves by the team
          tickets.clear();
          uint64_t t;
          while (tickets.size() < batch && source.claim(t)) tickets.push_back(t);
          used = tickets.size();
          exhausted = used < batch;
          if (b->in.size() < used) b->in.resize(used);
          parallelFor(used, io_threads, [&](size_t i) { source.fetch(tickets[i], b->in[i]); });
        } else {
          while (used < batch) {
            if (b->in.size() <= used) b->in.emplace_back();
            if (!source.next(b->in[used])) {  // the source refills the Message (and reuses its page-locked capacity)
              exhausted = !source.more();  // (more(): not the end -- hand on what there is, then ask again)
              break;
            }
            ++used;
            if (source.submitNow()) break;  // (the source will wait for the sink: what it delivered must be on its way)
          }
        }
        seconds_read += since(t_read);
        if (used == 0) {
          free_q.push(b);
          if (exhausted) break;
          continue;  // (the source only asked for what is in flight to be waited for)
        }
        b->in.resize(used);
        b->seq = seq++;
        to_gpu.push(b);
        if (exhausted) break;
      }
    } catch (...) {
      reader_error = std::current_exception();
      failed.store(true);
    }
    to_gpu.close();
  });
  std::mutex stats_mutex;
  std::atomic<size_t> gpu_workers_left{n_workers};
  std::vector<std::thread> gpu_workers;
  for (size_t w = 0; w < n_workers; ++w) {
    gpu_workers.emplace_back([&, w] {
      TranscodeStats mine;
      Batch* b = nullptr;
      bool device_set = false;
      while (to_gpu.pop(b)) {
        if (!failed.load()) {
          try {
            if (fake) {
              b->out.assign(b->in.size(), {});
              opt.test_stage(w, b->in, b->out);
              mine.messages += b->in.size();
              mine.gpu_batches += 1;
            } else {
              if (!device_set) {
                if (cldn_hip_set_current_device(devices[w]) != CLDN_HIP_OK) throw std::runtime_error(cldn_hip_last_error());
                device_set = true;
              }
              if (opt.decode) decodeGpuPhase(*b, &mine);
              else gpuPhase(*b, opt, &mine);
            }
          } catch (...) {
            gpu_error[w] = std::current_exception();
            failed.store(true);
          }
        }
        to_stage2.push(b);  // failed batches travel on (nobody touches them) so that they return to the reader
      }
      {
        std::lock_guard<std::mutex> lock(stats_mutex);
        stats.messages += mine.messages;
        stats.points += mine.points;
        stats.input_bytes += mine.input_bytes;
        stats.gpu_batches += mine.gpu_batches;
        stats.seconds_gpu += mine.seconds_gpu;
      }
      if (gpu_workers_left.fetch_sub(1) == 1) to_stage2.close();
    });
  }
  std::thread stage2([&] {
    Batch* b = nullptr;
    while (to_stage2.pop(b)) {
      if (!failed.load() && !fake) {
        try {
          if (opt.decode) decodeWrapPhase(*b, &stats2);
          else stage2Phase(*b, opt, &stats2);
        } catch (...) {
          stage2_error = std::current_exception();
          failed.store(true);
        }
      }
      to_write.push(b);
    }
    to_write.close();
  });
  std::thread writer([&] {
    Batch* b = nullptr;
    uint64_t next_seq = 0;
    std::vector<Batch*> parked;  // batches that overtook an earlier one on another GPU worker
    auto write_one = [&](Batch* x) {
      if (!failed.load()) {
        try {
          const auto t_write = Clock::now();
          parallelFor(x->in.size(), sink.concurrent() ? io_threads : 1u,
                      [&](size_t i) { sink.write(x->in[i].name, x->out[i].data(), x->out[i].size()); });
          seconds_write += since(t_write);
        } catch (...) {
          writer_error = std::current_exception();
          failed.store(true);
        }
      }
      ++next_seq;
      free_q.push(x);  // back to the reader
    };
    while (to_write.pop(b)) {
      if (failed.load()) {  // nothing more is written: hand everything back
        free_q.push(b);
        for (Batch* x : parked) free_q.push(x);
        parked.clear();
        continue;
      }
      parked.push_back(b);
      for (bool progress = true; progress;) {
        progress = false;
        for (size_t i = 0; i < parked.size(); ++i) {
          if (parked[i]->seq == next_seq) {
            Batch* x = parked[i];
            parked.erase(parked.begin() + i);
            write_one(x);
            progress = true;
            break;
          }
        }
      }
    }
    for (Batch* x : parked) free_q.push(x);
  });

  for (std::thread& t : gpu_workers) t.join();
  stage2.join();
  writer.join();
  free_q.close();
  reader.join();
  for (const std::exception_ptr& e : gpu_error)
    if (e) std::rethrow_exception(e);
  if (stage2_error) std::rethrow_exception(stage2_error);
  if (reader_error) std::rethrow_exception(reader_error);
  if (writer_error) std::rethrow_exception(writer_error);
  stats.output_bytes = stats2.output_bytes;
  stats.seconds_stage2 = stats2.seconds_stage2;
  stats.seconds_total = since(t0);
  stats.gpu_workers = n_workers;
  if (std::getenv("CLDN_HOST_TIMING"))
    std::fprintf(stderr, "[cloudini_amd] transcode: %.3f s total; source %.3f s, GPU stages %.3f s, stage 2 %.3f s, sink %.3f s\n",
                 stats.seconds_total, seconds_read, stats.seconds_gpu, stats.seconds_stage2, seconds_write);
  return stats;
}

}  // namespace cloudini_amd
