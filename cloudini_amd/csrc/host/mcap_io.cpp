// MCAP container reader / writer and the bag converter on top of the batched transcoder (include/cloudini_amd/mcap_io.hpp).
// Record layouts follow the published specification (https://mcap.dev/spec): every record is [opcode u8][length u64][body],
// strings are [u32 length][bytes], maps are [u32 byte length]([string key][string value])*, all little endian.
#include "cloudini_amd/mcap_io.hpp"

#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <stdexcept>

extern "C" {
size_t ZSTD_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
size_t ZSTD_decompress(void* dst, size_t dstCapacity, const void* src, size_t compressedSize);
size_t ZSTD_compressBound(size_t srcSize);
unsigned ZSTD_isError(size_t code);
unsigned long long ZSTD_getFrameContentSize(const void* src, size_t srcSize);  // zstd.h: 0ULL - 1 = unknown, 0ULL - 2 = error
// LZ4 frame format (liblz4): MCAP's "lz4" chunks are frames, not blocks
size_t LZ4F_compressFrameBound(size_t srcSize, const void* preferences);
size_t LZ4F_compressFrame(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* preferences);
unsigned LZ4F_isError(size_t code);
size_t LZ4F_createDecompressionContext(void** ctx, unsigned version);
size_t LZ4F_freeDecompressionContext(void* ctx);
size_t LZ4F_decompress(void* ctx, void* dst, size_t* dstSize, const void* src, size_t* srcSize, const void* options);
}

namespace cloudini_amd {
namespace {
constexpr unsigned long long kZstdContentSizeUnknown = 0ULL - 1, kZstdContentSizeError = 0ULL - 2;
}

const char* const kPointCloud2SchemaName = "sensor_msgs/msg/PointCloud2";
const char* const kCompressedPointCloud2SchemaName = "point_cloud_interfaces/msg/CompressedPointCloud2";

#define CLDN_MSG_SEP "================================================================================\n"
#define CLDN_MSG_COMMON                                                                                              \
  CLDN_MSG_SEP "MSG: sensor_msgs/PointField\n"                                                                       \
               "uint8 INT8=1\nuint8 UINT8=2\nuint8 INT16=3\nuint8 UINT16=4\nuint8 INT32=5\nuint8 UINT32=6\n"         \
               "uint8 FLOAT32=7\nuint8 FLOAT64=8\nstring name\nuint32 offset\nuint8 datatype\nuint32 count\n"        \
  CLDN_MSG_SEP "MSG: std_msgs/Header\nbuiltin_interfaces/Time stamp\nstring frame_id\n"                              \
  CLDN_MSG_SEP "MSG: builtin_interfaces/Time\nint32 sec\nuint32 nanosec\n"
// (field lines of the ROS 2 definitions without their comments: the reference embeds the packages' .msg files verbatim)
const char* const kPointCloud2SchemaText =
    "std_msgs/Header header\nuint32 height\nuint32 width\nsensor_msgs/PointField[] fields\nbool is_bigendian\n"
    "uint32 point_step\nuint32 row_step\nuint8[] data\nbool is_dense\n" CLDN_MSG_COMMON;
const char* const kCompressedPointCloud2SchemaText =
    "std_msgs/Header header\nuint32 height\nuint32 width\nsensor_msgs/PointField[] fields\nbool is_bigendian\n"
    "uint32 point_step\nuint32 row_step\nuint8[] compressed_data\nbool is_dense\nstring format\n" CLDN_MSG_COMMON;

namespace {

constexpr uint8_t kMagic[8] = {0x89, 'M', 'C', 'A', 'P', 0x30, '\r', '\n'};
enum : uint8_t {
  OP_HEADER = 0x01, OP_FOOTER = 0x02, OP_SCHEMA = 0x03, OP_CHANNEL = 0x04, OP_MESSAGE = 0x05, OP_CHUNK = 0x06,
  OP_MESSAGE_INDEX = 0x07, OP_CHUNK_INDEX = 0x08, OP_ATTACHMENT = 0x09, OP_ATTACHMENT_INDEX = 0x0A, OP_STATISTICS = 0x0B,
  OP_METADATA = 0x0C, OP_METADATA_INDEX = 0x0D, OP_SUMMARY_OFFSET = 0x0E, OP_DATA_END = 0x0F
};

[[noreturn]] void bad(const std::string& what) { throw std::runtime_error("MCAP: " + what); }

// ---- little-endian cursor over a record body
struct Cur {
  const uint8_t* p;
  const uint8_t* end;
  void need(size_t n) const {
    if ((size_t)(end - p) < n) bad("record ends inside a field");
  }
  uint8_t u8() { need(1); return *p++; }
  uint16_t u16() { need(2); uint16_t v; std::memcpy(&v, p, 2); p += 2; return v; }
  uint32_t u32() { need(4); uint32_t v; std::memcpy(&v, p, 4); p += 4; return v; }
  uint64_t u64() { need(8); uint64_t v; std::memcpy(&v, p, 8); p += 8; return v; }
  std::string str() {
    const uint32_t n = u32();
    need(n);
    std::string s(reinterpret_cast<const char*>(p), n);
    p += n;
    return s;
  }
  std::vector<std::pair<std::string, std::string>> map() {
    const uint32_t bytes = u32();
    need(bytes);
    Cur m{p, p + bytes};
    std::vector<std::pair<std::string, std::string>> out;
    while (m.p < m.end) {
      std::string k = m.str();
      std::string v = m.str();
      out.emplace_back(std::move(k), std::move(v));
    }
    p += bytes;
    return out;
  }
};

// ---- record builder
struct Rec {
  std::vector<uint8_t> b;
  explicit Rec(uint8_t op) : b(9, 0) { b[0] = op; }
  void raw(const void* d, size_t n) { b.insert(b.end(), static_cast<const uint8_t*>(d), static_cast<const uint8_t*>(d) + n); }
  void u8(uint8_t v) { b.push_back(v); }
  void u16(uint16_t v) { raw(&v, 2); }
  void u32(uint32_t v) { raw(&v, 4); }
  void u64(uint64_t v) { raw(&v, 8); }
  void str(const std::string& s) { u32((uint32_t)s.size()); raw(s.data(), s.size()); }
  void map(const std::vector<std::pair<std::string, std::string>>& m) {
    size_t bytes = 0;
    for (const auto& kv : m) bytes += 8 + kv.first.size() + kv.second.size();
    u32((uint32_t)bytes);
    for (const auto& kv : m) { str(kv.first); str(kv.second); }
  }
  std::vector<uint8_t>& done() {
    const uint64_t len = b.size() - 9;
    std::memcpy(&b[1], &len, 8);
    return b;
  }
};

std::vector<uint8_t> readFile(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) bad("cannot open " + path);
  std::vector<uint8_t> v;
  if (std::fseek(f, 0, SEEK_END) == 0) {
    const long n = std::ftell(f);
    std::rewind(f);
    if (n > 0) {
      v.resize((size_t)n);
      if (std::fread(v.data(), 1, v.size(), f) != v.size()) { std::fclose(f); bad("short read of " + path); }
    }
  }
  std::fclose(f);
  return v;
}

std::vector<uint8_t> lz4FrameDecompress(const uint8_t* src, size_t n, size_t expect) {
  void* ctx = nullptr;
  if (LZ4F_isError(LZ4F_createDecompressionContext(&ctx, 100))) bad("LZ4F context");
  std::vector<uint8_t> out(expect);
  size_t done = 0, at = 0;
  while (at < n) {
    size_t dst = out.size() - done, srcn = n - at;
    const size_t r = LZ4F_decompress(ctx, out.data() + done, &dst, src + at, &srcn, nullptr);
    if (LZ4F_isError(r)) { LZ4F_freeDecompressionContext(ctx); bad("corrupt lz4 chunk"); }
    done += dst;
    at += srcn;
    if (r == 0) break;
    if (dst == 0 && srcn == 0) break;
  }
  LZ4F_freeDecompressionContext(ctx);
  if (done != expect) bad("lz4 chunk has another size than its record says");
  return out;
}

// the records of a compressed chunk. The record's own size claim is not trusted with an allocation: bounded, and checked
// against the compressed frame
std::vector<uint8_t> decompressChunk(const std::string& comp, const uint8_t* p, uint64_t n, uint64_t usize) {
  std::vector<uint8_t> raw;
  if (usize > kMcapMaxChunkBytes) bad("chunk claims " + std::to_string(usize) + " uncompressed bytes (limit " + std::to_string(kMcapMaxChunkBytes) + ")");
  if (comp == "zstd") {
    const unsigned long long fcs = ZSTD_getFrameContentSize(p, n);
    if (fcs == kZstdContentSizeError || (fcs != kZstdContentSizeUnknown && fcs != usize)) bad("corrupt zstd chunk");
    if (fcs == kZstdContentSizeUnknown && usize / 4096u > n + 64u) bad("corrupt zstd chunk");  // (no frame expands that far)
    raw.resize(usize);
    const size_t r = ZSTD_decompress(raw.data(), raw.size(), p, n);
    if (ZSTD_isError(r) || r != usize) bad("corrupt zstd chunk");
  } else if (comp == "lz4") {
    if (usize / 256u > n + 64u) bad("corrupt lz4 chunk");  // (LZ4 expands by at most 255 x)
    raw = lz4FrameDecompress(p, n, usize);
  } else {
    bad("chunk compression '" + comp + "' is not supported");
  }
  return raw;
}

}  // namespace

// -----------------------------------------------------------------------------------------------------------------
// reader
// -----------------------------------------------------------------------------------------------------------------
McapFile::McapFile(const std::string& path) : image_(readFile(path)) {
  if (image_.size() < 16 + 9 || std::memcmp(image_.data(), kMagic, 8) != 0) bad(path + " does not begin with the MCAP magic");
  if (std::memcmp(image_.data() + image_.size() - 8, kMagic, 8) != 0) bad(path + " does not end with the MCAP magic (truncated?)");
  try {
    parseRecords(image_.data() + 8, image_.data() + image_.size() - 8, false);
  } catch (const std::bad_alloc&) {
    bad("out of memory while reading " + path);
  } catch (const std::length_error&) {
    bad("out of memory while reading " + path);
  }
}

void McapFile::parseRecords(const uint8_t* p, const uint8_t* end, bool in_chunk) {
  while (p < end) {
    if ((size_t)(end - p) < 9) bad("record header cut short");
    const uint8_t op = p[0];
    uint64_t len;
    std::memcpy(&len, p + 1, 8);
    p += 9;
    if (len > (uint64_t)(end - p)) bad("record longer than what is left of the file");
    Cur c{p, p + len};
    p += len;
    switch (op) {
      case OP_HEADER:
        profile = c.str();
        library = c.str();
        break;
      case OP_SCHEMA: {
        McapSchema s;
        s.id = c.u16();
        s.name = c.str();
        s.encoding = c.str();
        const uint32_t n = c.u32();
        c.need(n);
        s.data.assign(c.p, c.p + n);
        if (s.id != 0) schemas[s.id] = std::move(s);  // (the summary section repeats them: same content)
        break;
      }
      case OP_CHANNEL: {
        McapChannel ch;
        ch.id = c.u16();
        ch.schema_id = c.u16();
        ch.topic = c.str();
        ch.message_encoding = c.str();
        ch.metadata = c.map();
        channels[ch.id] = std::move(ch);
        break;
      }
      case OP_MESSAGE: {
        McapMessage m;
        m.channel_id = c.u16();
        m.sequence = c.u32();
        m.log_time = c.u64();
        m.publish_time = c.u64();
        m.data = c.p;
        m.size = (size_t)(c.end - c.p);
        messages.push_back(m);
        break;
      }
      case OP_CHUNK: {
        if (in_chunk) bad("chunk inside a chunk");
        c.u64();  // message_start_time
        c.u64();  // message_end_time
        const uint64_t usize = c.u64();
        c.u32();  // uncompressed_crc (0 = not computed; not checked)
        const std::string comp = c.str();
        const uint64_t n = c.u64();
        c.need(n);
        if (comp.empty()) {
          if (n != usize) bad("uncompressed chunk whose sizes disagree");
          parseRecords(c.p, c.p + n, true);
        } else {
          std::vector<uint8_t> raw = decompressChunk(comp, c.p, n, usize);
          chunks_.push_back(std::move(raw));
          parseRecords(chunks_.back().data(), chunks_.back().data() + chunks_.back().size(), true);
        }
        break;
      }
      case OP_METADATA: {
        McapMetadata md;
        md.name = c.str();
        md.entries = c.map();
        metadata.push_back(std::move(md));
        break;
      }
      case OP_DATA_END:
        return;  // the summary section behind it repeats schemas / channels and holds indexes: not needed
      case OP_FOOTER:
        return;
      default:
        break;  // message / chunk / attachment indexes, attachments, statistics, summary offsets, unknown opcodes: skipped
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// streaming reader
// -----------------------------------------------------------------------------------------------------------------
McapStream::McapStream(const std::string& path) : path_(path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) bad("cannot open " + path);
  file_ = f;
  try {
    if (std::fseek(f, 0, SEEK_END) != 0) bad("cannot seek in " + path);
    const long n = std::ftell(f);
    if (n < 16 + 9) bad(path + " does not begin with the MCAP magic");
    size_ = (uint64_t)n;
    uint8_t magic[8];
    if (std::fseek(f, n - 8, SEEK_SET) != 0 || std::fread(magic, 1, 8, f) != 8) bad("short read of " + path);
    const bool tail_ok = std::memcmp(magic, kMagic, 8) == 0;
    std::rewind(f);
    readExact(magic, 8, "the magic");
    if (std::memcmp(magic, kMagic, 8) != 0) bad(path + " does not begin with the MCAP magic");
    if (!tail_ok) bad(path + " does not end with the MCAP magic (truncated?)");
    size_ -= 8;  // (records end in front of the closing magic)
    Record r;
    if (!next(r) || r.op != OP_HEADER) bad("the file does not begin with a Header record");
    Cur c{r.data, r.data + r.size};
    profile = c.str();
    library = c.str();
  } catch (...) {
    std::fclose(f);
    file_ = nullptr;
    throw;
  }
}

McapStream::~McapStream() {
  if (file_) std::fclose((FILE*)file_);
}

void McapStream::readExact(void* dst, size_t n, const char* what) {
  if (n != 0 && std::fread(dst, 1, n, (FILE*)file_) != n) bad(std::string("short read of ") + what);
  pos_ += n;
}

bool McapStream::next(Record& r) {
  try {
    for (;;) {
      if (ended_) return false;
      if (in_chunk_) {
        if (chunk_at_ == chunk_.size()) {
          in_chunk_ = false;
          continue;
        }
        if (chunk_.size() - chunk_at_ < 9) bad("record header cut short");
        const uint8_t op = chunk_[chunk_at_];
        uint64_t len;
        std::memcpy(&len, &chunk_[chunk_at_ + 1], 8);
        chunk_at_ += 9;
        if (len > chunk_.size() - chunk_at_) bad("record longer than what is left of the file");
        if (op == OP_CHUNK) bad("chunk inside a chunk");
        r.op = op;
        r.data = chunk_.data() + chunk_at_;
        r.size = (size_t)len;
        chunk_at_ += (size_t)len;
        if (op == OP_DATA_END || op == OP_FOOTER) {
          ended_ = true;
          return false;
        }
        return true;
      }
      if (pos_ >= size_) {  // (a file without DataEnd / Footer: McapFile reads such a file to its end too)
        ended_ = true;
        return false;
      }
      if (size_ - pos_ < 9) bad("record header cut short");
      uint8_t head[9];
      readExact(head, 9, "a record header");
      uint64_t len;
      std::memcpy(&len, head + 1, 8);
      if (len > size_ - pos_) bad("record longer than what is left of the file");
      if (head[0] == OP_DATA_END || head[0] == OP_FOOTER) {  // the summary section behind DataEnd is not needed
        ended_ = true;
        return false;
      }
      // (a record of the data section is at most a chunk: a length beyond the chunk limit and its frame's overhead is not
      // trusted with an allocation)
      if (len > kMcapMaxChunkBytes + (1ull << 20)) bad("record of " + std::to_string(len) + " bytes (limit " + std::to_string(kMcapMaxChunkBytes) + ")");
      rec_.resize((size_t)len);
      readExact(rec_.data(), (size_t)len, "a record");
      if (head[0] != OP_CHUNK) {
        r.op = head[0];
        r.data = rec_.data();
        r.size = (size_t)len;
        return true;
      }
      Cur c{rec_.data(), rec_.data() + rec_.size()};
      c.u64();  // message_start_time
      c.u64();  // message_end_time
      const uint64_t usize = c.u64();
      c.u32();  // uncompressed_crc (0 = not computed; not checked)
      const std::string comp = c.str();
      const uint64_t n = c.u64();
      c.need(n);
      if (comp.empty()) {
        if (n != usize) bad("uncompressed chunk whose sizes disagree");
        chunk_.assign(c.p, c.p + n);
      } else {
        chunk_ = decompressChunk(comp, c.p, n, usize);
      }
      chunk_at_ = 0;
      in_chunk_ = true;
    }
  } catch (const std::bad_alloc&) {
    bad("out of memory while reading " + path_);
  } catch (const std::length_error&) {
    bad("out of memory while reading " + path_);
  }
}

McapSchema McapStream::parseSchema(const Record& r) {
  Cur c{r.data, r.data + r.size};
  McapSchema s;
  s.id = c.u16();
  s.name = c.str();
  s.encoding = c.str();
  const uint32_t n = c.u32();
  c.need(n);
  s.data.assign(c.p, c.p + n);
  return s;
}
McapChannel McapStream::parseChannel(const Record& r) {
  Cur c{r.data, r.data + r.size};
  McapChannel ch;
  ch.id = c.u16();
  ch.schema_id = c.u16();
  ch.topic = c.str();
  ch.message_encoding = c.str();
  ch.metadata = c.map();
  return ch;
}
McapMetadata McapStream::parseMetadata(const Record& r) {
  Cur c{r.data, r.data + r.size};
  McapMetadata md;
  md.name = c.str();
  md.entries = c.map();
  return md;
}
McapMessage McapStream::parseMessage(const Record& r) {
  Cur c{r.data, r.data + r.size};
  McapMessage m;
  m.channel_id = c.u16();
  m.sequence = c.u32();
  m.log_time = c.u64();
  m.publish_time = c.u64();
  m.data = c.p;
  m.size = (size_t)(c.end - c.p);
  return m;
}

// -----------------------------------------------------------------------------------------------------------------
// writer
// -----------------------------------------------------------------------------------------------------------------
McapWriter::McapWriter(const std::string& path, const std::string& profile, McapCompression compression, size_t chunk_size)
    : path_(path), tmp_path_(path + ".partial"), compression_(compression), chunk_size_(chunk_size ? chunk_size : 1) {
  FILE* f = std::fopen(tmp_path_.c_str(), "wb");
  if (!f) bad("cannot create " + tmp_path_);
  file_ = f;
  try {
    if (std::fwrite(kMagic, 1, 8, f) != 8) bad("write failed");
    pos_ = 8;
    Rec h(OP_HEADER);
    h.str(profile);
    h.str("cloudini_amd");
    put(h.done());
  } catch (...) {
    std::fclose(f);
    file_ = nullptr;
    std::remove(tmp_path_.c_str());
    throw;
  }
}

// Only close() finalizes. A writer that is destroyed without it (an exception on the way) leaves nothing behind: a bag
// with a footer would look complete while it misses every message behind the failure.
McapWriter::~McapWriter() {
  if (file_) {
    std::fclose(static_cast<FILE*>(file_));
    file_ = nullptr;
    std::remove(tmp_path_.c_str());
  }
}

void McapWriter::put(const std::vector<uint8_t>& r) {
  if (std::fwrite(r.data(), 1, r.size(), static_cast<FILE*>(file_)) != r.size()) bad("write failed");
  pos_ += r.size();
}

void McapWriter::addSchema(const McapSchema& s) {
  Rec r(OP_SCHEMA);
  r.u16(s.id);
  r.str(s.name);
  r.str(s.encoding);
  r.u32((uint32_t)s.data.size());
  r.raw(s.data.data(), s.data.size());
  const auto& rec = r.done();
  chunk_.insert(chunk_.end(), rec.begin(), rec.end());  // schemas and channels travel inside the chunks, ahead of their messages
  schema_records_.push_back(rec);
}

void McapWriter::addChannel(const McapChannel& c) {
  Rec r(OP_CHANNEL);
  r.u16(c.id);
  r.u16(c.schema_id);
  r.str(c.topic);
  r.str(c.message_encoding);
  r.map(c.metadata);
  const auto& rec = r.done();
  chunk_.insert(chunk_.end(), rec.begin(), rec.end());
  channel_records_.push_back(rec);
  channel_counts_[c.id] += 0;
}

void McapWriter::addMetadata(const McapMetadata& m) {
  flushChunk();  // metadata records live outside chunks
  Rec r(OP_METADATA);
  r.str(m.name);
  r.map(m.entries);
  put(r.done());
  ++n_metadata_;
}

void McapWriter::writeMessage(uint16_t channel_id, uint32_t sequence, uint64_t log_time, uint64_t publish_time,
                              const uint8_t* data, size_t size) {
  Rec r(OP_MESSAGE);
  r.u16(channel_id);
  r.u32(sequence);
  r.u64(log_time);
  r.u64(publish_time);
  r.raw(data, size);
  const auto& rec = r.done();
  chunk_msgs_[channel_id].emplace_back(log_time, (uint64_t)chunk_.size());  // MessageIndex: offset of the record in the uncompressed chunk
  chunk_.insert(chunk_.end(), rec.begin(), rec.end());
  if (!chunk_has_msg_) {
    chunk_t0_ = chunk_t1_ = log_time;
    chunk_has_msg_ = true;
  } else {
    chunk_t0_ = std::min(chunk_t0_, log_time);
    chunk_t1_ = std::max(chunk_t1_, log_time);
  }
  if (n_messages_ == 0) {
    t_min_ = t_max_ = log_time;
  } else {
    t_min_ = std::min(t_min_, log_time);
    t_max_ = std::max(t_max_, log_time);
  }
  ++n_messages_;
  ++channel_counts_[channel_id];
  if (chunk_.size() >= chunk_size_) flushChunk();
}

void McapWriter::flushChunk() {
  if (chunk_.empty()) return;
  std::vector<uint8_t> packed;
  const char* name = "";
  if (compression_ == McapCompression::Zstd) {
    packed.resize(ZSTD_compressBound(chunk_.size()));
    const size_t r = ZSTD_compress(packed.data(), packed.size(), chunk_.data(), chunk_.size(), 1);
    if (ZSTD_isError(r)) bad("zstd failed");
    packed.resize(r);
    name = "zstd";
  } else if (compression_ == McapCompression::Lz4) {
    packed.resize(LZ4F_compressFrameBound(chunk_.size(), nullptr));
    const size_t r = LZ4F_compressFrame(packed.data(), packed.size(), chunk_.data(), chunk_.size(), nullptr);
    if (LZ4F_isError(r)) bad("lz4 failed");
    packed.resize(r);
    name = "lz4";
  }
  const std::vector<uint8_t>& body = compression_ == McapCompression::None ? chunk_ : packed;
  Rec r(OP_CHUNK);
  r.u64(chunk_t0_);
  r.u64(chunk_t1_);
  r.u64(chunk_.size());
  r.u32(0u);  // uncompressed_crc: 0 = not computed
  r.str(name);
  r.u64(body.size());
  r.raw(body.data(), body.size());
  const auto& rec = r.done();
  ChunkIndex ci{chunk_t0_, chunk_t1_, pos_, rec.size(), body.size(), chunk_.size(), {}, 0};
  put(rec);
  // one MessageIndex record per channel with messages in the chunk, right behind it (ascending channel ids)
  const uint64_t index_start = pos_;
  for (const auto& kv : chunk_msgs_) {
    Rec mi(OP_MESSAGE_INDEX);
    mi.u16(kv.first);
    mi.u32((uint32_t)(kv.second.size() * 16u));
    for (const auto& e : kv.second) {
      mi.u64(e.first);
      mi.u64(e.second);
    }
    ci.message_index_offsets[kv.first] = pos_;
    put(mi.done());
  }
  ci.message_index_length = pos_ - index_start;
  chunk_msgs_.clear();
  chunk_index_.push_back(ci);
  chunk_.clear();
  chunk_has_msg_ = false;
  chunk_t0_ = chunk_t1_ = 0;
}

void McapWriter::close() {
  if (closed_ || !file_) return;
  closed_ = true;
  try {
    finish();
  } catch (...) {  // a failed write leaves no file (and no open handle)
    if (file_) {
      std::fclose(static_cast<FILE*>(file_));
      file_ = nullptr;
    }
    std::remove(tmp_path_.c_str());
    throw;
  }
}

void McapWriter::finish() {
  flushChunk();
  {
    Rec r(OP_DATA_END);
    r.u32(0u);  // data_section_crc: 0 = not computed
    put(r.done());
  }
  // ---- summary section: groups of records + the offsets of the groups
  const uint64_t summary_start = pos_;
  struct Group {
    uint8_t op;
    uint64_t start, length;
  };
  std::vector<Group> groups;
  auto group = [&](uint8_t op, const std::vector<std::vector<uint8_t>>& recs) {
    if (recs.empty()) return;
    const uint64_t start = pos_;
    for (const auto& rec : recs) put(rec);
    groups.push_back({op, start, pos_ - start});
  };
  group(OP_SCHEMA, schema_records_);
  group(OP_CHANNEL, channel_records_);
  {
    std::vector<std::vector<uint8_t>> recs;
    for (const ChunkIndex& ci : chunk_index_) {
      Rec r(OP_CHUNK_INDEX);
      r.u64(ci.start_time);
      r.u64(ci.end_time);
      r.u64(ci.offset);
      r.u64(ci.length);
      r.u32((uint32_t)(ci.message_index_offsets.size() * 10u));  // map<u16 channel, u64 offset of its MessageIndex record>
      for (const auto& kv : ci.message_index_offsets) {
        r.u16(kv.first);
        r.u64(kv.second);
      }
      r.u64(ci.message_index_length);
      r.str(compression_ == McapCompression::Zstd ? "zstd" : compression_ == McapCompression::Lz4 ? "lz4" : "");
      r.u64(ci.compressed_size);
      r.u64(ci.uncompressed_size);
      recs.push_back(r.done());
    }
    group(OP_CHUNK_INDEX, recs);
  }
  {
    Rec r(OP_STATISTICS);
    r.u64(n_messages_);
    r.u16((uint16_t)schema_records_.size());
    r.u32((uint32_t)channel_records_.size());
    r.u32(0u);  // attachments
    r.u32(n_metadata_);
    r.u32((uint32_t)chunk_index_.size());
    r.u64(t_min_);
    r.u64(t_max_);
    r.u32((uint32_t)(channel_counts_.size() * 10u));  // map<u16, u64>
    for (const auto& kv : channel_counts_) {
      r.u16(kv.first);
      r.u64(kv.second);
    }
    std::vector<std::vector<uint8_t>> recs{r.done()};
    group(OP_STATISTICS, recs);
  }
  const uint64_t summary_offset_start = pos_;
  for (const Group& g : groups) {
    Rec r(OP_SUMMARY_OFFSET);
    r.u8(g.op);
    r.u64(g.start);
    r.u64(g.length);
    put(r.done());
  }
  {
    Rec r(OP_FOOTER);
    r.u64(summary_start);
    r.u64(summary_offset_start);
    r.u32(0u);  // summary_crc: 0 = not computed
    put(r.done());
  }
  FILE* f = static_cast<FILE*>(file_);
  const bool ok = std::fwrite(kMagic, 1, 8, f) == 8;
  file_ = nullptr;
  if (std::fclose(f) != 0 || !ok) {
    std::remove(tmp_path_.c_str());
    bad("write failed");
  }
  if (std::rename(tmp_path_.c_str(), path_.c_str()) != 0) {
    std::remove(tmp_path_.c_str());
    bad("cannot move " + tmp_path_ + " to " + path_);
  }
}

// -----------------------------------------------------------------------------------------------------------------
// converter
// -----------------------------------------------------------------------------------------------------------------
namespace {

// What the reader thread (McapSource::next) and the writer thread (McapSink::write) of the pipeline share: the output file
// and the messages read but not written yet, in file order -- point clouds on their way through the GPU pipeline (header
// fields only) and the other messages that lie between them (whole). A message that is not a point cloud and has no cloud
// in front of it goes straight to the output: what waits is bounded by the batches the pipeline keeps in flight.
struct McapFlow {
  McapFlow(McapStream& in_, McapWriter& out_, bool decode, McapTranscodeStats& stats_)
      : in(in_), out(out_), stats(stats_),
        from(decode ? kCompressedPointCloud2SchemaName : kPointCloud2SchemaName),
        to(decode ? kPointCloud2SchemaName : kCompressedPointCloud2SchemaName),
        to_text(decode ? kPointCloud2SchemaText : kCompressedPointCloud2SchemaText) {}
  struct Item {
    bool cloud = false;
    uint64_t id = 0;
    McapMessage head;           // (data / size: of the input message)
    std::vector<uint8_t> body;  // messages copied through
  };
  McapStream& in;
  McapWriter& out;
  McapTranscodeStats& stats;
  const std::string from, to;
  const char* to_text;
  std::mutex mutex;  // `out`, `pending`, `stats`
  std::condition_variable drained;
  std::deque<Item> pending;
  size_t clouds_pending = 0;
  // bytes of the copied-through messages that wait behind a cloud. Beyond kHoldBytes the source asks the pipeline to hand
  // its batch on (MessageSource::submitNow after a cloud; in the middle of a run of other messages nextCloud() gives up
  // with `yielded` set and MessageSource::more() says so) and its next call waits until the sink has written enough of them
  size_t held_bytes = 0;
  bool yielded = false;  // the last nextCloud() returned false because of kHoldBytes, not because the data ended
  static constexpr size_t kHoldBytes = 64u << 20;
  bool failed = false;  // the sink gave up: nothing waits any more
  uint64_t written = 0;  // messages the sink has written (progress, for the wait above)
  uint64_t next_id = 0;
  std::map<uint16_t, bool> schema_is_cloud;   // schema id -> carries the message type to convert
  std::map<uint16_t, uint16_t> channel_schema;

  // reads on until a point-cloud message turns up (true: `m` is it, its bytes in `bytes`, already queued) or the data
  // section ends (false). Everything else is handed on as it comes.
  bool nextCloud(McapMessage& m, std::vector<uint8_t>& bytes, std::string& name) {
    yielded = false;
    {
      // (the clouds delivered so far are on their way -- submitNow() was true after the last one --, so the sink will get to
      // them; a pipeline that has failed never does: no progress for a minute ends the wait with an error of its own, behind
      // which transcodePointClouds rethrows the first one)
      std::unique_lock<std::mutex> lock(mutex);
      uint64_t seen = written;
      int idle = 0;
      while (!drained.wait_for(lock, std::chrono::seconds(1), [&] { return held_bytes <= kHoldBytes || clouds_pending == 0 || failed; })) {
        idle = written == seen ? idle + 1 : 0;
        seen = written;
        if (idle >= 60) bad("the converter stopped taking messages");
      }
    }
    McapStream::Record r;
    while (in.next(r)) {
      switch (r.op) {
        case OP_SCHEMA: {
          McapSchema s = McapStream::parseSchema(r);
          if (s.id == 0) break;
          const bool is_cloud = s.name == from;
          if (is_cloud) {  // the schema of the converted messages (duplicateSchemasAndChannels, mcap_converter.cpp:59-125)
            s.name = to;
            s.data.assign(to_text, to_text + std::strlen(to_text));
          }
          std::lock_guard<std::mutex> lock(mutex);
          if (schema_is_cloud.find(s.id) == schema_is_cloud.end()) out.addSchema(s);  // (a repeated record: same content)
          schema_is_cloud[s.id] = is_cloud;
          break;
        }
        case OP_CHANNEL: {
          const McapChannel ch = McapStream::parseChannel(r);
          std::lock_guard<std::mutex> lock(mutex);
          if (channel_schema.find(ch.id) == channel_schema.end()) out.addChannel(ch);
          channel_schema[ch.id] = ch.schema_id;
          break;
        }
        case OP_METADATA: {
          const McapMetadata md = McapStream::parseMetadata(r);
          std::lock_guard<std::mutex> lock(mutex);
          out.addMetadata(md);
          break;
        }
        case OP_MESSAGE: {
          m = McapStream::parseMessage(r);
          std::lock_guard<std::mutex> lock(mutex);
          ++stats.messages;
          const auto ch = channel_schema.find(m.channel_id);
          if (ch == channel_schema.end()) bad("message on a channel the file does not declare");
          const auto sc = schema_is_cloud.find(ch->second);
          if (sc != schema_is_cloud.end() && sc->second) {
            Item it;
            it.cloud = true;
            it.id = next_id++;
            it.head = m;
            it.head.data = nullptr;
            name = std::to_string(it.id);
            pending.push_back(std::move(it));
            ++clouds_pending;
            bytes.resize(m.size);
            if (m.size) std::memcpy(bytes.data(), m.data, m.size);
            return true;
          }
          if (clouds_pending == 0) {
            out.writeMessage(m.channel_id, m.sequence, m.log_time, m.publish_time, m.data, m.size);
          } else {
            Item it;
            it.head = m;
            it.head.data = nullptr;
            it.body.assign(m.data, m.data + m.size);
            held_bytes += it.body.size();
            stats.peak_held_bytes = std::max<uint64_t>(stats.peak_held_bytes, held_bytes);
            pending.push_back(std::move(it));
            if (held_bytes > kHoldBytes) {
              // enough is held back behind clouds that may still sit in a batch the pipeline has not handed on (sparse clouds,
              // a lidar topic that ends early, an odd count against --batch): hand control back -- the pipeline submits what
              // it has, the next call waits at the top until the sink has caught up
              yielded = true;
              return false;
            }
          }
          break;
        }
        default:
          break;  // indexes, attachments, statistics, unknown opcodes: skipped (as McapFile does)
      }
    }
    return false;
  }
  // (mutex held) the messages in front of the next cloud leave
  void flushCopies() {
    while (!pending.empty() && !pending.front().cloud) {
      const Item& it = pending.front();
      out.writeMessage(it.head.channel_id, it.head.sequence, it.head.log_time, it.head.publish_time, it.body.data(), it.body.size());
      held_bytes -= it.body.size();
      pending.pop_front();
    }
    drained.notify_all();
  }
};

class McapSource : public MessageSource {
 public:
  explicit McapSource(McapFlow& flow) : flow_(flow) {}
  // the first cloud was read ahead (to know whether the pipeline is needed at all)
  void pushBack(std::string name, std::vector<uint8_t> bytes) {
    have_first_ = true;
    first_name_ = std::move(name);
    first_bytes_ = std::move(bytes);
  }
  bool submitNow() const override {
    std::lock_guard<std::mutex> lock(flow_.mutex);
    return flow_.held_bytes > McapFlow::kHoldBytes;
  }
  bool more() const override { return flow_.yielded; }  // (reader thread only, like nextCloud)
  bool next(Message& out) override {
    if (have_first_) {
      have_first_ = false;
      out.name = first_name_;
      out.bytes.resize(first_bytes_.size());
      if (!first_bytes_.empty()) std::memcpy(out.bytes.data(), first_bytes_.data(), first_bytes_.size());
      first_bytes_ = std::vector<uint8_t>();
      return true;
    }
    McapMessage m;
    if (!flow_.nextCloud(m, scratch_, out.name)) return false;
    out.bytes.resize(scratch_.size());
    if (!scratch_.empty()) std::memcpy(out.bytes.data(), scratch_.data(), scratch_.size());
    return true;
  }

 private:
  McapFlow& flow_;
  bool have_first_ = false;
  std::string first_name_;
  std::vector<uint8_t> first_bytes_, scratch_;
};

// converted messages arrive in input order; every other message of the bag is copied through in front of the converted
// message that follows it in the file
class McapSink : public MessageSink {
 public:
  explicit McapSink(McapFlow& flow) : flow_(flow) {}
  void write(const std::string& name, const uint8_t* data, size_t size) override {
    const uint64_t id = std::stoull(name);
    std::lock_guard<std::mutex> lock(flow_.mutex);
    try {
      flow_.flushCopies();
      if (flow_.pending.empty() || flow_.pending.front().id != id) bad("converted messages arrived out of order");
      const McapMessage& h = flow_.pending.front().head;
      flow_.out.writeMessage(h.channel_id, h.sequence, h.log_time, h.publish_time, data, size);
      flow_.stats.input_bytes += h.size;
      flow_.stats.output_bytes += size;
      ++flow_.stats.converted;
      ++flow_.written;
      flow_.pending.pop_front();
      --flow_.clouds_pending;
      flow_.flushCopies();
    } catch (...) {
      flow_.failed = true;
      flow_.drained.notify_all();
      throw;
    }
  }

 private:
  McapFlow& flow_;
};

}  // namespace

McapTranscodeStats transcodeMcap(const std::string& file_in, const std::string& file_out, TranscodeOptions options,
                                 McapCompression mcap_compression) {
  McapStream in(file_in);
  McapTranscodeStats stats;
  // no need to compress twice (mcap_converter.cpp:199-202)
  if (!options.decode && mcap_compression == McapCompression::Zstd) options.compression = Cloudini::CompressionOption::NONE;

  McapWriter out(file_out, in.profile, mcap_compression);
  McapFlow flow(in, out, options.decode, stats);
  McapSource source(flow);
  McapSink sink(flow);
  // up to the first point cloud nothing needs the pipeline (a bag without one is copied without a GPU)
  McapMessage first;
  std::vector<uint8_t> first_bytes;
  std::string first_name;
  if (flow.nextCloud(first, first_bytes, first_name)) {
    source.pushBack(std::move(first_name), std::move(first_bytes));
    stats.pipeline = transcodePointClouds(source, sink, options);
  }
  {
    std::lock_guard<std::mutex> lock(flow.mutex);
    flow.flushCopies();
    if (!flow.pending.empty()) bad("point clouds were read but not converted");
  }
  out.close();
  return stats;
}

}  // namespace cloudini_amd
