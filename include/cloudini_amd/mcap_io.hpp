// MCAP container I/O for the batch transcoder ("batched MCAP/rosbag transcoder").
//
// What it replaces in the reference: the mcap library calls of McapConverter (cloudini_lib/tools/src/mcap_converter.cpp:
// open :32-57, duplicateSchemasAndChannels :59-125, encodePointClouds :141-222, decodePointClouds :240-300). The reference
// links the upstream `mcap` C++ library; this image has neither that library nor a sample bag, so the reader and the writer
// below are written against the published container specification (https://mcap.dev/spec, version 0x30) and are pinned only
// by their own round trip and by hand-checked record layouts (tests/test_mcap_io.py) -- PARITY UNPINNED against the mcap
// library itself. What is pinned as everywhere else: every converted message's bytes equal the reference's converter.
//
//   McapFile    reads a whole file: Header, Schema / Channel / Message / Metadata records of the data section, inside Chunk
//               records ("" / "zstd" / "lz4" frame compression) or outside; messages in FILE order (the order
//               McapReader::readMessages takes by default). Index and summary records are not needed and skipped.
//   McapStream  (round 5) the same records one after the other, straight from the file: one record -- or one decompressed
//               chunk -- in memory at a time. transcodeMcap reads through it: a bag of many gigabytes is converted in the
//               memory of a few batches, like the reference's converter streams through McapReader.
//   McapWriter  Header, Schema / Channel records, Messages in chunks of <= chunk_size uncompressed bytes, Metadata, DataEnd, a
//               summary section (Schemas, Channels, ChunkIndexes, Statistics, SummaryOffsets) and the Footer. Round 5: one
//               MessageIndex record per channel behind every chunk ((log_time, offset of the Message record inside the
//               uncompressed chunk)), their places in the ChunkIndex's message_index_offsets -- indexed readers pick the
//               chunks to read from that map. The file is written under "<path>.partial" and renamed by close(): a
//               conversion that fails leaves no well-formed file that is silently missing its tail.
//   transcodeMcap  the converter: schemas and channels duplicated with the point-cloud schema swapped, metadata copied,
//               PointCloud2 (or CompressedPointCloud2) messages through the batched GPU pipeline, everything else copied.
#pragma once

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "cloudini_amd/batch_transcoder.hpp"

namespace cloudini_amd {

extern const char* const kPointCloud2SchemaName;            // "sensor_msgs/msg/PointCloud2"
extern const char* const kCompressedPointCloud2SchemaName;  // "point_cloud_interfaces/msg/CompressedPointCloud2"
extern const char* const kPointCloud2SchemaText;            // ros2msg definitions (field lines only: no comments)
extern const char* const kCompressedPointCloud2SchemaText;

struct McapSchema {
  uint16_t id = 0;
  std::string name, encoding;
  std::vector<uint8_t> data;
};
struct McapChannel {
  uint16_t id = 0, schema_id = 0;
  std::string topic, message_encoding;
  std::vector<std::pair<std::string, std::string>> metadata;
};
struct McapMetadata {
  std::string name;
  std::vector<std::pair<std::string, std::string>> entries;
};
struct McapMessage {
  uint16_t channel_id = 0;
  uint32_t sequence = 0;
  uint64_t log_time = 0, publish_time = 0;
  const uint8_t* data = nullptr;  // into the file image or a decompressed chunk owned by the McapFile
  size_t size = 0;
};

class McapFile {
 public:
  explicit McapFile(const std::string& path);  // throws std::runtime_error on a malformed file
  std::string profile, library;
  std::map<uint16_t, McapSchema> schemas;
  std::map<uint16_t, McapChannel> channels;
  std::vector<McapMetadata> metadata;
  std::vector<McapMessage> messages;  // file order

 private:
  void parseRecords(const uint8_t* p, const uint8_t* end, bool in_chunk);
  std::vector<uint8_t> image_;
  std::vector<std::vector<uint8_t>> chunks_;
};
// (McapFile keeps the file image and every decompressed chunk until it is destroyed: messages point into them -- small
// files, tests, tools. A chunk that claims more than kMcapMaxChunkBytes uncompressed bytes, or more than its compressed
// frame can hold, is refused.)
constexpr uint64_t kMcapMaxChunkBytes = 1ull << 30;

// The records of the data section in file order, one at a time (Chunk records are opened: their records follow). What a
// Record points at is valid until the next call. Ends (false) at DataEnd / the Footer. Throws std::runtime_error like McapFile.
class McapStream {
 public:
  explicit McapStream(const std::string& path);  // opens, checks both magics, reads the Header record
  ~McapStream();
  McapStream(const McapStream&) = delete;
  McapStream& operator=(const McapStream&) = delete;
  struct Record {
    uint8_t op = 0;
    const uint8_t* data = nullptr;
    size_t size = 0;
  };
  bool next(Record& r);
  std::string profile, library;
  // parsers of the record bodies McapFile also understands
  static McapSchema parseSchema(const Record& r);
  static McapChannel parseChannel(const Record& r);
  static McapMetadata parseMetadata(const Record& r);
  static McapMessage parseMessage(const Record& r);  // (data points into the record)

 private:
  void readExact(void* dst, size_t n, const char* what);
  void* file_ = nullptr;
  std::string path_;
  uint64_t pos_ = 0, size_ = 0;
  std::vector<uint8_t> rec_, chunk_;
  size_t chunk_at_ = 0;
  bool in_chunk_ = false, ended_ = false;
};

enum class McapCompression { None, Lz4, Zstd };

class McapWriter {
 public:
  McapWriter(const std::string& path, const std::string& profile, McapCompression compression, size_t chunk_size = 2u << 20);
  ~McapWriter();
  void addSchema(const McapSchema& s);
  void addChannel(const McapChannel& c);
  void addMetadata(const McapMetadata& m);
  void writeMessage(uint16_t channel_id, uint32_t sequence, uint64_t log_time, uint64_t publish_time, const uint8_t* data, size_t size);
  void close();

 private:
  struct ChunkIndex {
    uint64_t start_time, end_time, offset, length, compressed_size, uncompressed_size;
    std::map<uint16_t, uint64_t> message_index_offsets;  // channel -> file offset of its MessageIndex record
    uint64_t message_index_length = 0;
  };
  void flushChunk();
  void finish();
  void put(const std::vector<uint8_t>& record);
  void* file_ = nullptr;
  std::string path_, tmp_path_;
  uint64_t pos_ = 0;
  std::map<uint16_t, std::vector<std::pair<uint64_t, uint64_t>>> chunk_msgs_;  // open chunk: channel -> (log_time, offset)
  McapCompression compression_;
  size_t chunk_size_;
  std::vector<uint8_t> chunk_;  // uncompressed records of the open chunk
  uint64_t chunk_t0_ = 0, chunk_t1_ = 0;
  bool chunk_has_msg_ = false;
  std::vector<std::vector<uint8_t>> schema_records_, channel_records_;
  std::vector<ChunkIndex> chunk_index_;
  std::map<uint16_t, uint64_t> channel_counts_;
  uint64_t n_messages_ = 0, t_min_ = 0, t_max_ = 0;
  uint32_t n_metadata_ = 0;
  bool closed_ = false;
};

struct McapTranscodeStats {
  uint64_t messages = 0, converted = 0, input_bytes = 0, output_bytes = 0;
  uint64_t peak_held_bytes = 0;  // most bytes of copied-through messages that waited for a point cloud in front of them
  TranscodeStats pipeline;
};

// McapConverter::encodePointClouds / decodePointClouds (options.decode) on the batched pipeline. `mcap_compression` is the
// chunk compression of the output file; as in the reference (mcap_converter.cpp:199-202) a ZSTD-compressed container turns
// the messages' own second stage off.
McapTranscodeStats transcodeMcap(const std::string& file_in, const std::string& file_out, TranscodeOptions options,
                                 McapCompression mcap_compression);

}  // namespace cloudini_amd
