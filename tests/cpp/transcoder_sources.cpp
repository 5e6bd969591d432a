// The two kinds of MessageSource / MessageSink the batch transcoder accepts (include/cloudini_amd/batch_transcoder.hpp):
// a sequential source (next() only) with a sink that insists on input order, and the concurrent directory pair read and
// written by several threads. Same messages, same bytes, same order. argv: <directory of CDR PointCloud2 files>.
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "cloudini_amd/batch_transcoder.hpp"

namespace ca = cloudini_amd;

struct VectorSource : ca::MessageSource {  // sequential: no claim / fetch
  std::vector<std::pair<std::string, std::vector<uint8_t>>> items;
  size_t at = 0;
  bool next(ca::Message& out) override {
    if (at >= items.size()) return false;
    out.name = items[at].first;
    out.bytes.assign(items[at].second.begin(), items[at].second.end());
    ++at;
    return true;
  }
};

struct OrderedSink : ca::MessageSink {  // not concurrent: write() must come in input order, one call at a time
  std::vector<std::pair<std::string, std::vector<uint8_t>>> got;
  void write(const std::string& name, const uint8_t* data, size_t size) override { got.emplace_back(name, std::vector<uint8_t>(data, data + size)); }
};

struct MapSink : ca::MessageSink {  // concurrent: any order inside a batch
  std::mutex mutex;
  std::map<std::string, std::vector<uint8_t>> got;
  bool concurrent() const override { return true; }
  void write(const std::string& name, const uint8_t* data, size_t size) override {
    std::lock_guard<std::mutex> lock(mutex);
    got[name].assign(data, data + size);
  }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  int failures = 0;
  // the messages, once through the directory source (sequentially) into memory
  VectorSource vs;
  {
    ca::DirectorySource ds(argv[1]);
    ca::Message m;
    while (ds.next(m)) vs.items.emplace_back(m.name, std::vector<uint8_t>(m.bytes.begin(), m.bytes.end()));
  }
  if (vs.items.size() < 5) {
    std::printf("too few messages\n");
    return 2;
  }
  ca::TranscodeOptions opt;
  opt.batch_messages = 3;  // several batches, the last one short
  opt.compression = Cloudini::CompressionOption::ZSTD;

  opt.io_threads = 1;
  OrderedSink ordered;
  const ca::TranscodeStats s1 = ca::transcodePointClouds(vs, ordered, opt);
  if (s1.messages != vs.items.size() || ordered.got.size() != vs.items.size()) {
    std::printf("sequential run: %llu messages, %zu written, %zu expected\n", (unsigned long long)s1.messages, ordered.got.size(), vs.items.size());
    ++failures;
  }
  for (size_t i = 0; i < ordered.got.size() && i < vs.items.size(); ++i)
    if (ordered.got[i].first != vs.items[i].first) {
      std::printf("sequential run: message %zu is %s, expected %s\n", i, ordered.got[i].first.c_str(), vs.items[i].first.c_str());
      ++failures;
      break;
    }

  opt.io_threads = 4;
  ca::DirectorySource ds(argv[1]);
  MapSink concurrent;
  const ca::TranscodeStats s2 = ca::transcodePointClouds(ds, concurrent, opt);
  if (s2.messages != vs.items.size() || concurrent.got.size() != vs.items.size()) {
    std::printf("concurrent run: %llu messages, %zu written\n", (unsigned long long)s2.messages, concurrent.got.size());
    ++failures;
  }
  for (const auto& kv : ordered.got) {
    auto it = concurrent.got.find(kv.first);
    if (it == concurrent.got.end() || it->second != kv.second) {
      std::printf("message %s differs between the two runs\n", kv.first.c_str());
      ++failures;
    }
  }
  // a source that ends exactly at a batch boundary, and an empty one
  VectorSource six;
  six.items.assign(vs.items.begin(), vs.items.begin() + 3);
  OrderedSink three;
  opt.io_threads = 1;
  if (ca::transcodePointClouds(six, three, opt).messages != 3 || three.got.size() != 3) {
    std::printf("exact batch: %zu written\n", three.got.size());
    ++failures;
  }
  VectorSource none;
  OrderedSink nothing;
  if (ca::transcodePointClouds(none, nothing, opt).messages != 0 || !nothing.got.empty()) ++failures;
  ca::releasePinnedCache();
  if (failures) {
    std::printf("%d check(s) FAILED\n", failures);
    return 1;
  }
  std::printf("all checks passed\n");
  return 0;
}
