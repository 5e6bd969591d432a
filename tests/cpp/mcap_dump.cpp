// prints what McapFile read from a file, one line per item (tests/test_mcap_io.py compares it with what the Python writer put in)
#include <cstdio>
#include <string>

#include "cloudini_amd/mcap_io.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  try {
    const cloudini_amd::McapFile f(argv[1]);
    std::printf("header %s\n", f.profile.c_str());
    for (const auto& kv : f.schemas) std::printf("schema %u %s %s %zu\n", kv.first, kv.second.name.c_str(), kv.second.encoding.c_str(), kv.second.data.size());
    for (const auto& kv : f.channels)
      std::printf("channel %u %u %s %s %zu\n", kv.first, kv.second.schema_id, kv.second.topic.c_str(), kv.second.message_encoding.c_str(), kv.second.metadata.size());
    for (const auto& m : f.metadata) std::printf("metadata %s %zu\n", m.name.c_str(), m.entries.size());
    for (const auto& m : f.messages) {
      unsigned long long h = 1469598103934665603ull;
      for (size_t i = 0; i < m.size; ++i) h = (h ^ m.data[i]) * 1099511628211ull;
      std::printf("message %u %u %llu %llu %zu %llu\n", m.channel_id, m.sequence, (unsigned long long)m.log_time,
                  (unsigned long long)m.publish_time, m.size, h);
    }
  } catch (const std::exception& e) {
    std::printf("error %s\n", e.what());
    return 1;
  }
  return 0;
}
