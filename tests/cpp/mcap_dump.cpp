// prints what McapFile -- or, with "stream" as the second argument, McapStream -- read from a file, one line per item
// (tests/test_mcap_io.py compares it with what the Python writer put in)
#include <cstdio>
#include <cstring>
#include <map>
#include <string>

#include "cloudini_amd/mcap_io.hpp"

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  try {
    if (argc > 2 && !std::strcmp(argv[2], "stream")) {  // the same lines from the record stream (same grouping: the kinds sorted)
      cloudini_amd::McapStream in(argv[1]);
      std::printf("header %s\n", in.profile.c_str());
      std::map<unsigned, std::string> schemas, channels;
      std::string metadata, messages;
      cloudini_amd::McapStream::Record r;
      char line[512];
      while (in.next(r)) {
        if (r.op == 0x03) {
          const auto s = cloudini_amd::McapStream::parseSchema(r);
          std::snprintf(line, sizeof line, "schema %u %s %s %zu\n", s.id, s.name.c_str(), s.encoding.c_str(), s.data.size());
          if (s.id != 0) schemas[s.id] = line;
        } else if (r.op == 0x04) {
          const auto c = cloudini_amd::McapStream::parseChannel(r);
          std::snprintf(line, sizeof line, "channel %u %u %s %s %zu\n", c.id, c.schema_id, c.topic.c_str(), c.message_encoding.c_str(), c.metadata.size());
          channels[c.id] = line;
        } else if (r.op == 0x0C) {
          const auto m = cloudini_amd::McapStream::parseMetadata(r);
          std::snprintf(line, sizeof line, "metadata %s %zu\n", m.name.c_str(), m.entries.size());
          metadata += line;
        } else if (r.op == 0x05) {
          const auto m = cloudini_amd::McapStream::parseMessage(r);
          unsigned long long h = 1469598103934665603ull;
          for (size_t i = 0; i < m.size; ++i) h = (h ^ m.data[i]) * 1099511628211ull;
          std::snprintf(line, sizeof line, "message %u %u %llu %llu %zu %llu\n", m.channel_id, m.sequence, (unsigned long long)m.log_time,
                        (unsigned long long)m.publish_time, m.size, h);
          messages += line;
        }
      }
      for (const auto& kv : schemas) std::fputs(kv.second.c_str(), stdout);
      for (const auto& kv : channels) std::fputs(kv.second.c_str(), stdout);
      std::fputs(metadata.c_str(), stdout);
      std::fputs(messages.c_str(), stdout);
      return 0;
    }
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
