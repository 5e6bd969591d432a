// MCAP writer -> reader round trip without a GPU (tests/test_mcap_io.py): three compressions, chunks of a few hundred bytes
// (many chunks), schemas / channels / metadata / interleaved messages come back as they were written. Also leaves the
// uncompressed file behind for the Python reader of the test, which restates the record layouts on its own.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "cloudini_amd/mcap_io.hpp"

using namespace cloudini_amd;

#define CHECK(c)                                                       \
  do {                                                                 \
    if (!(c)) {                                                        \
      std::fprintf(stderr, "FAILED %s (line %d)\n", #c, __LINE__);     \
      return 1;                                                        \
    }                                                                  \
  } while (0)

static std::vector<uint8_t> payload(uint32_t i) {
  std::vector<uint8_t> v(17u + (i * 37u) % 400u);
  for (size_t k = 0; k < v.size(); ++k) v[k] = (uint8_t)((i * 131u + k * 7u) >> (k & 3));
  return v;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = argv[1];
  const McapCompression comps[3] = {McapCompression::None, McapCompression::Lz4, McapCompression::Zstd};
  const char* names[3] = {"none", "lz4", "zstd"};
  for (int c = 0; c < 3; ++c) {
    const std::string path = dir + "/rt_" + names[c] + ".mcap";
    {
      McapWriter w(path, "ros2", comps[c], 700);
      McapSchema s1;
      s1.id = 1;
      s1.name = kPointCloud2SchemaName;
      s1.encoding = "ros2msg";
      s1.data.assign(kPointCloud2SchemaText, kPointCloud2SchemaText + std::strlen(kPointCloud2SchemaText));
      McapSchema s2;
      s2.id = 7;
      s2.name = "std_msgs/msg/String";
      s2.encoding = "ros2msg";
      s2.data = {'s', 't', 'r', 'i', 'n', 'g', ' ', 'd', 'a', 't', 'a'};
      w.addSchema(s1);
      w.addSchema(s2);
      McapChannel a{3, 1, "/lidar/points", "cdr", {{"offered_qos_profiles", "x"}}};
      McapChannel b{4, 7, "/chatter", "cdr", {}};
      McapChannel d{9, 1, "/depth/points", "cdr", {{"k", "v"}, {"k2", ""}}};
      w.addChannel(a);
      w.addChannel(b);
      w.addChannel(d);
      w.addMetadata(McapMetadata{"rosbag2", {{"ROS_DISTRO", "jazzy"}}});
      for (uint32_t i = 0; i < 60; ++i) {
        const std::vector<uint8_t> p = payload(i);
        const uint16_t ch = i % 3 == 0 ? 3 : (i % 3 == 1 ? 4 : 9);
        w.writeMessage(ch, i, 1000u + 10u * i, 999u + 10u * i, p.data(), p.size());
      }
      w.close();
    }
    McapFile f(path);
    CHECK(f.profile == "ros2");
    CHECK(f.library == "cloudini_amd");
    CHECK(f.schemas.size() == 2 && f.schemas.at(1).name == kPointCloud2SchemaName && f.schemas.at(7).name == "std_msgs/msg/String");
    CHECK(f.schemas.at(1).data.size() == std::strlen(kPointCloud2SchemaText));
    CHECK(f.schemas.at(7).encoding == "ros2msg" && f.schemas.at(7).data.size() == 11);
    CHECK(f.channels.size() == 3 && f.channels.at(3).topic == "/lidar/points" && f.channels.at(9).schema_id == 1);
    CHECK(f.channels.at(9).metadata.size() == 2 && f.channels.at(9).metadata[1].first == "k2" && f.channels.at(9).metadata[1].second.empty());
    CHECK(f.channels.at(4).message_encoding == "cdr" && f.channels.at(4).metadata.empty());
    CHECK(f.metadata.size() == 1 && f.metadata[0].name == "rosbag2" && f.metadata[0].entries[0].second == "jazzy");
    CHECK(f.messages.size() == 60);
    for (uint32_t i = 0; i < 60; ++i) {
      const McapMessage& m = f.messages[i];
      const std::vector<uint8_t> p = payload(i);
      CHECK(m.sequence == i && m.log_time == 1000u + 10u * i && m.publish_time == 999u + 10u * i);
      CHECK(m.channel_id == (i % 3 == 0 ? 3 : (i % 3 == 1 ? 4 : 9)));
      CHECK(m.size == p.size() && std::memcmp(m.data, p.data(), p.size()) == 0);
    }
  }
  // malformed files are refused, not read past their end
  {
    const std::string path = dir + "/rt_none.mcap";
    FILE* fp = std::fopen(path.c_str(), "rb");
    std::vector<uint8_t> img(1 << 20);
    img.resize(std::fread(img.data(), 1, img.size(), fp));
    std::fclose(fp);
    const std::string cut = dir + "/cut.mcap";
    fp = std::fopen(cut.c_str(), "wb");
    std::fwrite(img.data(), 1, img.size() / 2, fp);
    std::fclose(fp);
    bool threw = false;
    try {
      McapFile g(cut);
    } catch (const std::exception&) {
      threw = true;
    }
    CHECK(threw);
  }
  if (argc > 2 && std::string(argv[2]) == "abandon") {
    // a writer that is destroyed without close() (an exception on the way) leaves neither the file nor its ".partial"
    const std::string path = dir + "/abandoned.mcap";
    {
      McapWriter w(path, "ros2", McapCompression::None, 700);
      const std::vector<uint8_t> p = payload(1);
      McapChannel a{3, 0, "/x", "cdr", {}};
      w.addChannel(a);
      for (uint32_t i = 0; i < 20; ++i) w.writeMessage(3, i, i, i, p.data(), p.size());
    }
    FILE* f1 = std::fopen(path.c_str(), "rb");
    FILE* f2 = std::fopen((path + ".partial").c_str(), "rb");
    CHECK(f1 == nullptr && f2 == nullptr);
    std::printf("abandoned writer left nothing\n");
    // a chunk record that claims 4 GiB of uncompressed bytes behind a 20-byte zstd frame: refused before anything is allocated
    const std::string big = dir + "/big.mcap";
    FILE* fp = std::fopen(big.c_str(), "wb");
    const uint8_t magic[8] = {0x89, 'M', 'C', 'A', 'P', '0', '\r', '\n'};
    std::fwrite(magic, 1, 8, fp);
    auto rec = [&](uint8_t op, const std::vector<uint8_t>& body) {
      std::fputc(op, fp);
      const uint64_t n = body.size();
      std::fwrite(&n, 8, 1, fp);
      std::fwrite(body.data(), 1, body.size(), fp);
    };
    auto u32 = [](std::vector<uint8_t>& v, uint32_t x) { for (int k = 0; k < 4; ++k) v.push_back((uint8_t)(x >> (8 * k))); };
    auto u64 = [](std::vector<uint8_t>& v, uint64_t x) { for (int k = 0; k < 8; ++k) v.push_back((uint8_t)(x >> (8 * k))); };
    std::vector<uint8_t> h;
    u32(h, 0);
    u32(h, 0);
    rec(0x01, h);
    std::vector<uint8_t> c;
    u64(c, 0);
    u64(c, 0);
    u64(c, 4ull << 30);  // uncompressed_size
    u32(c, 0);
    u32(c, 4);
    c.insert(c.end(), {'z', 's', 't', 'd'});
    u64(c, 20);
    for (int k = 0; k < 20; ++k) c.push_back((uint8_t)k);
    rec(0x06, c);
    std::vector<uint8_t> e;
    u32(e, 0);
    rec(0x0F, e);
    std::vector<uint8_t> ft;
    u64(ft, 0);
    u64(ft, 0);
    u32(ft, 0);
    rec(0x02, ft);
    std::fwrite(magic, 1, 8, fp);
    std::fclose(fp);
    bool refused = false;
    try {
      McapFile g(big);
    } catch (const std::runtime_error& ex) {
      refused = std::string(ex.what()).find("chunk") != std::string::npos;
    }
    CHECK(refused);
    std::printf("oversized chunk refused\n");
  }
  std::printf("all checks passed\n");
  return 0;
}
