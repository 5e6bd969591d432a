// The batch transcoder's pipeline without a GPU: TranscodeOptions::test_stage puts the caller's function in the place of
// the GPU stages (and stage 2), so that what is exercised is what a multi-GPU run adds -- batches handed to whichever of
// N stages is free, the writer putting them back into input order, an error on any stage stopping all of them with the
// output a prefix of the input (include/cloudini_amd/batch_transcoder.hpp; cloudini_amd/csrc/host/batch_transcoder.cpp).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "cloudini_amd/batch_transcoder.hpp"

using namespace cloudini_amd;

#define CHECK(c)                                                          \
  do {                                                                    \
    if (!(c)) {                                                           \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);          \
      return 1;                                                           \
    }                                                                     \
  } while (0)

struct CountingSource : MessageSource {
  size_t n, at = 0;
  explicit CountingSource(size_t n_) : n(n_) {}
  bool next(Message& out) override {
    if (at >= n) return false;
    out.name = "m" + std::to_string(at);
    out.bytes.resize(16 + at % 7);
    for (size_t i = 0; i < out.bytes.size(); ++i) out.bytes[i] = static_cast<uint8_t>(at * 31 + i);
    ++at;
    return true;
  }
};

struct OrderSink : MessageSink {
  std::vector<std::string> names;
  std::vector<std::vector<uint8_t>> data;
  void write(const std::string& name, const uint8_t* d, size_t size) override {
    names.push_back(name);
    data.emplace_back(d, d + size);
  }
};

static std::vector<uint8_t> transformed(const Message& m) {  // what a "stage" makes of a message
  std::vector<uint8_t> o(m.bytes.begin(), m.bytes.end());
  for (uint8_t& b : o) b = static_cast<uint8_t>(b ^ 0x5a);
  o.push_back(static_cast<uint8_t>(m.bytes.size()));
  return o;
}

int main() {
  // 1. three stages of very different speed: every message arrives, in input order, transformed by exactly one stage
  for (size_t workers : {1u, 3u, 5u}) {
    const size_t n = 237;
    CountingSource src(n);
    OrderSink sink;
    TranscodeOptions opt;
    opt.batch_messages = 4;
    opt.test_workers = workers;
    std::atomic<size_t> per_worker[8] = {};
    opt.test_stage = [&](size_t w, const std::vector<Message>& in, std::vector<std::vector<uint8_t>>& out) {
      std::this_thread::sleep_for(std::chrono::microseconds(w == 0 ? 1500 : 100 * (w + 1)));  // stage 0 is the slow one
      for (size_t i = 0; i < in.size(); ++i) out[i] = transformed(in[i]);
      per_worker[w] += in.size();
    };
    const TranscodeStats st = transcodePointClouds(src, sink, opt);
    CHECK(st.messages == n && st.gpu_workers == workers);
    CHECK(sink.names.size() == n);
    CountingSource again(n);
    Message m;
    for (size_t k = 0; k < n; ++k) {
      CHECK(again.next(m));
      CHECK(sink.names[k] == m.name);
      CHECK(sink.data[k] == transformed(m));
    }
    size_t total = 0, busy = 0;
    for (size_t w = 0; w < workers; ++w) {
      total += per_worker[w];
      busy += per_worker[w] != 0;
    }
    CHECK(total == n);
    if (workers >= 3) CHECK(busy >= 2);  // the fast stages took over work while stage 0 slept
  }
  // 2. a stage fails in the middle: the error reaches the caller, what was written is a prefix of the input in order
  {
    const size_t n = 400;
    CountingSource src(n);
    OrderSink sink;
    TranscodeOptions opt;
    opt.batch_messages = 5;
    opt.test_workers = 4;
    std::atomic<size_t> batches{0};
    opt.test_stage = [&](size_t w, const std::vector<Message>& in, std::vector<std::vector<uint8_t>>& out) {
      std::this_thread::sleep_for(std::chrono::microseconds(200 * (w + 1)));
      if (batches.fetch_add(1) == 23) throw std::runtime_error("stage failure injected by the test");
      for (size_t i = 0; i < in.size(); ++i) out[i] = transformed(in[i]);
    };
    bool thrown = false;
    try {
      transcodePointClouds(src, sink, opt);
    } catch (const std::runtime_error& e) {
      thrown = std::strstr(e.what(), "injected") != nullptr;
    }
    CHECK(thrown);
    CHECK(sink.names.size() < n);
    for (size_t k = 0; k < sink.names.size(); ++k) CHECK(sink.names[k] == "m" + std::to_string(k));
  }
  // 3. an empty source
  {
    CountingSource src(0);
    OrderSink sink;
    TranscodeOptions opt;
    opt.test_workers = 3;
    opt.test_stage = [&](size_t, const std::vector<Message>&, std::vector<std::vector<uint8_t>>&) {};
    const TranscodeStats st = transcodePointClouds(src, sink, opt);
    CHECK(st.messages == 0 && sink.names.empty());
  }
  std::printf("all checks passed\n");
  return 0;
}
