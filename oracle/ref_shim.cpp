// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or called from the product path.
//
// C shim around the *real* reference (facontidavide/cloudini, cloudini_lib 1.2.2) compiled from
// the sources where they lie under /root/reference (see oracle/Makefile). It exposes the reference's
// PointcloudEncoder / PointcloudDecoder / header / ROS-message functions with a plain C ABI so that
// tests (ctypes) can (a) validate the plain-C restatement in oracle/cloudini_oracle.c, (b) generate the
// golden fixtures under tests/golden/, and (c) serve as bench.py's cpu_baseline (kind "reference").
//
// This file is our own code; it only *includes* reference headers. No reference source is copied.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "cloudini_lib/cloudini.hpp"
#include "cloudini_lib/ros_msg_utils.hpp"

#define REF_API extern "C" __attribute__((visibility("default")))

namespace {

struct RefField {
  const char* name;
  uint32_t offset;
  uint8_t type;          // Cloudini::FieldType
  uint8_t has_resolution;
  uint8_t pad[2];
  float resolution;
};

thread_local std::string g_last_error;

Cloudini::EncodingInfo makeInfo(
    const RefField* fields, uint32_t n_fields, uint32_t point_step, uint32_t width, uint32_t height,
    uint8_t encoding_opt, uint8_t compression_opt, uint8_t version, uint8_t use_threads) {
  Cloudini::EncodingInfo info;
  for (uint32_t i = 0; i < n_fields; ++i) {
    Cloudini::PointField f;
    f.name = fields[i].name ? fields[i].name : "";
    f.offset = fields[i].offset;
    f.type = static_cast<Cloudini::FieldType>(fields[i].type);
    if (fields[i].has_resolution) {
      f.resolution = fields[i].resolution;
    }
    info.fields.push_back(f);
  }
  info.point_step = point_step;
  info.width = width;
  info.height = height;
  info.encoding_opt = static_cast<Cloudini::EncodingOptions>(encoding_opt);
  info.compression_opt = static_cast<Cloudini::CompressionOption>(compression_opt);
  info.version = version;
  info.use_threads = use_threads != 0;
  return info;
}

}  // namespace

REF_API const char* ref_last_error() {
  return g_last_error.c_str();
}

// Full encode (header + chunks), exactly PointcloudEncoder::encode(ConstBufferView, BufferView&, true).
// Returns bytes written, or -1 on exception.
REF_API int64_t ref_encode(
    const RefField* fields, uint32_t n_fields, uint32_t point_step, uint32_t width, uint32_t height,
    uint8_t encoding_opt, uint8_t compression_opt, uint8_t version, uint8_t use_threads,
    const uint8_t* data, uint64_t data_size, uint8_t* out, uint64_t out_capacity) {
  try {
    auto info = makeInfo(fields, n_fields, point_step, width, height, encoding_opt, compression_opt, version, use_threads);
    Cloudini::PointcloudEncoder encoder(info);
    Cloudini::ConstBufferView in(data, data_size);
    Cloudini::BufferView view(out, out_capacity);
    return static_cast<int64_t>(encoder.encode(in, view, true));
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

REF_API int64_t ref_max_compressed_size(
    const RefField* fields, uint32_t n_fields, uint32_t point_step, uint32_t width, uint32_t height,
    uint8_t encoding_opt, uint8_t compression_opt, uint8_t version, uint64_t n_points, uint8_t include_header) {
  try {
    auto info = makeInfo(fields, n_fields, point_step, width, height, encoding_opt, compression_opt, version, 0);
    return static_cast<int64_t>(Cloudini::MaxCompressedSize(info, n_points, include_header != 0));
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// Header bytes as written by the encoder (YAML form). Returns size or -1.
REF_API int64_t ref_encode_header(
    const RefField* fields, uint32_t n_fields, uint32_t point_step, uint32_t width, uint32_t height,
    uint8_t encoding_opt, uint8_t compression_opt, uint8_t version, uint8_t binary, uint8_t* out,
    uint64_t out_capacity) {
  try {
    auto info = makeInfo(fields, n_fields, point_step, width, height, encoding_opt, compression_opt, version, 0);
    std::vector<uint8_t> hdr;
    Cloudini::EncodeHeader(info, hdr, binary ? Cloudini::HeaderEncoding::BINARY : Cloudini::HeaderEncoding::YAML);
    if (hdr.size() > out_capacity) {
      g_last_error = "header buffer too small";
      return -1;
    }
    memcpy(out, hdr.data(), hdr.size());
    return static_cast<int64_t>(hdr.size());
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// Decode a full stream (header + chunks) into out (width*height*point_step bytes, caller pre-fills it so that
// "untouched padding" can be observed). Returns decoded byte count or -1. If yaml_out != nullptr the YAML
// rendering of the decoded header is copied there (NUL terminated, at most yaml_cap bytes).
REF_API int64_t ref_decode(
    const uint8_t* stream, uint64_t stream_size, uint8_t* out, uint64_t out_capacity, char* yaml_out,
    uint64_t yaml_cap) {
  try {
    Cloudini::ConstBufferView in(stream, stream_size);
    Cloudini::EncodingInfo info = Cloudini::DecodeHeader(in);
    if (yaml_out && yaml_cap) {
      const std::string yaml = Cloudini::EncodingInfoToYAML(info);
      const size_t n = std::min<size_t>(yaml.size(), yaml_cap - 1);
      memcpy(yaml_out, yaml.data(), n);
      yaml_out[n] = 0;
    }
    const uint64_t need = static_cast<uint64_t>(info.width) * info.height * info.point_step;
    if (need > out_capacity) {
      g_last_error = "decode buffer too small";
      return -1;
    }
    Cloudini::PointcloudDecoder decoder;
    Cloudini::BufferView view(out, need);
    decoder.decode(info, in, view);
    return static_cast<int64_t>(need);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// Decode with an explicitly supplied schema (no header in `data`), mirrors PointcloudDecoder::decode(info, ...).
REF_API int64_t ref_decode_noheader(
    const RefField* fields, uint32_t n_fields, uint32_t point_step, uint32_t width, uint32_t height,
    uint8_t encoding_opt, uint8_t compression_opt, uint8_t version, const uint8_t* data, uint64_t data_size,
    uint8_t* out, uint64_t out_capacity) {
  try {
    auto info = makeInfo(fields, n_fields, point_step, width, height, encoding_opt, compression_opt, version, 0);
    const uint64_t need = static_cast<uint64_t>(info.width) * info.height * info.point_step;
    if (need > out_capacity) {
      g_last_error = "decode buffer too small";
      return -1;
    }
    Cloudini::PointcloudDecoder decoder;
    Cloudini::ConstBufferView in(data, data_size);
    Cloudini::BufferView view(out, need);
    decoder.decode(info, in, view);
    return static_cast<int64_t>(need);
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// ---- ROS / CDR message path (ros_msg_utils.cpp) -------------------------------------------------------------

// sensor_msgs/PointCloud2 CDR bytes -> CompressedPointCloud2 CDR bytes; every FLOAT32 field gets `resolution`
// (as cldn_EncodePointcloudMessage / the ROS plugin would set), compression per argument. Returns size or -1.
REF_API int64_t ref_ros_compress(
    const uint8_t* dds, uint64_t dds_size, float resolution, uint8_t compression_opt, uint8_t* out,
    uint64_t out_capacity) {
  try {
    Cloudini::ConstBufferView in(dds, dds_size);
    auto pc = cloudini_ros::getDeserializedPointCloudMessage(in);
    cloudini_ros::applyResolutionProfile({}, pc.fields, resolution);
    auto info = cloudini_ros::toEncodingInfo(pc);
    info.compression_opt = static_cast<Cloudini::CompressionOption>(compression_opt);
    std::vector<uint8_t> msg;
    cloudini_ros::convertPointCloud2ToCompressedCloud(pc, info, msg);
    if (msg.size() > out_capacity) {
      g_last_error = "ros output too small";
      return -1;
    }
    memcpy(out, msg.data(), msg.size());
    return static_cast<int64_t>(msg.size());
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// CompressedPointCloud2 CDR bytes -> PointCloud2 CDR bytes.
REF_API int64_t ref_ros_decompress(const uint8_t* dds, uint64_t dds_size, uint8_t* out, uint64_t out_capacity) {
  try {
    Cloudini::ConstBufferView in(dds, dds_size);
    auto pc = cloudini_ros::getDeserializedPointCloudMessage(in);
    std::vector<uint8_t> msg;
    cloudini_ros::convertCompressedCloudToPointCloud2(pc, msg);
    if (msg.size() > out_capacity) {
      g_last_error = "ros output too small";
      return -1;
    }
    memcpy(out, msg.data(), msg.size());
    return static_cast<int64_t>(msg.size());
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// Parse a PointCloud2 CDR message; report schema as YAML-ish text plus the offset/size of the data blob.
REF_API int64_t ref_ros_describe(
    const uint8_t* dds, uint64_t dds_size, char* text_out, uint64_t text_cap, uint64_t* data_offset,
    uint64_t* data_size) {
  try {
    Cloudini::ConstBufferView in(dds, dds_size);
    auto pc = cloudini_ros::getDeserializedPointCloudMessage(in);
    auto info = cloudini_ros::toEncodingInfo(pc);
    std::string yaml = Cloudini::EncodingInfoToYAML(info);
    const size_t n = std::min<size_t>(yaml.size(), text_cap ? text_cap - 1 : 0);
    if (text_out && text_cap) {
      memcpy(text_out, yaml.data(), n);
      text_out[n] = 0;
    }
    *data_offset = static_cast<uint64_t>(pc.data.data() - dds);
    *data_size = pc.data.size();
    return static_cast<int64_t>(yaml.size());
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// applyVizLossyPreprocessing (ros_msg_utils.hpp:175-221, src/ros_msg_utils.cpp:249-341) on a bare point buffer.
// out receives the surviving points (capacity >= data_size); res_out[i] = resolution of field i afterwards
// (NaN = none), so that the FLOAT64 -> 1 us rule is visible. Returns the surviving byte count, or -1.
REF_API int64_t ref_viz_preprocess(
    const RefField* fields, uint32_t n_fields, uint32_t point_step, const uint8_t* data, uint64_t data_size,
    uint8_t* out, uint64_t out_capacity, float* res_out, uint32_t* width_out, uint32_t* height_out) {
  try {
    cloudini_ros::RosPointCloud2 pc;
    for (uint32_t i = 0; i < n_fields; ++i) {
      Cloudini::PointField f;
      f.name = fields[i].name ? fields[i].name : "";
      f.offset = fields[i].offset;
      f.type = static_cast<Cloudini::FieldType>(fields[i].type);
      if (fields[i].has_resolution) {
        f.resolution = fields[i].resolution;
      }
      pc.fields.push_back(f);
    }
    pc.point_step = point_step;
    pc.width = point_step ? static_cast<uint32_t>(data_size / point_step) : 0;
    pc.height = 1;
    pc.row_step = pc.width * point_step;
    pc.data = Cloudini::ConstBufferView(data, data_size);
    cloudini_ros::applyVizLossyPreprocessing(pc);
    if (pc.data.size() > out_capacity) {
      g_last_error = "ref_viz_preprocess: output capacity";
      return -1;
    }
    if (pc.data.size()) memcpy(out, pc.data.data(), pc.data.size());
    for (uint32_t i = 0; i < n_fields; ++i) {
      res_out[i] = pc.fields[i].resolution.has_value() ? pc.fields[i].resolution.value()
                                                       : std::numeric_limits<float>::quiet_NaN();
    }
    *width_out = pc.width;
    *height_out = pc.height;
    return static_cast<int64_t>(pc.data.size());
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

// ---- CPU baseline timing (same bracket as tools/src/mcap_codec_benchmark.cpp:447-457) ----------------------
// Encoder constructed outside the timed region; output pre-sized; `reps` timed encode() calls per thread on
// `threads` independent encoders. times_out[threads*reps] seconds. Returns encoded size or -1.
REF_API int64_t ref_bench_encode(
    const RefField* fields, uint32_t n_fields, uint32_t point_step, uint32_t width, uint32_t height,
    uint8_t encoding_opt, uint8_t compression_opt, uint8_t version, uint8_t use_threads, const uint8_t* data,
    uint64_t data_size, uint32_t reps, uint32_t threads, double* times_out) {
  try {
    auto info = makeInfo(fields, n_fields, point_step, width, height, encoding_opt, compression_opt, version, use_threads);
    const size_t points = data_size / point_step;
    const size_t cap = Cloudini::MaxCompressedSize(info, points, true);
    std::atomic<int64_t> result{0};
    std::atomic<bool> failed{false};
    std::atomic<uint32_t> ready{0};
    std::atomic<bool> go{false};
    auto worker = [&](uint32_t tid) {
      try {
        Cloudini::PointcloudEncoder encoder(info);
        std::vector<uint8_t> out(cap);
        Cloudini::ConstBufferView in(data, data_size);
        {  // warm-up (first-touch of buffers), untimed
          Cloudini::BufferView view(out.data(), out.size());
          result = static_cast<int64_t>(encoder.encode(in, view, true));
        }
        ready.fetch_add(1);
        while (!go.load()) {
          std::this_thread::yield();
        }
        for (uint32_t r = 0; r < reps; ++r) {
          Cloudini::BufferView view(out.data(), out.size());
          const auto t0 = std::chrono::steady_clock::now();
          const size_t sz = encoder.encode(in, view, true);
          const auto t1 = std::chrono::steady_clock::now();
          times_out[tid * reps + r] = std::chrono::duration<double>(t1 - t0).count();
          result = static_cast<int64_t>(sz);
        }
      } catch (const std::exception& e) {
        g_last_error = e.what();
        failed = true;
        ready.fetch_add(1);
      }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 0; t < threads; ++t) {
      pool.emplace_back(worker, t);
    }
    while (ready.load() < threads) {
      std::this_thread::yield();
    }
    go = true;
    for (auto& t : pool) {
      t.join();
    }
    return failed ? -1 : result.load();
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}

REF_API int64_t ref_bench_decode(
    const uint8_t* stream, uint64_t stream_size, uint8_t* out, uint64_t out_capacity, uint32_t reps,
    double* times_out) {
  try {
    Cloudini::PointcloudDecoder decoder;
    int64_t need = 0;
    for (uint32_t r = 0; r < reps + 1; ++r) {
      Cloudini::ConstBufferView in(stream, stream_size);
      const auto t0 = std::chrono::steady_clock::now();
      Cloudini::EncodingInfo info = Cloudini::DecodeHeader(in);
      need = static_cast<int64_t>(info.width) * info.height * info.point_step;
      if (static_cast<uint64_t>(need) > out_capacity) {
        g_last_error = "decode buffer too small";
        return -1;
      }
      Cloudini::BufferView view(out, need);
      decoder.decode(info, in, view);
      const auto t1 = std::chrono::steady_clock::now();
      if (r > 0) {
        times_out[r - 1] = std::chrono::duration<double>(t1 - t0).count();
      }
    }
    return need;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return -1;
  }
}
