// cloudini_batch_transcode: directory of CDR sensor_msgs/PointCloud2 messages -> directory of CompressedPointCloud2
// messages, through the batched HIP encoder (include/cloudini_amd/batch_transcoder.hpp). The batched counterpart of the
// reference's cloudini_rosbag_converter encode loop (tools/src/mcap_converter.cpp:140-222) for containers that are
// plain directories.
//   cloudini_batch_transcode <in_dir> <out_dir> [--resolution 0.001] [--compression none|lz4|zstd] [--viz] [--batch 64]
//   cloudini_batch_transcode <in_dir> <out_dir> --decode [--batch 64]      (CompressedPointCloud2 -> PointCloud2)
//   ... --devices 0,1,2,3   spreads the batches over these GPUs (one GPU stage per entry; "0,0" = two stages on GPU 0)
//   cloudini_batch_transcode <in.mcap> <out.mcap> [...same options] [--mcap-compression none|lz4|zstd]
//       a bag: point-cloud messages converted, everything else copied (McapConverter, tools/src/mcap_converter.cpp:141-300);
//       read as a stream, a chunk at a time. CLDN_DEBUG_MEM=1 prints the process's memory high-water marks to stderr
//       (VmHWM of /proc/self/status; tests/test_mcap_io.py checks that they do not follow the size of the bag).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "cloudini_amd/batch_transcoder.hpp"
#include "cloudini_amd/mcap_io.hpp"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <in_dir> <out_dir> [--resolution r] [--compression none|lz4|zstd] [--viz] [--batch n] [--devices 0,1,...] | --decode [--batch n] [--devices ...]\n", argv[0]);
    return 2;
  }
  cloudini_amd::TranscodeOptions opt;
  cloudini_amd::McapCompression mcap_comp = cloudini_amd::McapCompression::Zstd;
  for (int i = 3; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "--resolution" && i + 1 < argc) opt.default_resolution = std::strtof(argv[++i], nullptr);
    else if (a == "--compression" && i + 1 < argc) opt.compression = Cloudini::CompressionOptionFromString(
        std::string(argv[i + 1]) == "none" ? "NONE" : (std::string(argv[i + 1]) == "lz4" ? "LZ4" : "ZSTD")), ++i;
    else if (a == "--viz") opt.viz_lossy = true;
    else if (a == "--mcap-compression" && i + 1 < argc) {
      const std::string v = argv[++i];
      mcap_comp = v == "none" ? cloudini_amd::McapCompression::None : (v == "lz4" ? cloudini_amd::McapCompression::Lz4 : cloudini_amd::McapCompression::Zstd);
    }
    else if (a == "--decode") opt.decode = true;
    else if (a == "--batch" && i + 1 < argc) opt.batch_messages = (size_t)std::strtoul(argv[++i], nullptr, 10);
    else if (a == "--devices" && i + 1 < argc) {
      for (const char* p = argv[++i]; *p;) {
        char* end = nullptr;
        opt.devices.push_back((int)std::strtol(p, &end, 10));
        if (end == p) {
          std::fprintf(stderr, "--devices wants a comma-separated list of device numbers\n");
          return 2;
        }
        p = *end == ',' ? end + 1 : end;
      }
    }
    else {
      std::fprintf(stderr, "unknown argument %s\n", a.c_str());
      return 2;
    }
  }
  try {
    const std::string in_path = argv[1];
    if (in_path.size() > 5 && in_path.compare(in_path.size() - 5, 5, ".mcap") == 0) {
      const cloudini_amd::McapTranscodeStats ms = cloudini_amd::transcodeMcap(in_path, argv[2], opt, mcap_comp);
      std::printf("{\"messages\": %llu, \"converted\": %llu, \"input_bytes\": %llu, \"output_bytes\": %llu, \"points\": %llu, "
                  "\"seconds_total\": %.6f, \"gpu_batches\": %llu, \"peak_held_bytes\": %llu}\n",
                  (unsigned long long)ms.messages, (unsigned long long)ms.converted, (unsigned long long)ms.input_bytes,
                  (unsigned long long)ms.output_bytes, (unsigned long long)ms.pipeline.points, ms.pipeline.seconds_total,
                  (unsigned long long)ms.pipeline.gpu_batches, (unsigned long long)ms.peak_held_bytes);
      if (std::getenv("CLDN_DEBUG_MEM")) {  // diagnostics: the process's memory high-water marks
        if (FILE* f = std::fopen("/proc/self/status", "r")) {
          char line[256];
          while (std::fgets(line, sizeof line, f))
            if (!std::strncmp(line, "VmHWM", 5) || !std::strncmp(line, "Rss", 3) || !std::strncmp(line, "VmPeak", 6)) std::fputs(line, stderr);
          std::fclose(f);
        }
      }
      return 0;
    }
    cloudini_amd::DirectorySource source(argv[1]);
    cloudini_amd::DirectorySink sink(argv[2]);
    const cloudini_amd::TranscodeStats st = cloudini_amd::transcodePointClouds(source, sink, opt);
    std::printf("{\"messages\": %llu, \"points\": %llu, \"input_bytes\": %llu, \"output_bytes\": %llu, \"gpu_batches\": %llu, "
                "\"seconds_total\": %.6f, \"seconds_gpu\": %.6f, \"seconds_stage2\": %.6f, \"gpu_stages\": %llu, \"Mpoints_per_s\": %.1f}\n",
                (unsigned long long)st.messages, (unsigned long long)st.points, (unsigned long long)st.input_bytes,
                (unsigned long long)st.output_bytes, (unsigned long long)st.gpu_batches, st.seconds_total, st.seconds_gpu,
                st.seconds_stage2, (unsigned long long)st.gpu_workers, st.seconds_total > 0 ? st.points / st.seconds_total / 1e6 : 0.0);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "cloudini_batch_transcode: %s\n", e.what());
    return 1;
  }
  return 0;
}
