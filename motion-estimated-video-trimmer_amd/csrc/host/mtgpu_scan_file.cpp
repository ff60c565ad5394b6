// mtgpu_scan_file — the scan + merge half of `motion_trim <in> <out>` on the GPU, reading
// extracted motion vectors from a .mtmv container instead of decoding with FFmpeg:
//   mtgpu_scan_file stream.mtmv [--threads N]
// Configuration comes from the same environment variables as the reference
// (MV_THRESHOLD_SQ, VECTORS_NEEDED, CHUNK_DURATION_SEC, TARGET_FPS, ...).  Prints one JSON
// object: the FFmpegJob segment list (%.17g doubles) and the merge result.
#include <cstdio>
#include <cstring>
#include <string>

#include "mtgpu_host.hpp"

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s stream.mtmv [--threads N]\n", argv[0]);
    return 2;
  }
  int threads = 4;
  for (int i = 2; i + 1 < argc; ++i)
    if (!std::strcmp(argv[i], "--threads")) threads = std::atoi(argv[i + 1]);
  try {
    mtgpu_host::MtmvFile file(argv[1]);
    mtgpu_host::PipelineResult r;
    int rc = mtgpu_host::run_scan_pipeline(
        [&] { return std::unique_ptr<mtgpu_host::FrameSource>(new mtgpu_host::MtmvSource(file)); }, threads, r);
    if (rc != 0) {
      std::fprintf(stderr, "error: %s\n", r.error.c_str());
      return 1;
    }
    std::printf("{\"chunks\": %d, \"threads\": %d, \"motion_frames\": %zu, \"n_timestamps\": %llu, "
                "\"do_cut\": %d, \"time_removed\": %.17g, \"saved_pct\": %.17g, \"seek_us\": %ld, "
                "\"decode_us\": %ld, \"analyze_us\": %ld, \"segments\": [",
                r.chunks, r.threads, r.motion_frames, (unsigned long long)r.merge.n_timestamps, r.merge.do_cut,
                r.merge.time_removed, r.merge.saved_pct, r.seek_us, r.decode_us, r.analyze_us);
    for (size_t i = 0; i < r.segments.size(); ++i)
      std::printf("%s[%.17g, %.17g]", i ? ", " : "", r.segments[i].start, r.segments[i].end);
    std::printf("]}\n");
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
