// host_layer_test.cpp — CPU-tier checks of the C++ host layer (csrc/host/mtgpu_host.hpp):
// configuration getters, TaskQueue / ResultCollector / JobQueue semantics, the .mtmv reader and
// its backward-seek rule, and the loud failure of GpuMotionScanner without a GPU.
// Built and run by tests/test_host_cpp.py:  host_layer_test <stream.mtmv>
// Prints "key value" lines that the Python test compares with what it wrote into the file.
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "mtgpu_host.hpp"

using namespace mtgpu_host;

#define REQUIRE(cond)                                                   \
  do {                                                                  \
    if (!(cond)) { std::printf("FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

int main(int argc, char **argv) {
  // ---- Config: same variables / defaults / casts as include/motion_trim/config.hpp:56-125
  for (const char *v : {"MV_THRESHOLD_SQ", "BLOCK_SIZE", "BLOCK_SHIFT", "VECTORS_NEEDED", "CLUSTERS_NEEDED",
                        "VERTICAL_MASK", "MAX_GAP_SEC", "PADDING_SEC", "CHUNK_DURATION_SEC", "TARGET_FPS",
                        "MIN_SAVINGS_PCT"})
    unsetenv(v);
  REQUIRE(Config::mv_threshold_sq() == 16.0 && Config::block_size() == 16 && Config::block_shift() == 4);
  REQUIRE(Config::vectors_needed() == 2 && Config::clusters_needed() == 2 && Config::vertical_mask() == 0.05f);
  REQUIRE(Config::max_gap_sec() == 5.0 && Config::padding_sec() == 0.5 && Config::chunk_duration_sec() == 30.0);
  REQUIRE(Config::target_fps() == 0.0 && Config::min_savings_pct() == 5.0);
  // read once per process (function-local statics, config.hpp:56-59): a later setenv changes nothing —
  // the uint8 cast and every parse corner are pinned by tests/test_reference_host.py (`config` / `memo`)
  setenv("VECTORS_NEEDED", "260", 1);
  setenv("MV_THRESHOLD_SQ", "4.0", 1);
  REQUIRE(Config::vectors_needed() == 2 && Config::mv_threshold_sq() == 16.0);
  // a bad value is a returned error of the pipeline, never an exception out of a worker thread
  setenv("MTGPU_BATCH_MB", "abc", 1);
  {
    PipelineResult r;
    int rc = run_scan_pipeline([]() -> std::unique_ptr<FrameSource> { throw std::runtime_error("unreachable"); }, 2, r);
    REQUIRE(rc != 0 && r.error.rfind("configuration:", 0) == 0);
  }
  unsetenv("MTGPU_BATCH_MB");

  // ---- TaskQueue: FIFO, pop blocks until push or finish, drained queue + finish -> false
  {
    TaskQueue q;
    std::vector<int> got;
    std::thread consumer([&] { ScanTask t; while (q.pop(t)) got.push_back(t.id); });
    for (int i = 0; i < 100; ++i) q.push({i * 1.0, i + 1.0, i});
    q.finish();
    consumer.join();
    REQUIRE(got.size() == 100);
    for (int i = 0; i < 100; ++i) REQUIRE(got[i] == i);
    ScanTask t;
    REQUIRE(!q.pop(t));
  }
  // ---- ResultCollector: pooled from many threads, extract empties it
  {
    ResultCollector rc;
    std::vector<std::thread> th;
    for (int k = 0; k < 8; ++k) th.emplace_back([&rc, k] { for (int r = 0; r < 50; ++r) rc.add(std::vector<double>{k + r / 100.0}); });
    for (auto &t : th) t.join();
    auto all = rc.extract();
    REQUIRE(all.size() == 400);
    REQUIRE(rc.extract().empty());
  }
  // ---- JobQueue: single consumer sees every job, then finish
  {
    JobQueue jq;
    int seen = 0;
    std::thread consumer([&] { ScanJob j; while (jq.pop(j)) seen += j.stream_id; });
    for (int i = 1; i <= 10; ++i) { ScanJob j; j.stream_id = i; jq.push(std::move(j)); }
    jq.finish();
    consumer.join();
    REQUIRE(seen == 55);
  }
  if (argc < 2) { std::printf("no file given\n"); return 0; }

  // ---- .mtmv reader
  MtmvFile file(argv[1]);
  MtmvSource src(file);
  std::printf("width %d\nheight %d\nfps %.6f\nduration %.6f\ntime_base_den %.0f\nframes %llu\nrecords %llu\n",
              src.width(), src.height(), src.fps(), src.duration(), 1.0 / src.time_base(),
              (unsigned long long)file.hdr->n_frames, (unsigned long long)file.hdr->n_records);
  Frame f;
  unsigned long long n = 0, with_sd = 0, bytes = 0;
  src.seek(0.0);
  long long first_pts = -1;
  while (src.next(f)) {
    if (n == 0) first_pts = f.pts;
    ++n;
    with_sd += f.has_side_data;
    bytes += f.mv_bytes;
    if (f.has_side_data && f.mv_bytes) {                       // records are readable AVMotionVector bytes
      const mt_mv *r = static_cast<const mt_mv *>(f.mv);
      REQUIRE(r->w > 0 && r->motion_scale == 4);
    }
  }
  std::printf("iterated %llu\nwith_sd %llu\nbytes %llu\nfirst_pts %lld\n", n, with_sd, bytes, first_pts);
  // backward seek: lands on the last keyframe at or before the target (motion_scanner.cpp:321-325)
  for (double s : {0.0, 0.2, 1.0, 1.49, 1.5, 2.26, 100.0}) {
    src.seek(s);
    REQUIRE(src.next(f));
    std::printf("seek %.2f -> pts %lld key %d\n", s, (long long)f.pts, f.has_side_data ? 0 : 1);
  }

  // ---- no GPU: the scanner refuses loudly (when a GPU is present this part is skipped)
  if (mtgpu_device_count() == 0) {
    GpuMotionScanner sc(src, 0);
    REQUIRE(!sc.initialize());
    std::printf("init_error %s\n", sc.error().c_str());
    PipelineResult r;
    int rc = run_scan_pipeline([&] { return std::unique_ptr<FrameSource>(new MtmvSource(file)); }, 2, r);
    REQUIRE(rc != 0 && !r.error.empty());
  } else {
    std::printf("init_error skipped (GPU present)\n");
  }
  std::printf("OK\n");
  return 0;
}
