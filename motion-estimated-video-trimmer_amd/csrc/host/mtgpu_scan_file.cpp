// mtgpu_scan_file — the scan + merge half of `motion_trim` on the GPU, reading extracted motion
// vectors from .mtmv containers instead of decoding with FFmpeg:
//   mtgpu_scan_file stream.mtmv [more.mtmv ...] [--threads T] [--streams S] [--outdir DIR] [--timestamps] [--summary]
//   (--streams 0 / --threads 0: the reference's own sizing from PARALLEL_STREAMS / THREADS_PER_STREAM and the CPU limit)
// One file: like `motion_trim in out` (single ProcessingPipeline).  Several files: like
// `motion_trim in_dir out_dir` (BatchProcessor): S streams x T workers, jobs consumed by one
// thread (here: printed).  Configuration comes from the same environment variables as the
// reference (MV_THRESHOLD_SQ, VECTORS_NEEDED, CHUNK_DURATION_SEC, TARGET_FPS, ...).
// Prints one JSON object per input with a job: the FFmpegJob segment list (%.17g) + merge result.
// --summary (several files): one more line {"batch_summary": ...} — frames scanned, wall time, worker-time
// breakdown and what the S x T workers held (contexts, pipes, HIP streams, pinned / device bytes).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "mtgpu_host.hpp"

using namespace mtgpu_host;

// --repeat N (rate measurements only): the input presented as N back-to-back copies of itself — a
// long video whose MV bytes stay cache-resident, like side data a decoder thread has just written,
// instead of 10+ GB streamed from the page cache.
class RepeatSource : public FrameSource {
  const MtmvFile &f_;
  uint64_t reps_, pos_ = 0;
  int64_t period_;     // ticks per copy
 public:
  RepeatSource(const MtmvFile &f, uint64_t reps) : f_(f), reps_(reps) {
    period_ = (int64_t)std::llround(f.hdr->duration * (double)f.hdr->tb_den / (double)f.hdr->tb_num);
  }
  int width() const override { return (int)f_.hdr->width; }
  int height() const override { return (int)f_.hdr->height; }
  double duration() const override { return f_.hdr->duration * (double)reps_; }
  double fps() const override { return f_.hdr->fps; }
  double time_base() const override { return (double)f_.hdr->tb_num / (double)f_.hdr->tb_den; }
  void seek(double seconds) override {
    const int64_t target = static_cast<int64_t>(seconds / time_base());
    uint64_t rep = period_ > 0 ? (uint64_t)(target / period_) : 0;
    if (rep >= reps_) rep = reps_ - 1;
    const int64_t local = target - (int64_t)rep * period_;
    uint64_t key = 0;
    for (uint64_t i = 0; i < f_.hdr->n_frames && f_.frames[i].pts <= local; ++i)
      if (f_.frames[i].key) key = i;
    pos_ = rep * f_.hdr->n_frames + key;
  }
  bool next(Frame &fr) override {
    const uint64_t n = f_.hdr->n_frames;
    if (n == 0 || pos_ >= reps_ * n) return false;
    const uint64_t rep = pos_ / n;
    const MtmvFrameRec &r = f_.frames[pos_ % n];
    ++pos_;
    fr.pts = r.pts + (int64_t)rep * period_;
    fr.has_side_data = r.has_sd != 0;
    fr.mv = r.has_sd ? f_.records + 40ull * r.rec_off : nullptr;
    fr.mv_bytes = r.has_sd ? 40ull * r.n_rec : 0;
    return true;
  }
};

static bool g_summary = false;    // --summary
static bool g_print_ts = false;   // --timestamps: also print the pooled motion timestamps, sorted (%.17g)

static void print_job(const std::string &input, const PipelineResult &r, const std::vector<mt_segment> &segs) {
  std::printf("{\"input\": \"%s\", \"chunks\": %d, \"threads\": %d, \"frames_scanned\": %llu, \"motion_frames\": %zu, \"n_timestamps\": %llu, "
              "\"do_cut\": %d, \"time_removed\": %.17g, \"saved_pct\": %.17g, \"seek_us\": %ld, "
              "\"decode_us\": %ld, \"analyze_us\": %ld, \"init_us\": %ld, \"scan_wall_us\": %ld, \"scan_work_us\": %ld, "
              "\"copy_us\": %ld, \"submit_us\": %ld, \"wait_us\": %ld, \"segments\": [",
              input.c_str(), r.chunks, r.threads, (unsigned long long)r.frames_scanned, r.motion_frames,
              (unsigned long long)r.merge.n_timestamps,
              r.merge.do_cut, r.merge.time_removed, r.merge.saved_pct, r.seek_us, r.decode_us, r.analyze_us, r.init_us,
              r.scan_wall_us, r.scan_work_us, r.copy_us, r.submit_us, r.wait_us);
  for (size_t i = 0; i < segs.size(); ++i)
    std::printf("%s[%.17g, %.17g]", i ? ", " : "", segs[i].start, segs[i].end);
  std::printf("]");
  if (g_print_ts) {
    std::vector<double> ts = r.timestamps;
    std::sort(ts.begin(), ts.end());
    std::printf(", \"timestamps\": [");
    for (size_t i = 0; i < ts.size(); ++i) std::printf("%s%.17g", i ? ", " : "", ts[i]);
    std::printf("]");
  }
  std::printf("}\n");
  std::fflush(stdout);
}

int main(int argc, char **argv) {
  std::vector<std::string> files;
  int threads = 4, streams = 2;
  long repeat = 1;
  std::string outdir = ".";
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--threads") && i + 1 < argc) threads = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--streams") && i + 1 < argc) streams = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--outdir") && i + 1 < argc) outdir = argv[++i];
    else if (!std::strcmp(argv[i], "--timestamps")) g_print_ts = true;
    else if (!std::strcmp(argv[i], "--summary")) g_summary = true;
    else if (!std::strcmp(argv[i], "--repeat") && i + 1 < argc) repeat = std::atol(argv[++i]);
    else files.push_back(argv[i]);
  }
  // --streams 0 / --threads 0: sized from the CPU budget, the devices and the number of videos (default_batch_sizing,
  // mtgpu_host.hpp — deliberately not the reference's CPU-only rule, src/system.cpp:186-197); PARALLEL_STREAMS /
  // THREADS_PER_STREAM are honoured as in the reference (config.hpp:138-141, 165-168)
  if (streams <= 0 || threads <= 0) {
    try {
      const BatchSizing z = default_batch_sizing((int)files.size(), mtgpu_device_count(), cpu_budget(),
                                                 streams > 0 ? streams : Config::parallel_streams(),
                                                 threads > 0 ? threads : Config::threads_per_stream());
      if (streams <= 0) streams = z.streams;
      if (threads <= 0) threads = z.threads;
    } catch (const std::exception &e) {
      std::fprintf(stderr, "error: configuration: %s\n", e.what());
      return 1;
    }
  }
  if (files.empty()) {
    std::fprintf(stderr, "usage: %s stream.mtmv [more.mtmv ...] [--threads T] [--streams S] [--outdir DIR]\n", argv[0]);
    return 2;
  }
  try {
    if (files.size() == 1) {
      MtmvFile file(files[0]);
      PipelineResult r;
      int rc = run_scan_pipeline([&]() -> std::unique_ptr<FrameSource> {
        if (repeat > 1) return std::unique_ptr<FrameSource>(new RepeatSource(file, (uint64_t)repeat));
        return std::unique_ptr<FrameSource>(new MtmvSource(file));
      }, threads, r);
      if (rc != 0) { std::fprintf(stderr, "error: %s\n", r.error.c_str()); return 1; }
      print_job(files[0], r, r.segments);
      return 0;
    }
    // batch: mmaps are shared by the workers of a stream and kept until the end
    std::mutex mm;
    std::map<std::string, std::shared_ptr<MtmvFile>> open_files;
    auto open_source = [&](const std::string &path) {
      std::shared_ptr<MtmvFile> f = std::make_shared<MtmvFile>(path);   // mapped (and pre-faulted) outside the lock:
      { std::lock_guard<std::mutex> l(mm); open_files[path] = f; }       // 64 streams open their files concurrently
      return [f, repeat]() -> std::unique_ptr<FrameSource> {
        if (repeat > 1) return std::unique_ptr<FrameSource>(new RepeatSource(*f, (uint64_t)repeat));
        return std::unique_ptr<FrameSource>(new MtmvSource(*f));
      };
    };
    JobQueue jobs;
    std::vector<std::string> errors;
    int failed = 0;
    BatchSummary sum;
    std::thread producer([&] { failed = process_batch(files, outdir, streams, threads, open_source, jobs, &errors, &sum); });
    ScanJob job;                                   // the single consumer (batch_processor.cpp:138-150)
    while (jobs.pop(job)) print_job(job.input_path, job.result, job.segments);
    producer.join();
    if (g_summary) {
      const Resources &h = sum.held;
      std::printf("{\"batch_summary\": {\"streams\": %d, \"threads_per_stream\": %d, \"videos\": %zu, \"jobs\": %zu, "
                  "\"failed\": %zu, \"frames_scanned\": %llu, \"wall_us\": %ld, \"scan_wall_us\": %ld, \"scan_window_us\": %ld, \"init_us\": %ld, \"decode_us\": %ld, "
                  "\"analyze_us\": %ld, \"copy_us\": %ld, \"submit_us\": %ld, \"wait_us\": %ld, \"cpu_user_us\": %ld, \"cpu_sys_us\": %ld, \"worker_cpu_us\": %ld, \"gate_wait_us\": %ld, \"gate_tokens\": %d, \"cpu_window\": %d, \"cpu_window_first\": %d, "
                  "\"held\": {\"contexts\": %llu, \"pipes\": %llu, \"hip_streams\": %llu, \"hip_events\": %llu, "
                  "\"mem_pools\": %llu, \"pinned_bytes\": %llu, \"device_bytes\": %llu, \"scratch_pool_high_bytes\": %llu, "
                  "\"submits\": %llu, \"ctx_create_us\": %llu, \"pipe_create_us\": %llu, \"pipe_rebuilds\": %llu, \"pin_us\": %llu, \"pinned_batches\": %llu}}}\n",
                  sum.streams, sum.threads_per_stream, sum.videos, sum.jobs, sum.failed,
                  (unsigned long long)sum.frames_scanned, sum.wall_us, sum.scan_wall_us, sum.scan_window_us, sum.init_us, sum.decode_us, sum.analyze_us,
                  sum.copy_us, sum.submit_us, sum.wait_us, sum.cpu_user_us, sum.cpu_sys_us, sum.worker_cpu_us, sum.gate_wait_us, sum.gate_tokens, sum.cpu_window, sum.cpu_window_first, (unsigned long long)h.contexts, (unsigned long long)h.pipes,
                  (unsigned long long)h.hip_streams, (unsigned long long)h.hip_events, (unsigned long long)h.mem_pools,
                  (unsigned long long)h.pinned_bytes, (unsigned long long)h.device_bytes,
                  (unsigned long long)h.pool_reserved_high, (unsigned long long)h.submits,
                  (unsigned long long)h.ctx_create_us, (unsigned long long)h.pipe_create_us,
                  (unsigned long long)h.pipe_rebuilds, (unsigned long long)h.pin_us, (unsigned long long)h.pinned_batches);
      std::fflush(stdout);
    }
    for (auto &e : errors) std::fprintf(stderr, "error: %s\n", e.c_str());
    return failed ? 1 : 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
