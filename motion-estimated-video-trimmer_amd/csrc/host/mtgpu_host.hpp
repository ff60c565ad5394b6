// mtgpu_host.hpp — C++ host layer above the C ABI (include/mtgpu.h): the re-plumbed
// scan half of the reference's per-video pipeline.
//
//   reference (file:line in the reference tree)           here
//   ------------------------------------------------      -----------------------------------
//   Config:: getters (config.hpp:56-125)                   mtgpu_host::Config
//   ScanTask / TaskQueue (types.hpp:92-96, task_queue.*)   ScanTask / TaskQueue
//   ResultCollector (task_queue.cpp:43-57)                 ResultCollector
//   MotionScanner::initialize / scan_range                 GpuMotionScanner (decode stays behind
//     (motion_scanner.cpp:62-202, 297-391)                   the FrameSource interface)
//   worker loop + merge of ProcessingPipeline::run         run_scan_pipeline
//     (pipeline.cpp:130-167, 186-235, 297-358, 387-388)
//
// FFmpeg demux/decode is NOT here: a FrameSource hands over already-decoded frames
// (pts + MV side-data bytes).  A libav-backed FrameSource is a dozen lines around the
// reference's own decode loop (motion_scanner.cpp:334-354); MtmvSource below reads the
// repo's binary MV container so the whole path runs on boxes without FFmpeg.
#pragma once

#include <algorithm>
#include <atomic>
#include <charconv>
#include <chrono>
#include <ctime>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <mutex>
#include <queue>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include <dirent.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mtgpu.h"

namespace mtgpu_host {

// ---------------------------------------------------------------- configuration
struct Config {   // same variables, defaults and parse types as config.hpp:56-125
  static double env_d(const char *n, double d) { const char *v = std::getenv(n); return v ? std::stod(v) : d; }
  static int env_i(const char *n, int d) { const char *v = std::getenv(n); return v ? std::stoi(v) : d; }
  static float env_f(const char *n, float d) { const char *v = std::getenv(n); return v ? std::stof(v) : d; }
  // Every getter reads and parses its variable ONCE per process, into a function-local static, exactly as
  // config.hpp:56-59 does: later setenv() calls do not change the answer, initialisation is thread-safe
  // (workers of N x S threads call these), and a value that does not parse throws from the initialiser —
  // the static then stays uninitialised and the next call parses again (C++ [stmt.dcl]), as in the reference.
  // Pinned by the reference's own config.hpp through the `memo` scripts of tests/test_reference_host.py.
  static double mv_threshold_sq() { static double v = env_d("MV_THRESHOLD_SQ", 16.0); return v; }
  static int block_size() { static int v = env_i("BLOCK_SIZE", 16); return v; }
  static int block_shift() { static int v = env_i("BLOCK_SHIFT", 4); return v; }
  static int vectors_needed() { static uint8_t v = static_cast<uint8_t>(env_i("VECTORS_NEEDED", 2)); return v; }
  static int clusters_needed() { static int v = env_i("CLUSTERS_NEEDED", 2); return v; }
  static float vertical_mask() { static float v = env_f("VERTICAL_MASK", 0.05f); return v; }
  static double max_gap_sec() { static double v = env_d("MAX_GAP_SEC", 5.0); return v; }
  static double padding_sec() { static double v = env_d("PADDING_SEC", 0.5); return v; }
  static double chunk_duration_sec() { static double v = env_d("CHUNK_DURATION_SEC", 30.0); return v; }
  static double target_fps() { static double v = env_d("TARGET_FPS", 0.0); return v; }
  static double min_savings_pct() { static double v = env_d("MIN_SAVINGS_PCT", 5.0); return v; }
  static int parallel_streams() { static int v = env_i("PARALLEL_STREAMS", 0); return v; }      // config.hpp:138-141, 0 = auto
  static int threads_per_stream() { static int v = env_i("THREADS_PER_STREAM", 0); return v; }  // config.hpp:165-168, 0 = auto
  // not in the reference: staging layout of the host dispatcher (include/mtgpu.h), "aos40" or "compact8"
  // ("aos40", "compact8", either with the suffix "_zc" for zero-copy); read once like the rest
  static int staging_layout() {
    static int v = [] {
      const char *e = std::getenv("MTGPU_STAGING");
      const std::string s = e ? e : "compact8_zc";
      int layout = s.rfind("aos40", 0) == 0 ? MT_LAYOUT_AOS40 : MT_LAYOUT_COMPACT8;
      if (s.size() > 3 && s.compare(s.size() - 3, 3, "_zc") == 0) layout |= MT_LAYOUT_ZERO_COPY;
      return layout;
    }();
    return v;
  }
  // not in the reference: pinned staging per batch of the host dispatcher, in MiB.  Default 16 for either layout
  // (round 3, 64 concurrent streams on one device, steady-state window over 614 000 frames per run: 16 MiB batches
  // scan 4-24 % (64 x 1 workers), 45-58 % (16 x 4) and 45 % (4 x 16) more frames/s than 4 MiB ones — a batch costs a
  // launch, an event and a wake-up whatever its size — at 48 MiB of pinned memory per worker;
  // profiles/r03_host_batch64_ab2.json.  One hot stream with 16 workers does not care.)
  static int batch_mib() {
    static int v = std::max(1, env_i("MTGPU_BATCH_MB", 16));
    return v;
  }
  // Parse everything the scan path reads, in the calling thread: a value that does not parse surfaces
  // here as an exception (caught by run_scan_pipeline -> PipelineResult::error) instead of inside a worker.
  static void load_all() {
    (void)mv_threshold_sq(); (void)block_size(); (void)block_shift(); (void)vectors_needed(); (void)clusters_needed();
    (void)vertical_mask(); (void)max_gap_sec(); (void)padding_sec(); (void)chunk_duration_sec(); (void)target_fps();
    (void)min_savings_pct(); (void)staging_layout(); (void)batch_mib();
  }
};

// ---------------------------------------------------------------- CPU budget
// CPUs' worth of run TIME the scheduler grants this process: the cgroup quota (v2 cpu.max, v1 cfs quota), rounded
// up; without a quota the CPUs it may run on.  This is what the CpuGate, the CPU windows and the default batch sizing
// below are sized from.  It is deliberately NOT the reference's detect_cpu_limit(): that function takes the LARGER
// of the quota and the cpuset's CPU count (src/system.cpp:155-161) — on a box that shows 256 CPUs to a job with a
// 16-CPU quota the reference's own object code answers 256 (profiles/r04_sizing_on_gpu_box.txt), and 256 runnable
// workers are exactly what the quota throttles.
inline int cpu_budget() {
  auto read_two = [](const char *path, long &a, long &b) {
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    char q[64] = "", p[64] = "";
    const int n = std::fscanf(f, "%63s %63s", q, p);
    std::fclose(f);
    if (n < 1 || !std::strcmp(q, "max")) return false;
    a = std::atol(q);
    b = n >= 2 ? std::atol(p) : 0;
    return true;
  };
  long quota = 0, period = 0;
  if (read_two("/sys/fs/cgroup/cpu.max", quota, period) && quota > 0 && period > 0)
    return (int)((quota + period - 1) / period);
  long dummy = 0;
  if (read_two("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", quota, dummy) && read_two("/sys/fs/cgroup/cpu/cpu.cfs_period_us", period, dummy) &&
      quota > 0 && period > 0)
    return (int)((quota + period - 1) / period);
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) return CPU_COUNT(&set);
  const unsigned hc = std::thread::hardware_concurrency();
  return hc ? (int)hc : 1;
}

// ---------------------------------------------------------------- how many streams and workers
// Sizing of a batch run when the caller gives no counts (mtgpu_scan_file --streams 0 / --threads 0).  This is NOT
// the reference's rule: `motion_trim in_dir out_dir` opens min(PARALLEL_STREAMS, detect_cpu_limit()) streams with
// CPUs / streams threads each (src/system.cpp:186-197, src/batch_processor.cpp:81-95) because there every stream
// thread scans on a CPU of its own.  Here a worker only decodes and copies out — the scan runs on a GPU — so:
//   streams  PARALLEL_STREAMS when set (> 0), never more than there are videos; a stream count above the CPU budget
//            is legal (BASELINE config 4: 64 streams on a 16-CPU quota): the CpuGate, not the stream count, bounds
//            the runnable threads.  Auto: one stream per CPU of the budget, at least one per device.
//   threads  THREADS_PER_STREAM when set (> 0); auto: two workers per CPU of the budget over all streams (one fills a
//            batch while the other waits for the GPU: measured flat from 4 workers per hot stream on),
//            at least one per stream.
// What the reference itself would choose on a machine is printed by ITS object code (oracle/_ref/ref_host_probe
// `sizing`, built from src/system.cpp where it lies; tests/test_reference_host.py keeps that comparison).
struct BatchSizing { int streams = 1, threads = 1; };
inline BatchSizing default_batch_sizing(int n_videos, int n_devices, int budget, int configured_streams, int configured_threads) {
  BatchSizing z;
  n_videos = std::max(1, n_videos);
  n_devices = std::max(1, n_devices);
  budget = std::max(1, budget);
  z.streams = configured_streams > 0 ? configured_streams : std::max(budget, n_devices);
  z.streams = std::max(1, std::min(z.streams, n_videos));
  z.threads = configured_threads > 0 ? configured_threads : std::max(1, (2 * budget + z.streams - 1) / z.streams);
  return z;
}

// "0-3,8,10-11" (sysfs cpulists, MTGPU_CPU_WINDOW) -> CPU numbers in the order written, duplicates dropped.
// Anything that is not a list of non-negative numbers and ranges throws std::invalid_argument.
inline std::vector<int> parse_cpu_list(const std::string &text) {
  std::vector<int> out;
  std::vector<bool> seen;
  const char *p = text.c_str();
  auto skip = [&] { while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r') ++p; };
  auto number = [&]() -> long {
    skip();
    if (*p < '0' || *p > '9') throw std::invalid_argument("cpu list: number expected");
    char *end = nullptr;
    const long v = std::strtol(p, &end, 10);
    if (v < 0 || v > 1 << 20) throw std::invalid_argument("cpu list: number out of range");
    p = end;
    skip();
    return v;
  };
  skip();
  while (*p) {
    const long lo = number();
    long hi = lo;
    if (*p == '-') { ++p; hi = number(); }
    if (hi < lo) throw std::invalid_argument("cpu list: descending range");
    if ((size_t)hi >= seen.size()) seen.resize((size_t)hi + 1, false);       // (a bitmap: "0-1048576" must not cost n^2 look-ups)
    for (long c = lo; c <= hi; ++c)
      if (!seen[(size_t)c]) { seen[(size_t)c] = true; out.push_back((int)c); }
    if (*p == ',') { ++p; skip(); if (!*p) throw std::invalid_argument("cpu list: trailing comma"); }
    else if (*p) throw std::invalid_argument("cpu list: ',' expected");
  }
  return out;
}

// The CPUs this PROCESS may run on: the affinity of its main thread (tid == pid), not of the calling thread —
// worker threads pin themselves to their device's window and every thread they start inherits that mask, so a
// worker asking on behalf of another device would otherwise see only its own window.  Read once.
inline const std::vector<int> &process_allowed_cpus() {
  static const std::vector<int> cpus = [] {
    std::vector<int> v;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(getpid(), sizeof set, &set) == 0)
      for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &set)) v.push_back(c);
    return v;
  }();
  return cpus;
}

// ---------------------------------------------------------------- where the worker threads run
// The reference pins every stream thread to its own CPU set (src/batch_processor.cpp:102-110, pin_thread_to_cpus
// src/system.cpp:211-225).  With GPUs in place of CPU sets the workers of a device share ONE window of CPUs next to
// that device instead — and the window is sized from the CPU budget, for a measured reason: under a cgroup quota the
// kernel hands run time to CPUs in slices, so 64 workers that wake up on 64 different CPUs of a 256-CPU box drain a
// 16-CPU quota in bursts and the whole group is throttled for the rest of the period (28 of 40 periods), while the
// GPU — its submitters descheduled — sat idle 41 % of the time (profiles/r04_host_feed_kernel_view.txt).  A first
// trial with `taskset` (profiles/r04_host_feed_cpu_window.txt): confined to 24 CPUs the same run scanned 171-174 k
// frames/s instead of 141-145 k; to 16 CPUs 120 k (too few to also run the runtime's own threads), to 32 CPUs 155-159 k.
//
// pick_cpu_window: pure arithmetic (tested on the CPU tier).  `local` = the CPUs next to the device in sysfs order
// (e.g. "64-127,192-255": a socket's cores, then their SMT siblings), `gpu_index` of `gpus_on_node` GPUs share
// them: the cores are cut into equal parts, the window starts at part gpu_index and takes `want` physical cores from
// there on (into the neighbours' parts when a part is smaller, wrapping inside the node); SMT siblings only when the
// node has fewer cores than wanted.
inline std::vector<int> pick_cpu_window(const std::vector<int> &local, int gpu_index, int gpus_on_node, int want) {
  std::vector<int> out;
  if (local.empty() || want <= 0) return out;
  gpus_on_node = std::max(1, gpus_on_node);
  gpu_index = std::min(std::max(0, gpu_index), gpus_on_node - 1);
  // contiguous ranges of the list, in order
  std::vector<std::pair<size_t, size_t>> ranges;          // [begin, end) into `local`
  size_t b = 0;
  for (size_t i = 1; i <= local.size(); ++i)
    if (i == local.size() || local[i] != local[i - 1] + 1) { ranges.push_back({b, i}); b = i; }
  const size_t first_len = ranges[0].second - ranges[0].first;
  // "cores, then their SMT siblings": exactly two equally long ranges (a node's sysfs local_cpulist, "64-127,192-255")
  const bool smt_layout = ranges.size() == 2 && first_len >= 2 && ranges[1].second - ranges[1].first == first_len;
  const size_t pool_len = smt_layout ? first_len : local.size();          // the physical cores
  const size_t per = std::max<size_t>(1, pool_len / (size_t)gpus_on_node);
  const size_t lo = std::min(pool_len - 1, per * (size_t)gpu_index);
  // `want` CORES starting at this GPU's part and running on into its neighbours' (wrapping inside the node): two
  // hardware threads of one core are not two CPUs' worth of copy-out, and a window wider than a part only overlaps
  // CPUs its neighbour cannot fill either under the same quota
  const size_t n_cores = std::min<size_t>((size_t)want, pool_len);
  for (size_t i = 0; i < n_cores; ++i) out.push_back(local[(lo + i) % pool_len]);
  if (smt_layout)                                            // more wanted than the node has cores: their siblings
    for (size_t i = 0; (int)out.size() < want && i < first_len; ++i) out.push_back(local[ranges[1].first + (lo + i) % pool_len]);
  return out;
}

// The window for the workers of `device`, or empty = do not pin.  MTGPU_CPU_WINDOW: "off" / "0" = never pin;
// a number = that many CPUs; a cpulist ("64-87") = exactly those; unset = ceil(1.25 x cpu_budget()) cores, and only when
// the budget is a real restriction (fewer CPUs' worth of time than CPUs the process may run on).  Measured on a 16-CPU
// quota, three interleaved passes (profiles/r04_host_feed_ab_window.json): no window 147-157 k (64 x 1) / 139-145 k
// (16 x 4) frames/s; 20 cores 163-166 k / 179-182 k every time; 24 cores 124-174 k / 142-183 k (two modes); 32 cores
// 162-166 k / 165-166 k, throttled again; a window made of 16 cores + 8 of their own SMT siblings lost to no window at
// 64 x 1 (r04_host_feed_ab_window_smt_siblings.json), hence physical cores only.
inline std::vector<int> cpu_window_for_device(int device) {
  static std::mutex mu;
  static std::vector<std::pair<int, std::vector<int>>> cache;
  std::lock_guard<std::mutex> l(mu);
  for (auto &c : cache) if (c.first == device) return c.second;
  std::vector<int> win;
  try {
    // (the PROCESS's CPUs: a worker already pinned to its own device's window may be the first to ask for another
    //  device's — with the calling thread's mask every later device came back empty, for the life of the process)
    const std::vector<int> &allowed = process_allowed_cpus();
    const char *e = std::getenv("MTGPU_CPU_WINDOW");
    const std::string env = e ? e : "";
    int want = 0;
    if (env == "off" || env == "0") want = 0;
    else if (!env.empty() && env.find_first_of(",-") != std::string::npos) {
      for (int c : parse_cpu_list(env)) if (std::find(allowed.begin(), allowed.end(), c) != allowed.end()) win.push_back(c);
      cache.push_back({device, win});
      return win;
    } else if (!env.empty()) want = std::max(0, std::atoi(env.c_str()));
    else {
      const int budget = cpu_budget();
      want = budget < (int)allowed.size() ? (5 * budget + 3) / 4 : 0;
    }
    if (want > 0 && want < (int)allowed.size()) {
      std::vector<int> local = allowed;
      int gpu_index = 0, gpus_on_node = 1;
      char addr[32] = "";
      if (mtgpu_device_pci_address(device, addr, sizeof addr) == MT_OK) {
        const std::string dir = std::string("/sys/bus/pci/devices/") + addr;
        std::vector<int> near;
        std::string cpulist;
        { std::ifstream f(dir + "/local_cpulist"); if (f) std::getline(f, cpulist); }
        for (int c : parse_cpu_list(cpulist))
          if (std::find(allowed.begin(), allowed.end(), c) != allowed.end()) near.push_back(c);
        if ((int)near.size() >= want) {
          local = near;
          // the GPUs that share these CPUs: AMD devices of the same class on the same NUMA node, in address order
          auto slurp = [](const std::string &p) { std::ifstream f(p); std::string t; if (f) std::getline(f, t); return t; };
          const std::string my_node = slurp(dir + "/numa_node"), my_class = slurp(dir + "/class");
          std::vector<std::string> peers;
          if (DIR *d = opendir("/sys/bus/pci/devices")) {
            while (dirent *de = readdir(d)) {
              const std::string n = de->d_name;
              if (n.size() < 12) continue;
              const std::string pd = "/sys/bus/pci/devices/" + n;
              if (slurp(pd + "/vendor") == "0x1002" && slurp(pd + "/class") == my_class && slurp(pd + "/numa_node") == my_node) peers.push_back(n);
            }
            closedir(d);
          }
          std::sort(peers.begin(), peers.end());
          const auto it = std::find(peers.begin(), peers.end(), std::string(addr));
          if (it != peers.end()) { gpu_index = (int)(it - peers.begin()); gpus_on_node = (int)peers.size(); }
        }
      }
      win = pick_cpu_window(local, gpu_index, gpus_on_node, want);
      if ((int)win.size() < std::min(want, 2)) win.clear();
    }
  } catch (...) {
    win.clear();                      // an unreadable / malformed sysfs entry: no pinning
  }
  cache.push_back({device, win});
  return win;
}

inline bool pin_this_thread(const std::vector<int> &cpus) {    // pin_thread_to_cpus, src/system.cpp:211-225
  if (cpus.empty()) return false;
  cpu_set_t set;
  CPU_ZERO(&set);
  for (int c : cpus) if (c >= 0 && c < CPU_SETSIZE) CPU_SET(c, &set);
  return pthread_setaffinity_np(pthread_self(), sizeof set, &set) == 0;
}

// At most `tokens` workers of the process FILL a staging batch (decode + copy-out) at any time; waiting for the GPU
// holds no token.  The reference never runs more stream threads than CPUs (calculate_parallel_streams,
// src/system.cpp:186-197: min(configured, detect_cpu_limit())); with GPUs in place of CPU sets the number of open
// streams is no longer tied to CPUs (BASELINE config 4: 64 streams), but runnable threads still are: 64 runnable
// workers under a 16-CPU cgroup quota were each scheduled ~16 % of the time while "copying" (the quota is handed
// to CPUs in slices; round 4, profiles/r04_host_feed_*.json), i.e. a copy-out that streams 37 GB/s per thread
// when it runs delivered 2.7 GB/s per worker.  The gate keeps the runnable set at the CPU budget.
// MTGPU_CPU_TOKENS: 0 = no gate, N = that many tokens, unset = 3/4 of cpu_budget() (rounded up): the budget also has to
// carry the HIP runtime's own threads and the wake-ups of the waiting workers — measured on a 16-CPU quota with 64
// workers (profiles/r04_host_feed_ab_gate.json): no gate 84-99 k frames/s, 24 tokens 102 k, 16 tokens 115 k,
// 12 tokens 135 k, 8 tokens 134 k.
class CpuGate {
  std::mutex mu_;
  std::condition_variable cv_;
  int free_ = 0, tokens_ = 0;
  std::atomic<uint64_t> waits_{0}, wait_us_{0};
 public:
  explicit CpuGate(int tokens) : free_(tokens), tokens_(tokens) {}
  static CpuGate &instance() {
    static CpuGate g([] {
      const char *e = std::getenv("MTGPU_CPU_TOKENS");
      return e ? std::max(0, std::atoi(e)) : std::max(1, (3 * cpu_budget() + 3) / 4);
    }());
    return g;
  }
  int tokens() const { return tokens_; }
  uint64_t waits() const { return waits_.load(); }
  uint64_t wait_us() const { return wait_us_.load(); }
  void acquire() {
    if (tokens_ <= 0) return;
    std::unique_lock<std::mutex> l(mu_);
    if (free_ == 0) {
      const auto t0 = std::chrono::steady_clock::now();
      cv_.wait(l, [&] { return free_ > 0; });
      ++waits_;
      wait_us_ += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    }
    --free_;
  }
  void release() {
    if (tokens_ <= 0) return;
    { std::lock_guard<std::mutex> l(mu_); ++free_; }
    cv_.notify_one();
  }
};

// ---------------------------------------------------------------- work / result plumbing
struct ScanTask { double start, end; int id; };

class TaskQueue {   // mutex + condvar FIFO of chunks, as task_queue.cpp:20-39
  std::queue<ScanTask> q_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool done_ = false;
 public:
  void push(ScanTask t) { { std::lock_guard<std::mutex> l(mu_); q_.push(t); } cv_.notify_one(); }
  bool pop(ScanTask &t) {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return !q_.empty() || done_; });
    if (q_.empty()) return false;
    t = q_.front(); q_.pop();
    return true;
  }
  void finish() { { std::lock_guard<std::mutex> l(mu_); done_ = true; } cv_.notify_all(); }
};

class ResultCollector {   // unordered pooling of per-chunk timestamps, task_queue.cpp:43-57
  std::vector<double> ts_;
  std::mutex mu_;
 public:
  void add(std::vector<double> &&r) { std::lock_guard<std::mutex> l(mu_); ts_.insert(ts_.end(), r.begin(), r.end()); }
  std::vector<double> extract() { std::lock_guard<std::mutex> l(mu_); return std::move(ts_); }
};

// ---------------------------------------------------------------- decoded-frame interface
struct Frame {
  int64_t pts = 0;             // AVFrame::pts in time_base units
  const void *mv = nullptr;    // AV_FRAME_DATA_MOTION_VECTORS bytes, nullptr if absent
  size_t mv_bytes = 0;
  bool has_side_data = false;
};

class FrameSource {   // what the scan needs from a decoder (motion_scanner.cpp:204-215, 321-354)
 public:
  virtual ~FrameSource() = default;
  virtual int width() const = 0;
  virtual int height() const = 0;
  virtual double duration() const = 0;
  virtual double fps() const = 0;
  virtual double time_base() const = 0;
  // av_seek_frame(..., AVSEEK_FLAG_BACKWARD) + flush: continue from the last keyframe at or
  // before `seconds`
  virtual void seek(double seconds) = 0;
  // next decoded frame in decode order; the Frame's bytes stay valid until the next call only
  virtual bool next(Frame &f) = 0;
};

// .mtmv — the repo's binary MV container (written by mvfile.py); see that file for the layout.
struct MtmvHeader {
  char magic[8];
  uint32_t width, height, tb_num, tb_den;
  double fps, duration;
  uint64_t n_frames, n_records;
};
struct MtmvFrameRec {
  int64_t pts;
  uint64_t rec_off;
  uint32_t n_rec;
  uint8_t has_sd, key, pad[2];
};
static_assert(sizeof(MtmvHeader) == 56 && sizeof(MtmvFrameRec) == 24, "mtmv layout");

class MtmvFile {   // one mmap shared by all sources of a video (the reference mmaps the input once, memory_io.cpp:73-132)
  int fd_ = -1;
  const uint8_t *base_ = nullptr;
  size_t size_ = 0;
 public:
  const MtmvHeader *hdr = nullptr;
  const MtmvFrameRec *frames = nullptr;
  const uint8_t *records = nullptr;
  explicit MtmvFile(const std::string &path) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) throw std::runtime_error("cannot open " + path);
    struct stat st{};
    if (fstat(fd_, &st) != 0 || st.st_size < (off_t)sizeof(MtmvHeader)) throw std::runtime_error("bad mtmv file");
    size_ = (size_t)st.st_size;
    // MAP_POPULATE + sequential / huge-page advice, as the reference maps its input (memory_io.cpp:103-115)
    base_ = static_cast<const uint8_t *>(mmap(nullptr, size_, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd_, 0));
    if (base_ == MAP_FAILED) throw std::runtime_error("mmap failed");
    (void)madvise(const_cast<uint8_t *>(base_), size_, MADV_SEQUENTIAL);
#ifdef MADV_HUGEPAGE
    (void)madvise(const_cast<uint8_t *>(base_), size_, MADV_HUGEPAGE);
#endif
    hdr = reinterpret_cast<const MtmvHeader *>(base_);
    if (std::memcmp(hdr->magic, "MTMV1\0\0\0", 8) != 0) throw std::runtime_error("not an mtmv file");
    frames = reinterpret_cast<const MtmvFrameRec *>(base_ + sizeof(MtmvHeader));
    if (hdr->n_frames > (size_ - sizeof(MtmvHeader)) / sizeof(MtmvFrameRec) ||
        hdr->n_records > (size_ - sizeof(MtmvHeader) - sizeof(MtmvFrameRec) * hdr->n_frames) / 40ull)
      throw std::runtime_error("truncated mtmv file: " + path);
    for (uint64_t i = 0; i < hdr->n_frames; ++i)      // a damaged frame table must not send a worker outside the mapping
      if (frames[i].rec_off > hdr->n_records || frames[i].n_rec > hdr->n_records - frames[i].rec_off)
        throw std::runtime_error("corrupt mtmv frame table: " + path);
    records = base_ + sizeof(MtmvHeader) + sizeof(MtmvFrameRec) * hdr->n_frames;
  }
  ~MtmvFile() { if (base_ && base_ != MAP_FAILED) munmap(const_cast<uint8_t *>(base_), size_); if (fd_ >= 0) close(fd_); }
  MtmvFile(const MtmvFile &) = delete;
  MtmvFile &operator=(const MtmvFile &) = delete;
};

class MtmvSource : public FrameSource {
  const MtmvFile &f_;
  uint64_t pos_ = 0;
 public:
  explicit MtmvSource(const MtmvFile &f) : f_(f) {}
  int width() const override { return (int)f_.hdr->width; }
  int height() const override { return (int)f_.hdr->height; }
  double duration() const override { return f_.hdr->duration; }
  double fps() const override { return f_.hdr->fps; }
  double time_base() const override { return (double)f_.hdr->tb_num / (double)f_.hdr->tb_den; }   // av_q2d
  void seek(double seconds) override {
    const int64_t target = static_cast<int64_t>(seconds / time_base());   // motion_scanner.cpp:322
    uint64_t key = 0;
    for (uint64_t i = 0; i < f_.hdr->n_frames && f_.frames[i].pts <= target; ++i)
      if (f_.frames[i].key) key = i;
    pos_ = key;
  }
  bool next(Frame &fr) override {
    if (pos_ >= f_.hdr->n_frames) return false;
    const MtmvFrameRec &r = f_.frames[pos_++];
    fr.pts = r.pts;
    fr.has_side_data = r.has_sd != 0;
    fr.mv = r.has_sd ? f_.records + 40ull * r.rec_off : nullptr;
    fr.mv_bytes = r.has_sd ? 40ull * r.n_rec : 0;
    return true;
  }
};

// ---------------------------------------------------------------- scanner
// The GPU side of one worker: a scan context (cfg + launch plan) and its pinned pipe.  Kept
// separate from the decoder-facing scanner so that a batch worker can reuse it for the next
// video of the same size instead of re-pinning staging memory per file.
// What the GPU side of the host layer holds: one context + one pinned pipe per worker thread.
struct Resources {
  uint64_t contexts = 0, pipes = 0, hip_streams = 0, hip_events = 0, mem_pools = 0;
  uint64_t pinned_bytes = 0, device_bytes = 0, pool_reserved_high = 0, submits = 0;
  uint64_t ctx_create_us = 0, pipe_create_us = 0;        // set-up time spent in mtgpu_create / mtgpu_pipe_create_layout
  uint64_t pipe_rebuilds = 0;                            // pipes thrown away and re-created after a device-side failure
  uint64_t pin_us = 0, pinned_batches = 0;               // page-locking time of the pipes (creation + lazy first uses) / batches pinned
  void add(const Resources &o) {
    ctx_create_us += o.ctx_create_us; pipe_create_us += o.pipe_create_us; pipe_rebuilds += o.pipe_rebuilds;
    pin_us += o.pin_us; pinned_batches += o.pinned_batches;
    contexts += o.contexts; pipes += o.pipes; hip_streams += o.hip_streams; hip_events += o.hip_events;
    mem_pools += o.mem_pools; pinned_bytes += o.pinned_bytes; device_bytes += o.device_bytes;
    pool_reserved_high += o.pool_reserved_high; submits += o.submits;
  }
};

// One scan context per (device, parameter block) for the whole process, shared by every worker thread that
// needs it.  A context holds no per-frame state (the vote grid lives in LDS, launch scratch is stream-ordered,
// include/mtgpu.h: "may be called from many host threads"), so the reference's one-MotionScanner-per-worker
// model (src/pipeline.cpp:186-197) needs only a PIPE per worker: 64 streams x T workers then cost 1 context,
// 1 scratch pool and 1 context stream per device instead of 64 x T of each (profiles/r03_host_batch64.json:
// mtgpu_create took 0.3 s per worker when 64 threads called it at once).  The last user destroys the context.
class SharedContext {
  mtgpu_ctx *ctx_ = nullptr;
 public:
  int device = -1;
  mt_scan_params params{};
  uint64_t create_us = 0;
  explicit SharedContext(mtgpu_ctx *c) : ctx_(c) {}
  ~SharedContext() { if (ctx_) mtgpu_destroy(ctx_); }
  SharedContext(const SharedContext &) = delete;
  SharedContext &operator=(const SharedContext &) = delete;
  mtgpu_ctx *get() const { return ctx_; }

  static std::shared_ptr<SharedContext> acquire(const mt_scan_params &p, int device, std::string &err) {
    static std::mutex mu;
    static std::vector<std::weak_ptr<SharedContext>> live;
    std::lock_guard<std::mutex> l(mu);          // creation is serialised: the second caller finds the first one's context
    for (auto it = live.begin(); it != live.end();) {
      std::shared_ptr<SharedContext> sp = it->lock();
      if (!sp) { it = live.erase(it); continue; }
      if (sp->device == device && std::memcmp(&sp->params, &p, sizeof p) == 0) return sp;
      ++it;
    }
    const auto t0 = std::chrono::steady_clock::now();
    mtgpu_ctx *c = nullptr;
    if (mtgpu_create(&p, device, &c) != MT_OK) { err = mtgpu_last_error(); return nullptr; }
    auto sp = std::make_shared<SharedContext>(c);
    sp->device = device;
    sp->params = p;                              // (mtgpu_params_from_config zero-fills the padding: memcmp is exact)
    sp->create_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
    live.push_back(sp);
    return sp;
  }
};

class GpuBackend {
  std::shared_ptr<SharedContext> shared_;
  mtgpu_pipe *pipe_ = nullptr;
  int width_ = -1, height_ = -1, device_ = -1;
  uint64_t pipe_us_ = 0, rebuilds_ = 0;
  bool dirty_ = false;
 public:
  GpuBackend() = default;
  ~GpuBackend() { reset(); }
  GpuBackend(const GpuBackend &) = delete;
  GpuBackend &operator=(const GpuBackend &) = delete;
  void reset() {
    if (pipe_) mtgpu_pipe_destroy(pipe_);        // the pipe first: it launches on the context (destroy drains it)
    pipe_ = nullptr;
    shared_.reset();
    width_ = height_ = device_ = -1;
    dirty_ = false;
  }
  // A scanner that leaves this backend's pipe in an unknown state — a submit or collect failed, a batch may still
  // be in flight or retired (state 4) — marks it: the next video must not inherit that pipe (it would fail every
  // acquire, or collect the previous video's batch and pool ITS timestamps).  ensure() then rebuilds it.
  void mark_dirty() { dirty_ = true; }
  bool dirty() const { return dirty_; }
  mtgpu_ctx *ctx() { return shared_ ? shared_->get() : nullptr; }
  mtgpu_pipe *pipe() { return pipe_; }
  const void *context_identity() const { return shared_.get(); }
  // What THIS worker holds: its pipe.  The shared context is reported by context_resources(), once per context.
  Resources resources() {
    Resources r;
    r.pipe_create_us = pipe_us_;
    r.pipe_rebuilds = rebuilds_;
    mtgpu_pipe_stats ps;
    if (pipe_ && mtgpu_pipe_get_stats(pipe_, &ps) == MT_OK) {
      r.pipes = 1; r.hip_streams += ps.hip_streams; r.hip_events += ps.n_buffers;
      r.pinned_bytes += ps.pinned_bytes; r.device_bytes += ps.device_bytes + ps.list_bytes; r.submits = ps.submits;
      r.pin_us = ps.pin_us; r.pinned_batches = ps.pinned_batches;
    }
    return r;
  }
  Resources context_resources() {
    Resources r;
    mtgpu_ctx_stats cs;
    if (shared_ && mtgpu_get_stats(shared_->get(), &cs) == MT_OK) {
      r.contexts = 1; r.hip_streams += cs.hip_streams; r.mem_pools = cs.private_pool;
      r.device_bytes += cs.staging_device_bytes + cs.pool_reserved_high; r.pool_reserved_high = cs.pool_reserved_high;
      r.ctx_create_us = shared_->create_us;
    }
    return r;
  }
  // between two videos: hand the scratch pool's cached blocks back (the spill queue of a banded plan is
  // 4 bytes per record of the largest batch ever scanned) — but only when no other worker shares the context:
  // while others scan, their next launches would map the same scratch again at once, and the pool is per
  // device now, not per worker, so what it retains no longer multiplies with N x S
  void trim() { if (shared_ && shared_.use_count() == 1) (void)mtgpu_trim(shared_->get()); }
  // cfg/grid derivation of MotionScanner::initialize (motion_scanner.cpp:184-199) + device setup;
  // a backend already set up for this frame size and device is reused as it is.
  bool ensure(int width, int height, int device, uint64_t batch_records, uint32_t batch_frames, int n_buffers,
              std::string &err) {
    if (shared_ && pipe_ && !dirty_ && width == width_ && height == height_ && device == device_) return true;
    if (dirty_) ++rebuilds_;
    reset();
    mt_scan_params p;
    int rc = mtgpu_params_from_config(&p, width, height, Config::mv_threshold_sq(), Config::block_size(),
                                      Config::block_shift(), Config::vectors_needed(), Config::clusters_needed(),
                                      Config::vertical_mask());
    if (rc != MT_OK) { err = mtgpu_last_error(); return false; }
    shared_ = SharedContext::acquire(p, device, err);
    if (!shared_) return false;
    const auto t1 = std::chrono::steady_clock::now();
    rc = mtgpu_pipe_create_layout(shared_->get(), batch_records, batch_frames, n_buffers, Config::staging_layout(), &pipe_);
    pipe_us_ += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count();
    if (rc != MT_OK) { err = mtgpu_last_error(); reset(); return false; }
    width_ = width; height_ = height; device_ = device;
    return true;
  }
};

class GpuMotionScanner {   // public shape of MotionScanner (motion_scanner.hpp:113-152)
  FrameSource &src_;
  int device_;
  GpuBackend own_;
  GpuBackend *be_;           // own_ or a backend shared across the videos of one batch worker
  mtgpu_pipe *pipe_ = nullptr;
  mtgpu_batch *cur_ = nullptr;
  int inflight_ = 0;
  std::string err_;
  long copy_us_ = 0, submit_us_ = 0, wait_us_ = 0;   // inside analyze_us: copy-out / submit calls / waiting for the GPU
  uint64_t frames_fed_ = 0;                           // frames that reached check_frame (after the filter of :357-371)
  static long since(std::chrono::high_resolution_clock::time_point t0) {
    return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t0).count();
  }

  bool device_error_ = false;                         // a submit / collect / release failed: the pipe's state is unknown
  bool ok(int rc) {
    if (rc == MT_OK) return true;
    err_ = mtgpu_last_error();
    if (rc != MT_ERR_CAPACITY && rc != MT_ERR_BUSY && rc != MT_ERR_INVALID) device_error_ = true;
    return false;
  }

  bool collect_one(std::vector<double> &ts) {
    mtgpu_batch *b = nullptr;
    const uint8_t *flags = nullptr;
    const double *pts = nullptr;
    uint32_t n = 0;
    const auto w0 = std::chrono::high_resolution_clock::now();
    const bool got = ok(mtgpu_pipe_collect(pipe_, &b, &flags, &pts, nullptr, &n));
    wait_us_ += since(w0);
    if (!got) {
      // a failed collect still hands the batch out (include/mtgpu.h): give it back, keep the first error
      if (b) { const std::string first = err_; --inflight_; (void)mtgpu_pipe_release(pipe_, b); err_ = first; }
      return false;
    }
    for (uint32_t i = 0; i < n; ++i)
      if (flags[i]) ts.push_back(pts[i]);                    // :382-383
    --inflight_;
    return ok(mtgpu_pipe_release(pipe_, b));
  }
  // the CPU token (CpuGate) is held exactly while a batch is being filled: from its acquire to its submit / release
  bool token_ = false;
  void take_token() { if (!token_) { CpuGate::instance().acquire(); token_ = true; } }
  void drop_token() { if (token_) { CpuGate::instance().release(); token_ = false; } }
  void release_current() { if (cur_) { mtgpu_pipe_release(pipe_, cur_); cur_ = nullptr; } drop_token(); }
  bool submit() {
    if (!cur_) return true;
    const auto s0 = std::chrono::high_resolution_clock::now();
    const bool sent = ok(mtgpu_pipe_submit(pipe_, cur_));
    submit_us_ += since(s0);
    if (!sent) return false;
    cur_ = nullptr;
    drop_token();
    ++inflight_;
    return true;
  }
  bool feed(const Frame &f, double pts, std::vector<double> &ts) {
    for (;;) {
      if (!cur_) {
        int rc = mtgpu_pipe_acquire(pipe_, &cur_);
        if (rc == MT_ERR_BUSY) { if (!collect_one(ts)) return false; continue; }   // back-pressure (no token held)
        if (!ok(rc)) return false;
        take_token();
      }
      const auto c0 = std::chrono::high_resolution_clock::now();
      int rc = mtgpu_batch_add_frame(cur_, f.mv, f.mv_bytes, f.has_side_data ? 1 : 0, pts, 0);
      copy_us_ += since(c0);
      if (rc == MT_ERR_CAPACITY) { if (!submit()) return false; continue; }
      if (rc == MT_OK) ++frames_fed_;
      return ok(rc);
    }
  }

 public:
  GpuMotionScanner(FrameSource &src, int device, GpuBackend *shared = nullptr)
      : src_(src), device_(device), be_(shared ? shared : &own_) {}
  ~GpuMotionScanner() {      // leave a shared pipe idle: nothing in flight, nothing half-filled
    if (!pipe_) { drop_token(); return; }
    std::vector<double> sink;
    while (inflight_ > 0 && collect_one(sink)) {}
    release_current();
    // anything that went wrong on the device side, or a batch that could not be collected: the pipe is not
    // handed to the next video as it is (GpuBackend::ensure rebuilds a dirty backend; destroying a pipe drains it)
    if (device_error_ || inflight_ > 0) be_->mark_dirty();
  }
  GpuMotionScanner(const GpuMotionScanner &) = delete;
  GpuMotionScanner &operator=(const GpuMotionScanner &) = delete;
  // Empty while every call so far succeeded.  scan_range keeps the reference's signature (it
  // returns the timestamps), so a failed range is reported here: callers MUST check failed()
  // after each scan_range — a range that failed returns only part of its timestamps.
  const std::string &error() const { return err_; }
  bool failed() const { return !err_.empty(); }
  long copy_us() const { return copy_us_; }
  long submit_us() const { return submit_us_; }
  long wait_us() const { return wait_us_; }
  uint64_t frames_fed() const { return frames_fed_; }
  // The scan context of this worker's device.  It is SHARED by every worker of the process that scans the same
  // frame size on that device (SharedContext): treat it as read-only — query it (mtgpu_get_plan / _get_params /
  // _get_stats) and pass it to the merge, but do not call per-context mutators (mtgpu_set_slices) or the
  // host-pointer scan entry points on it from a worker: they take the context's lock and its single stream and
  // would serialise or re-plan all S x T workers.  (run_scan_pipeline's own merge call is one short call per video.)
  mtgpu_ctx *context() { return be_->ctx(); }

  // batch_records == 0: sized from the source and the staging layout — MTGPU_BATCH_MB MiB of
  // pinned staging per batch, and never less than two frames of one record per 4x4 block (the
  // finest partition H.264/HEVC export).  Batches of a few MiB amortise the per-batch copy /
  // launch / event calls; a single frame beyond a whole batch still works (the pipe grows an
  // empty batch, include/mtgpu.h).
  bool initialize(uint64_t batch_records = 0, uint32_t batch_frames = 256, int n_buffers = 3) {
    if (batch_records == 0) {
      const uint64_t fine = (uint64_t)((src_.width() + 3) / 4) * (uint64_t)((src_.height() + 3) / 4);
      const uint64_t rec_bytes = (Config::staging_layout() & MT_LAYOUT_AOS40) ? MT_MV_BYTES : MT_COMPACT_BYTES;
      batch_records = std::max<uint64_t>(2 * fine, ((uint64_t)Config::batch_mib() << 20) / rec_bytes);
    }
    if (!be_->ensure(src_.width(), src_.height(), device_, batch_records, batch_frames, n_buffers, err_)) return false;
    pipe_ = be_->pipe();
    return true;
  }
  double get_duration() { return src_.duration(); }
  double get_fps() { return src_.fps(); }

  // MotionScanner::scan_range (motion_scanner.cpp:297-391): same filter, same timestamps;
  // check_frame runs on the GPU, overlapped with the "decode" of the following frames.
  std::vector<double> scan_range(double start, double end, long &seek_us, long &decode_us, long &analyze_us) {
    using clk = std::chrono::high_resolution_clock;
    auto us = [](clk::time_point a, clk::time_point b) {
      return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
    std::vector<double> ts;
    // however this call is left — also by an exception out of the FrameSource — the CPU token goes back: a worker
    // that died holding it would starve the others (with one token: for good)
    struct TokenGuard { GpuMotionScanner &s; ~TokenGuard() { s.drop_token(); } } token_guard{*this};
    const double time_base = src_.time_base();
    const double video_fps = src_.fps();
    const double target = Config::target_fps();
    const int frame_skip = (target > 0 && target < video_fps) ? static_cast<int>(video_fps / target) : 1;   // :311-313
    int frame_count = 0;
    auto t0 = clk::now();
    if (start > 0) src_.seek(start);                                     // :321-325
    else src_.seek(0.0);
    auto t1 = clk::now();
    seek_us += us(t0, t1);
    Frame f;
    for (;;) {
      auto d0 = clk::now();
      const bool got = src_.next(f);
      auto d1 = clk::now();
      decode_us += us(d0, d1);
      if (!got) break;
      if (++frame_count % frame_skip != 0) continue;                     // :357
      const double pts = f.pts * time_base;                              // :361
      if (pts < start) continue;                                         // :364-365
      if (pts >= end) break;                                             // :368-371
      auto a0 = clk::now();
      const bool fed = feed(f, pts, ts);                                 // copy-out + async scan (:376)
      analyze_us += us(a0, clk::now());
      if (!fed) break;                                                   // err_ is set: failed()
    }
    auto a0 = clk::now();
    if (!failed() && cur_ && mtgpu_batch_frames(cur_) > 0) submit();
    if (failed()) {
      // leave the pipe idle: finish what is in flight (results discarded), drop the partial batch
      std::vector<double> sink;
      const std::string first = err_;
      while (inflight_ > 0 && collect_one(sink)) {}
      release_current();
      err_ = first;
    } else {
      while (inflight_ > 0) if (!collect_one(ts)) break;
    }
    analyze_us += us(a0, clk::now());
    return ts;
  }
};

// ---------------------------------------------------------------- per-video pipeline (scan + merge)
struct PipelineResult {
  std::vector<mt_segment> segments;   // what FFmpegJob::segments would carry (pipeline.cpp:366, 396)
  mt_merge_result merge{};
  size_t motion_frames = 0;           // pooled timestamps before sort/unique (pipeline.cpp:294-295)
  uint64_t frames_scanned = 0;        // frames that went through check_frame on the GPU, all workers
  std::vector<double> timestamps;     // those timestamps, in pooling order (ResultCollector::extract, :268)
  int chunks = 0, threads = 0;
  long seek_us = 0, decode_us = 0, analyze_us = 0;   // summed over workers, as pipeline.cpp:229-233
  long init_us = 0, scan_wall_us = 0;                // worker init (summed) / wall time of the scan phase
  long scan_work_us = 0;                             // wall time from "every worker initialised" to the last result
  long long work_begin_abs_us = 0, work_end_abs_us = 0;   // those two instants on the steady clock (process_batch
                                                          // needs them to find the window in which ANY video was scanned)
  long copy_us = 0, submit_us = 0, wait_us = 0;      // parts of analyze_us (summed over workers): copy-out into pinned
                                                     // staging, submit calls, waiting for the GPU
  long worker_cpu_us = 0;                            // CPU time the worker threads actually got (CLOCK_THREAD_CPUTIME_ID, summed):
                                                     // far below their wall time = they were runnable but not running
  std::string error;
};

// The scan + merge part of ProcessingPipeline::run: chunk the timeline, N workers (each with its
// own source + GpuMotionScanner, worker i on device i % n_devices), pool, merge on the GPU.
// `make_source` is called once per worker (+ once for the probe), like the per-thread
// MotionScanner(file_buffer) of pipeline.cpp:197.
// `pool` (optional, >= num_threads entries) lends each worker a GpuBackend that outlives this call.
// Which GPU worker `worker` of a pipeline uses: pipelines (streams) start at device_base =
// stream * threads_per_stream (process_batch), so concurrent streams and their workers spread
// round-robin over the node's GPUs — the reference's stream -> CPU-set assignment
// (batch_processor.cpp:102-110) with GPUs in place of CPU sets.
inline int worker_device(int device_base, int worker, int n_devices) {
  return n_devices > 0 ? (device_base + worker) % n_devices : 0;
}

template <class MakeSource>
int run_scan_pipeline(MakeSource make_source, int num_threads, PipelineResult &out, int device_base = 0,
                      std::vector<std::unique_ptr<GpuBackend>> *pool = nullptr) {
  // every environment value is parsed here, once, in the calling thread (never first inside a worker)
  try { Config::load_all(); } catch (const std::exception &e) { out.error = std::string("configuration: ") + e.what(); return 1; }
  std::unique_ptr<FrameSource> probe;
  try { probe = make_source(); } catch (const std::exception &e) { out.error = e.what(); return 1; }   // pipeline.cpp:110-120
  const double duration = probe->duration();
  const double chunk = Config::chunk_duration_sec();
  int n_dev = mtgpu_device_count();
  if (n_dev < 1) { out.error = "no GPU"; return 1; }
  const int num_chunks = static_cast<int>(std::ceil(duration / chunk));  // :141-142
  num_threads = std::max(1, std::min(num_threads, std::max(1, num_chunks)));   // :143
  TaskQueue tasks;
  ResultCollector results;
  int chunk_id = 0;
  for (double t = 0; t < duration; t += chunk)                           // :163-167
    tasks.push({t, std::min(t + chunk, duration), chunk_id++});
  std::atomic<long> seek_us{0}, decode_us{0}, analyze_us{0}, init_us{0}, copy_us{0}, submit_us{0}, wait_us{0};
  std::atomic<uint64_t> frames_scanned{0};
  std::atomic<long> worker_cpu_us{0};
  const auto wall0 = std::chrono::high_resolution_clock::now();
  std::mutex err_mu;
  std::vector<std::thread> workers;
  mtgpu_ctx *merge_ctx = nullptr;
  std::mutex ctx_mu;
  std::vector<std::unique_ptr<GpuMotionScanner>> scanners(num_threads);
  std::vector<std::unique_ptr<FrameSource>> sources(num_threads);
  // workers start pulling chunks together, once all of them are initialised (or have failed):
  // separates GPU context / pinned-memory set-up from the steady-state scan in the timings
  std::mutex start_mu;
  std::condition_variable start_cv;
  int ready = 0;
  std::chrono::high_resolution_clock::time_point work0 = wall0;
  auto arrive = [&] {
    std::unique_lock<std::mutex> l(start_mu);
    if (++ready == num_threads) { work0 = std::chrono::high_resolution_clock::now(); start_cv.notify_all(); }
    else start_cv.wait(l, [&] { return ready == num_threads; });
  };
  for (int i = 0; i < num_threads; ++i) {
    workers.emplace_back([&, i] {                                        // :186-235
      // Nothing may escape a std::thread body (std::terminate): a throwing make_source() or a bad_alloc
      // ends this worker with the video failed, and the start barrier still gets its arrival.
      bool arrived = false;
      auto fail_with = [&](const std::string &what) {
        { std::lock_guard<std::mutex> l(err_mu); if (out.error.empty()) out.error = what; }
        if (!arrived) { arrived = true; arrive(); }
      };
      try {
        const auto i0 = std::chrono::high_resolution_clock::now();
        const int dev = worker_device(device_base, i, n_dev);
        (void)pin_this_thread(cpu_window_for_device(dev));               // :192-194 (optional pinning), next to the device
        sources[i] = make_source();
        GpuBackend *shared = (pool && (size_t)i < pool->size()) ? (*pool)[i].get() : nullptr;
        scanners[i] = std::make_unique<GpuMotionScanner>(*sources[i], dev, shared);
        if (!scanners[i]->initialize()) {                                // :198-199 (here: reported)
          fail_with(scanners[i]->error());
          return;
        }
        init_us += (long)std::chrono::duration_cast<std::chrono::microseconds>(
                       std::chrono::high_resolution_clock::now() - i0).count();   // :195-206
        arrived = true;
        arrive();
        long s = 0, d = 0, a = 0;
        ScanTask task;
        while (tasks.pop(task)) {                                        // :216-223
          auto r = scanners[i]->scan_range(task.start, task.end, s, d, a);
          if (scanners[i]->failed()) {                                   // a failed range must fail the video
            std::lock_guard<std::mutex> l(err_mu);
            if (out.error.empty()) out.error = scanners[i]->error();
            break;
          }
          if (!r.empty()) results.add(std::move(r));
        }
        seek_us += s; decode_us += d; analyze_us += a;
        copy_us += scanners[i]->copy_us(); submit_us += scanners[i]->submit_us(); wait_us += scanners[i]->wait_us();
        frames_scanned += scanners[i]->frames_fed();
        struct timespec tc{};
        if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &tc) == 0) worker_cpu_us += (long)tc.tv_sec * 1000000L + tc.tv_nsec / 1000;
      } catch (const std::exception &e) {
        fail_with(std::string("worker ") + std::to_string(i) + ": " + e.what());
      } catch (...) {
        fail_with(std::string("worker ") + std::to_string(i) + ": unknown exception");
      }
    });
  }
  tasks.finish();
  for (auto &w : workers) w.join();
  out.chunks = chunk_id;
  out.threads = num_threads;
  out.seek_us = seek_us; out.decode_us = decode_us; out.analyze_us = analyze_us; out.init_us = init_us;
  out.copy_us = copy_us; out.submit_us = submit_us; out.wait_us = wait_us;
  out.frames_scanned = frames_scanned;
  out.worker_cpu_us = worker_cpu_us;
  {
    const auto wall1 = std::chrono::high_resolution_clock::now();
    out.scan_wall_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(wall1 - wall0).count();
    out.scan_work_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(wall1 - work0).count();
    out.work_begin_abs_us = (long long)std::chrono::duration_cast<std::chrono::microseconds>(work0.time_since_epoch()).count();
    out.work_end_abs_us = (long long)std::chrono::duration_cast<std::chrono::microseconds>(wall1.time_since_epoch()).count();
  }
  if (!out.error.empty()) return 1;
  out.timestamps = results.extract();
  const std::vector<double> &timestamps = out.timestamps;
  out.motion_frames = timestamps.size();
  for (auto &s : scanners) if (s && s->context()) { merge_ctx = s->context(); break; }
  if (!merge_ctx) { out.error = "no scanner context"; return 1; }
  // sort + unique + merge + clamp + savings + cut decision on the device (pipeline.cpp:302-358, 387-388)
  mt_merge_params mp{Config::max_gap_sec(), Config::padding_sec(), duration, Config::min_savings_pct()};
  out.segments.assign(timestamps.size() + 1, mt_segment{0, 0});
  int rc = mtgpu_merge_segments(merge_ctx, timestamps.data(), timestamps.size(), &mp, 1, out.segments.data(),
                                out.segments.size(), &out.merge);
  if (rc != MT_OK) { out.error = mtgpu_last_error(); return 1; }
  out.segments.resize(out.merge.n_segments);
  return 0;
}

// ---------------------------------------------------------------- batch of videos
// What the reference pushes to its FFmpeg consumer (ffmpeg_queue.hpp:32-38), minus the CPU set:
// the untouched cut executor consumes `segments` as they are.
struct ScanJob {
  int stream_id = -1;
  std::string input_path, output_path;
  std::vector<mt_segment> segments;
  PipelineResult result;
};

// The text the untouched cut executor builds from a job's segments and feeds to `ffmpeg -f concat`
// (ffmpeg_executor.cpp:38-50, same lines in pipeline.cpp:464-470): per segment with end > start
//   file '<absolute input path>' / inpoint <start, 2 decimals> / outpoint <end, 2 decimals>.
// This is where the doubles of the merge are rounded for the first time ({:.2f} of fmt == "%.2f": correctly
// rounded decimal of the exact binary value); a consumer that wants to diff against the reference's cut list
// without running ffmpeg formats it here.  Convenience only: the hand-off to the executor is ScanJob::segments.
inline std::string concat_list(const std::vector<mt_segment> &segments, const std::string &abs_input_path) {
  // fmt's {:.2f} prints every digit of the exact binary value and ignores the C locale; std::to_chars(fixed, 2)
  // does the same (snprintf("%.2f") would follow LC_NUMERIC's decimal comma and needs a buffer sized for 1e308)
  auto fixed2 = [](double v) {
    char buf[400];                                          // DBL_MAX has 309 integer digits
    const std::to_chars_result r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed, 2);
    return r.ec == std::errc() ? std::string(buf, r.ptr) : std::string("nan");
  };
  std::string out;
  for (const mt_segment &s : segments) {
    if (s.end <= s.start) continue;                                      // :45-46
    out += "file '" + abs_input_path + "'\n";
    out += "inpoint " + fixed2(s.start) + "\n";
    out += "outpoint " + fixed2(s.end) + "\n";
  }
  return out;
}

class JobQueue {   // producer/consumer queue of finished scans, as ffmpeg_queue.cpp:10-34
  std::queue<ScanJob> q_;
  std::mutex mu_;
  std::condition_variable cv_;
  bool done_ = false;
 public:
  void push(ScanJob &&j) { { std::lock_guard<std::mutex> l(mu_); q_.push(std::move(j)); } cv_.notify_one(); }
  bool pop(ScanJob &j) {
    std::unique_lock<std::mutex> l(mu_);
    cv_.wait(l, [&] { return !q_.empty() || done_; });
    if (q_.empty()) return false;
    j = std::move(q_.front()); q_.pop();
    return true;
  }
  void finish() { { std::lock_guard<std::mutex> l(mu_); done_ = true; } cv_.notify_all(); }
  bool empty() { std::lock_guard<std::mutex> l(mu_); return q_.empty(); }           // ffmpeg_queue.hpp:80-83
  bool is_done() { std::lock_guard<std::mutex> l(mu_); return done_ && q_.empty(); } // ffmpeg_queue.hpp:75
};

// BatchProcessor's stream fan-out (batch_processor.cpp:81-157, 307-350) with GPUs in place of
// CPU sets: S stream threads pull files from one queue; the workers of stream s use devices
// (s * threads_per_stream + i) % n_devices, so concurrent streams spread over the node's GPUs.
// Every finished scan is pushed to `jobs` (no motion -> no job, as pipeline.cpp:308-319).
// open_source(path) must return a factory of per-worker FrameSources for that file.
struct BatchSummary {   // what a whole process_batch run did and what it held (reported, never decided on)
  int streams = 0, threads_per_stream = 0;
  size_t videos = 0, jobs = 0, failed = 0;
  uint64_t frames_scanned = 0;
  long wall_us = 0;                                        // first stream thread started -> last one finished (teardown included)
  long scan_wall_us = 0;                                   // ... -> last video finished (contexts / pipes still alive)
  long scan_window_us = 0;                                 // first worker of any video ready -> last result of any video:
                                                           // the steady-state window (pipes persist from video to video, so
                                                           // only each stream's first video pays set-up, before this window)
  long long window_begin_abs_us = 0, window_end_abs_us = 0;
  long init_us = 0, decode_us = 0, analyze_us = 0, copy_us = 0, submit_us = 0, wait_us = 0;   // summed over all workers
  long worker_cpu_us = 0;                                  // CPU time the worker threads got (thread clocks, summed)
  long gate_wait_us = 0; int gate_tokens = 0;              // CpuGate: time workers waited for a CPU token / tokens
  int cpu_window = 0, cpu_window_first = -1;               // CPUs the workers of device 0 are confined to (0 = not pinned) / the first of them
  long cpu_user_us = 0, cpu_sys_us = 0;                    // CPU time the whole process spent during the run (getrusage):
                                                           // (user + sys) / wall = CPUs kept busy, against the box's quota
  Resources held;                                          // summed over the S x T backends alive at the end
};

template <class OpenSource>
int process_batch(const std::vector<std::string> &files, const std::string &output_dir, int parallel_streams,
                  int threads_per_stream, OpenSource open_source, JobQueue &jobs, std::vector<std::string> *errors,
                  BatchSummary *summary = nullptr) {
  parallel_streams = std::max(1, std::min<int>(parallel_streams, (int)files.size()));
  threads_per_stream = std::max(1, threads_per_stream);
  std::mutex q_mu, e_mu, s_mu;
  size_t next = 0;
  std::atomic<int> failed{0};
  BatchSummary sum;
  std::vector<const void *> seen_ctx;
  sum.streams = parallel_streams;
  sum.threads_per_stream = threads_per_stream;
  sum.videos = files.size();
  const auto wall0 = std::chrono::high_resolution_clock::now();
  struct rusage ru0{};
  (void)getrusage(RUSAGE_SELF, &ru0);
  const uint64_t gate_wait0 = CpuGate::instance().wait_us();
  std::vector<std::thread> streams;
  for (int s = 0; s < parallel_streams; ++s) {
    streams.emplace_back([&, s] {
      {   // the stream thread itself (probe, chunking, merge call) sits with its first worker (batch_processor.cpp:311-322)
        const int nd = mtgpu_device_count();
        if (nd > 0) (void)pin_this_thread(cpu_window_for_device(worker_device(s * threads_per_stream, 0, nd)));
      }
      // this stream's workers keep their GPU contexts + pinned pipes from one video to the next
      std::vector<std::unique_ptr<GpuBackend>> pool;
      for (int i = 0; i < threads_per_stream; ++i) pool.emplace_back(new GpuBackend());
      struct AtExit {                                      // every way out of the loop reports what this stream held
        std::vector<std::unique_ptr<GpuBackend>> &pool; std::mutex &mu; BatchSummary &sum;
        std::vector<const void *> &seen;
        ~AtExit() {
          std::lock_guard<std::mutex> l(mu);
          for (auto &b : pool) {
            sum.held.add(b->resources());
            const void *id = b->context_identity();          // a shared context is counted once for the whole batch
            if (id && std::find(seen.begin(), seen.end(), id) == seen.end()) {
              seen.push_back(id);
              sum.held.add(b->context_resources());
            }
          }
        }
      } at_exit{pool, s_mu, sum, seen_ctx};
      for (;;) {
        size_t idx;
        {
          std::lock_guard<std::mutex> l(q_mu);                                                     // get_next_file
          if (next >= files.size()) {
            const long now = (long)std::chrono::duration_cast<std::chrono::microseconds>(
                                 std::chrono::high_resolution_clock::now() - wall0).count();
            std::lock_guard<std::mutex> l2(s_mu);
            sum.scan_wall_us = std::max(sum.scan_wall_us, now);
            return;
          }
          idx = next++;
        }
        const std::string &in = files[idx];
        ScanJob job;
        job.stream_id = s;
        job.input_path = in;
        const size_t slash = in.find_last_of('/');
        job.output_path = output_dir + "/" + (slash == std::string::npos ? in : in.substr(slash + 1));
        int rc = 1;
        try {
          auto factory = open_source(in);
          rc = run_scan_pipeline(factory, threads_per_stream, job.result, s * threads_per_stream, &pool);
        } catch (const std::exception &e) {
          job.result.error = e.what();
        }
        for (auto &b : pool) b->trim();                      // scratch of this video goes back to the device
        {
          std::lock_guard<std::mutex> l(s_mu);
          const PipelineResult &r = job.result;
          sum.frames_scanned += r.frames_scanned;
          if (rc == 0 && r.work_end_abs_us > 0) {
            if (sum.window_begin_abs_us == 0 || r.work_begin_abs_us < sum.window_begin_abs_us) sum.window_begin_abs_us = r.work_begin_abs_us;
            if (r.work_end_abs_us > sum.window_end_abs_us) sum.window_end_abs_us = r.work_end_abs_us;
          }
          sum.init_us += r.init_us; sum.decode_us += r.decode_us; sum.analyze_us += r.analyze_us;
          sum.copy_us += r.copy_us; sum.submit_us += r.submit_us; sum.wait_us += r.wait_us;
          sum.worker_cpu_us += r.worker_cpu_us;
        }
        if (rc != 0) {
          ++failed;
          if (errors) { std::lock_guard<std::mutex> l(e_mu); errors->push_back(in + ": " + job.result.error); }
          continue;
        }
        if (job.result.merge.do_cut < 0) continue;           // "No motion found": nothing to cut
        job.segments = job.result.segments;
        { std::lock_guard<std::mutex> l(s_mu); ++sum.jobs; }
        jobs.push(std::move(job));
      }
    });
  }
  for (auto &t : streams) t.join();
  jobs.finish();
  sum.failed = (size_t)failed.load();
  sum.scan_window_us = (long)(sum.window_end_abs_us - sum.window_begin_abs_us);
  sum.wall_us = (long)std::chrono::duration_cast<std::chrono::microseconds>(
                    std::chrono::high_resolution_clock::now() - wall0).count();
  {
    struct rusage ru1{};
    (void)getrusage(RUSAGE_SELF, &ru1);
    auto us = [](const timeval &a, const timeval &b) { return (long)(b.tv_sec - a.tv_sec) * 1000000L + (long)(b.tv_usec - a.tv_usec); };
    sum.cpu_user_us = us(ru0.ru_utime, ru1.ru_utime);
    sum.cpu_sys_us = us(ru0.ru_stime, ru1.ru_stime);
    sum.gate_wait_us = (long)(CpuGate::instance().wait_us() - gate_wait0);
    sum.gate_tokens = CpuGate::instance().tokens();
    if (mtgpu_device_count() > 0) {
      const std::vector<int> w = cpu_window_for_device(0);
      sum.cpu_window = (int)w.size();
      sum.cpu_window_first = w.empty() ? -1 : w.front();
    }
  }
  if (summary) *summary = sum;
  return failed.load();
}

}  // namespace mtgpu_host
