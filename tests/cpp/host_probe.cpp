// CPU-tier test helper: drives csrc/host/mtgpu_host.hpp (Config, TaskQueue, ResultCollector, JobQueue) with the
// same commands and the same output format as oracle/ref_host_probe.cpp drives the reference's classes, so that
// tests/test_reference_host.py can compare the two line by line (fixture: tests/golden/reference_host_vectors.json,
// recorded from the reference itself).  Needs no GPU: nothing here creates a scan context.
#include <cstddef>
#include <clocale>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <typeinfo>
#include <vector>

#include "mtgpu_host.hpp"

namespace h = mtgpu_host;

template <class F> static void show_d(const char *name, F f) {
  try {
    double v = f();
    unsigned long long bits;
    std::memcpy(&bits, &v, 8);
    std::printf("%s f64 %.17g 0x%016llx\n", name, v, bits);
  } catch (const std::invalid_argument &) { std::printf("%s error invalid_argument\n", name);
  } catch (const std::out_of_range &) { std::printf("%s error out_of_range\n", name); }
}
template <class F> static void show_f(const char *name, F f) {
  try {
    float v = f();
    unsigned bits;
    std::memcpy(&bits, &v, 4);
    std::printf("%s f32 %.9g 0x%08x\n", name, static_cast<double>(v), bits);
  } catch (const std::invalid_argument &) { std::printf("%s error invalid_argument\n", name);
  } catch (const std::out_of_range &) { std::printf("%s error out_of_range\n", name); }
}
template <class F> static void show_i(const char *name, F f) {
  try {
    long long v = static_cast<long long>(f());
    std::printf("%s int %lld\n", name, v);
  } catch (const std::invalid_argument &) { std::printf("%s error invalid_argument\n", name);
  } catch (const std::out_of_range &) { std::printf("%s error out_of_range\n", name); }
}

static int cmd_config() {
  show_d("mv_threshold_sq", [] { return h::Config::mv_threshold_sq(); });
  show_i("block_size", [] { return h::Config::block_size(); });
  show_i("block_shift", [] { return h::Config::block_shift(); });
  show_i("vectors_needed", [] { return h::Config::vectors_needed(); });
  show_i("clusters_needed", [] { return h::Config::clusters_needed(); });
  show_f("vertical_mask", [] { return h::Config::vertical_mask(); });
  show_d("max_gap_sec", [] { return h::Config::max_gap_sec(); });
  show_d("padding_sec", [] { return h::Config::padding_sec(); });
  show_d("chunk_duration_sec", [] { return h::Config::chunk_duration_sec(); });
  show_d("target_fps", [] { return h::Config::target_fps(); });
  show_d("min_savings_pct", [] { return h::Config::min_savings_pct(); });
  show_i("parallel_streams", [] { return h::Config::parallel_streams(); });
  show_i("threads_per_stream", [] { return h::Config::threads_per_stream(); });
  return 0;
}

// `memo`: the getters are function-local statics (config.hpp:56-59) — what a process sees when the environment
// changes between two calls.  Script lines:  get <getter> | set NAME VALUE | unset NAME
static int cmd_memo() {
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string op, a, b;
    if (!(in >> op) || op[0] == '#') continue;
    in >> a;
    std::getline(in, b);
    if (!b.empty() && b[0] == ' ') b.erase(0, 1);
    if (op == "set") { setenv(a.c_str(), b.c_str(), 1); std::printf("set ok\n"); }
    else if (op == "unset") { unsetenv(a.c_str()); std::printf("unset ok\n"); }
    else if (op == "get") {
      if (a == "mv_threshold_sq") show_d("mv_threshold_sq", [] { return h::Config::mv_threshold_sq(); });
      else if (a == "block_size") show_i("block_size", [] { return h::Config::block_size(); });
      else if (a == "block_shift") show_i("block_shift", [] { return h::Config::block_shift(); });
      else if (a == "vectors_needed") show_i("vectors_needed", [] { return h::Config::vectors_needed(); });
      else if (a == "clusters_needed") show_i("clusters_needed", [] { return h::Config::clusters_needed(); });
      else if (a == "vertical_mask") show_f("vertical_mask", [] { return h::Config::vertical_mask(); });
      else if (a == "max_gap_sec") show_d("max_gap_sec", [] { return h::Config::max_gap_sec(); });
      else if (a == "padding_sec") show_d("padding_sec", [] { return h::Config::padding_sec(); });
      else if (a == "chunk_duration_sec") show_d("chunk_duration_sec", [] { return h::Config::chunk_duration_sec(); });
      else if (a == "target_fps") show_d("target_fps", [] { return h::Config::target_fps(); });
      else if (a == "min_savings_pct") show_d("min_savings_pct", [] { return h::Config::min_savings_pct(); });
      else if (a == "parallel_streams") show_i("parallel_streams", [] { return h::Config::parallel_streams(); });
      else if (a == "threads_per_stream") show_i("threads_per_stream", [] { return h::Config::threads_per_stream(); });
      else { std::printf("unknown getter %s\n", a.c_str()); return 2; }
    } else { std::printf("unknown %s\n", op.c_str()); return 2; }
  }
  return 0;
}

// mt_segment is the ABI twin of the reference's TimeSegment (two doubles, 16 bytes, 16-aligned); the host
// layer's ScanTask carries the same three fields but is a plain struct (no cache-line padding is needed for
// correctness), so only its field order is reported.
static int cmd_layout() {
  std::printf("TimeSegment size %zu align %zu start %zu end %zu\n", sizeof(mt_segment), alignof(mt_segment),
              offsetof(mt_segment, start), offsetof(mt_segment, end));
  std::printf("ScanTask start %zu end %zu id %zu\n", offsetof(h::ScanTask, start), offsetof(h::ScanTask, end),
              offsetof(h::ScanTask, id));
  return 0;
}

static int cmd_queue() {
  h::TaskQueue tq;
  h::ResultCollector rc;
  h::JobQueue jq;
  size_t t_size = 0, j_size = 0;
  bool t_done = false, j_done = false;
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string op;
    if (!(in >> op) || op[0] == '#') continue;
    if (op == "tpush") {
      h::ScanTask t{};
      in >> t.start >> t.end >> t.id;
      tq.push(t);
      ++t_size;
      std::printf("tpush ok\n");
    } else if (op == "tpop") {
      if (t_size == 0 && !t_done) { std::printf("tpop would_block\n"); continue; }
      h::ScanTask t{};
      if (tq.pop(t)) { --t_size; std::printf("tpop 1 %.17g %.17g %d\n", t.start, t.end, t.id); }
      else std::printf("tpop 0\n");
    } else if (op == "tfinish") {
      tq.finish();
      t_done = true;
      std::printf("tfinish ok\n");
    } else if (op == "radd") {
      size_t n = 0;
      in >> n;
      std::vector<double> v(n);
      for (auto &x : v) in >> x;
      rc.add(std::move(v));
      std::printf("radd ok\n");
    } else if (op == "rextract") {
      std::vector<double> v = rc.extract();
      std::printf("rextract %zu", v.size());
      for (double x : v) std::printf(" %.17g", x);
      std::printf("\n");
    } else if (op == "jpush") {
      h::ScanJob j;
      size_t n = 0;
      in >> j.stream_id >> n;
      j.input_path = "in" + std::to_string(j.stream_id);
      j.output_path = "out" + std::to_string(j.stream_id);
      j.segments.resize(n);
      for (auto &s : j.segments) in >> s.start >> s.end;
      jq.push(std::move(j));
      ++j_size;
      std::printf("jpush ok\n");
    } else if (op == "jpop") {
      if (j_size == 0 && !j_done) { std::printf("jpop would_block\n"); continue; }
      h::ScanJob j;
      if (jq.pop(j)) {
        --j_size;
        std::printf("jpop 1 %d %s %s %zu", j.stream_id, j.input_path.c_str(), j.output_path.c_str(), j.segments.size());
        for (auto &s : j.segments) std::printf(" %.17g %.17g", s.start, s.end);
        std::printf("\n");
      } else std::printf("jpop 0\n");
    } else if (op == "jfinish") {
      jq.finish();
      j_done = true;
      std::printf("jfinish ok\n");
    } else if (op == "jdone") {
      std::printf("jdone %d\n", jq.is_done() ? 1 : 0);
    } else if (op == "jempty") {
      std::printf("jempty %d\n", jq.empty() ? 1 : 0);
    } else {
      std::printf("unknown %s\n", op.c_str());
      return 2;
    }
  }
  return 0;
}

static int cmd_race(int n, int threads) {
  h::TaskQueue tq;
  h::ResultCollector rc;
  std::vector<std::thread> pool;
  for (int i = 0; i < threads; ++i)
    pool.emplace_back([&] {
      h::ScanTask t{};
      while (tq.pop(t)) rc.add(std::vector<double>{static_cast<double>(t.id), t.start, t.end});
    });
  for (int i = 0; i < n; ++i) tq.push(h::ScanTask{i * 30.0, (i + 1) * 30.0, i});
  tq.finish();
  for (auto &t : pool) t.join();
  std::vector<double> all = rc.extract();
  std::vector<int> seen(static_cast<size_t>(n), 0);
  bool triples_intact = all.size() % 3 == 0;
  for (size_t i = 0; i + 2 < all.size(); i += 3) {
    int id = static_cast<int>(all[i]);
    if (id < 0 || id >= n || all[i + 1] != id * 30.0 || all[i + 2] != (id + 1) * 30.0) triples_intact = false;
    else ++seen[static_cast<size_t>(id)];
  }
  int once = 0;
  for (int c : seen) once += c == 1;
  std::printf("race values %zu once %d of %d triples_intact %d\n", all.size(), once, n, triples_intact ? 1 : 0);
  return once == n && triples_intact ? 0 : 1;
}

// stream s (of S), worker i (of T) -> device, on an N-GPU node: the table process_batch / run_scan_pipeline use
static int cmd_devices(int streams, int threads, int n_dev) {
  for (int s = 0; s < streams; ++s) {
    std::printf("stream %d:", s);
    for (int i = 0; i < threads; ++i) std::printf(" %d", h::worker_device(s * threads, i, n_dev));
    std::printf("\n");
  }
  return 0;
}

// concat <path> < "start end" lines: the cut list the executor would write (ffmpeg_executor.cpp:38-50)
static int cmd_concat(const char *path) {
  std::vector<mt_segment> segs;
  double a, b;
  while (std::cin >> a >> b) segs.push_back(mt_segment{a, b});
  std::fputs(h::concat_list(segs, path).c_str(), stdout);
  return 0;
}

// gate N T R: T threads pass R times through a CpuGate of N tokens; prints the highest number of threads ever
// inside at once, the passes made and whether anybody had to wait; `cpulimit` prints cpu_budget()
static int cmd_gate(int tokens, int threads, int rounds) {
  h::CpuGate gate(tokens);
  std::atomic<int> inside{0}, peak{0}, passes{0};
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t)
    th.emplace_back([&] {
      for (int r = 0; r < rounds; ++r) {
        gate.acquire();
        const int now = ++inside;
        int p = peak.load();
        while (now > p && !peak.compare_exchange_weak(p, now)) {}
        std::this_thread::sleep_for(std::chrono::microseconds(200));
        --inside;
        ++passes;
        gate.release();
      }
    });
  for (auto &x : th) x.join();
  std::printf("peak %d passes %d waited %d tokens %d\n", peak.load(), passes.load(), gate.waits() > 0 ? 1 : 0, gate.tokens());
  return 0;
}

int main(int argc, char **argv) {
  const std::string cmd = argc > 1 ? argv[1] : "";
  if (cmd == "gate" && argc == 5) return cmd_gate(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]));
  if (cmd == "cpulimit") { std::printf("%d\n", h::cpu_budget()); return 0; }
  if (cmd == "window" && argc == 6) {        // local_cpulist gpu_index gpus_on_node want -> the window
    const std::vector<int> w = h::pick_cpu_window(h::parse_cpu_list(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]));
    for (size_t i = 0; i < w.size(); ++i) std::printf("%s%d", i ? "," : "", w[i]);
    std::printf("\n");
    return 0;
  }
  if (cmd == "batchsizing" && argc == 7) {   // n_videos n_devices budget configured_streams configured_threads -> streams threads
    const h::BatchSizing z = h::default_batch_sizing(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]), std::atoi(argv[6]));
    std::printf("%d %d\n", z.streams, z.threads);
    return 0;
  }
  if (cmd == "cpulist" && argc == 3) {       // parse_cpu_list: the numbers, or "invalid"
    try {
      const std::vector<int> v = h::parse_cpu_list(argv[2]);
      for (size_t i = 0; i < v.size(); ++i) std::printf("%s%d", i ? "," : "", v[i]);
      std::printf("\n");
    } catch (const std::invalid_argument &) {
      std::printf("invalid\n");
    }
    return 0;
  }
  if (cmd == "windowpinned") {
    // A worker that is already confined to device 0's window asks for device 1's (what process_batch's stream
    // threads do on a multi-GPU node): both answers must come from the PROCESS's CPUs, not from the asking thread's.
    std::vector<int> w0, w1;
    std::thread t([&] {
      w0 = h::cpu_window_for_device(0);
      (void)h::pin_this_thread(w0);
      std::thread inner([&] { w1 = h::cpu_window_for_device(1); });   // inherits the pinned mask
      inner.join();
    });
    t.join();
    std::printf("allowed %zu window0 %zu window1 %zu\n", h::process_allowed_cpus().size(), w0.size(), w1.size());
    return 0;
  }
  if (cmd == "concat" && argc == 4 && std::string(argv[3]) == "--setlocale") {
    std::setlocale(LC_ALL, "");          // adopt LC_ALL from the environment, as a host application might
    return cmd_concat(argv[2]);
  }
  if (cmd == "concat" && argc == 3) return cmd_concat(argv[2]);
  if (cmd == "config") return cmd_config();
  if (cmd == "memo") return cmd_memo();
  if (cmd == "layout") return cmd_layout();
  if (cmd == "queue") return cmd_queue();
  if (cmd == "race" && argc == 4) return cmd_race(std::atoi(argv[2]), std::atoi(argv[3]));
  if (cmd == "devices" && argc == 5) return cmd_devices(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]));
  std::fprintf(stderr, "usage: host_probe config|layout|queue|race N T\n");
  return 2;
}
