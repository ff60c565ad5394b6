// TEST INFRASTRUCTURE ONLY (see oracle/mt_oracle.h).  A probe around the parts of the reference's
// scan path that DO compile here from their own sources with nothing but libstdc++:
//   include/motion_trim/config.hpp   (env parsing + defaults of every scan/merge parameter)
//   include/motion_trim/types.hpp    (TimeSegment / ScanTask layout)
//   src/task_queue.cpp               (TaskQueue, ResultCollector)
//   src/ffmpeg_queue.cpp             (FFmpegQueue / FFmpegJob — the consumer contract of the merge)
// `make -C oracle ref` compiles this file together with those two reference .cpp files, taken where
// they lie under /root/reference, into oracle/_ref/ref_host_probe.  Nothing of the reference is
// copied into the repo; this file only CALLS it and prints what it answered, so that
// tests/golden/make_reference_host_vectors.py can record the answers as fixtures.
//
//   ref_host_probe config            -> one line per Config getter, values as %.17g / raw bits
//   ref_host_probe memo   < script   -> get / setenv / get again: the getters are function-local statics
//   ref_host_probe layout            -> sizeof / alignof / offsets of TimeSegment, ScanTask
//   ref_host_probe queue  < script   -> TaskQueue / ResultCollector / FFmpegQueue driven by a script
//   ref_host_probe race N T          -> T threads drain N tasks; prints how often each id was popped
//   ref_host_probe sizing            -> detect_cpu_limit(), calculate_parallel_streams(), get_available_cpus() of
//                                       src/system.cpp on THIS machine (only when the recipe could build system.cpp:
//                                       it needs <fmt/core.h>, taken from the fmt headers the image ships inside
//                                       PyTorch — the real library, header-only mode, not a stand-in)
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <typeinfo>
#include <vector>

#include "motion_trim/config.hpp"
#include "motion_trim/ffmpeg_queue.hpp"
#include "motion_trim/task_queue.hpp"
#include "motion_trim/types.hpp"

#ifdef MT_REF_HAVE_SYSTEM
#include "motion_trim/system.hpp"
#endif

namespace mt = motion_trim;

static int cmd_sizing() {
#ifdef MT_REF_HAVE_SYSTEM
  std::printf("detect_cpu_limit %d\n", mt::detect_cpu_limit());
  try {
    std::printf("calculate_parallel_streams %d\n", mt::calculate_parallel_streams());
  } catch (const std::exception &e) {
    std::printf("calculate_parallel_streams throws %s\n", typeid(e).name());
  }
  const std::vector<int> cpus = mt::get_available_cpus();
  std::printf("available_cpus %zu first %d last %d\n", cpus.size(), cpus.empty() ? -1 : cpus.front(), cpus.empty() ? -1 : cpus.back());
  return 0;
#else
  std::printf("unavailable\n");
  return 0;
#endif
}

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
  show_d("mv_threshold_sq", [] { return mt::Config::mv_threshold_sq(); });
  show_i("block_size", [] { return mt::Config::block_size(); });
  show_i("block_shift", [] { return mt::Config::block_shift(); });
  show_i("vectors_needed", [] { return mt::Config::vectors_needed(); });
  show_i("clusters_needed", [] { return mt::Config::clusters_needed(); });
  show_f("vertical_mask", [] { return mt::Config::vertical_mask(); });
  show_d("max_gap_sec", [] { return mt::Config::max_gap_sec(); });
  show_d("padding_sec", [] { return mt::Config::padding_sec(); });
  show_d("chunk_duration_sec", [] { return mt::Config::chunk_duration_sec(); });
  show_d("target_fps", [] { return mt::Config::target_fps(); });
  show_d("min_savings_pct", [] { return mt::Config::min_savings_pct(); });
  show_i("parallel_streams", [] { return mt::Config::parallel_streams(); });
  show_i("threads_per_stream", [] { return mt::Config::threads_per_stream(); });
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
      if (a == "mv_threshold_sq") show_d("mv_threshold_sq", [] { return mt::Config::mv_threshold_sq(); });
      else if (a == "block_size") show_i("block_size", [] { return mt::Config::block_size(); });
      else if (a == "block_shift") show_i("block_shift", [] { return mt::Config::block_shift(); });
      else if (a == "vectors_needed") show_i("vectors_needed", [] { return mt::Config::vectors_needed(); });
      else if (a == "clusters_needed") show_i("clusters_needed", [] { return mt::Config::clusters_needed(); });
      else if (a == "vertical_mask") show_f("vertical_mask", [] { return mt::Config::vertical_mask(); });
      else if (a == "max_gap_sec") show_d("max_gap_sec", [] { return mt::Config::max_gap_sec(); });
      else if (a == "padding_sec") show_d("padding_sec", [] { return mt::Config::padding_sec(); });
      else if (a == "chunk_duration_sec") show_d("chunk_duration_sec", [] { return mt::Config::chunk_duration_sec(); });
      else if (a == "target_fps") show_d("target_fps", [] { return mt::Config::target_fps(); });
      else if (a == "min_savings_pct") show_d("min_savings_pct", [] { return mt::Config::min_savings_pct(); });
      else if (a == "parallel_streams") show_i("parallel_streams", [] { return mt::Config::parallel_streams(); });
      else if (a == "threads_per_stream") show_i("threads_per_stream", [] { return mt::Config::threads_per_stream(); });
      else { std::printf("unknown getter %s\n", a.c_str()); return 2; }
    } else { std::printf("unknown %s\n", op.c_str()); return 2; }
  }
  return 0;
}

static int cmd_layout() {
  std::printf("TimeSegment size %zu align %zu start %zu end %zu\n", sizeof(mt::TimeSegment), alignof(mt::TimeSegment),
              offsetof(mt::TimeSegment, start), offsetof(mt::TimeSegment, end));
  std::printf("ScanTask size %zu align %zu start %zu end %zu id %zu\n", sizeof(mt::ScanTask), alignof(mt::ScanTask),
              offsetof(mt::ScanTask, start), offsetof(mt::ScanTask, end), offsetof(mt::ScanTask, id));
  std::printf("CACHE_LINE_SIZE %zu\n", static_cast<size_t>(mt::CACHE_LINE_SIZE));
  return 0;
}

// Script lines (single-threaded, so a pop on an empty unfinished queue — which would block — is refused):
//   tpush <start> <end> <id> | tpop | tfinish
//   radd <n> <v1> ... <vn>   | rextract
//   jpush <stream_id> <nseg> <s1> <e1> ... | jpop | jfinish | jdone | jempty
static int cmd_queue() {
  mt::TaskQueue tq;
  mt::ResultCollector rc;
  mt::FFmpegQueue jq;
  size_t t_size = 0, j_size = 0;
  bool t_done = false, j_done = false;
  std::string line;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string op;
    if (!(in >> op) || op[0] == '#') continue;
    if (op == "tpush") {
      mt::ScanTask t{};
      in >> t.start >> t.end >> t.id;
      tq.push(t);
      ++t_size;
      std::printf("tpush ok\n");
    } else if (op == "tpop") {
      if (t_size == 0 && !t_done) { std::printf("tpop would_block\n"); continue; }
      mt::ScanTask t{};
      bool ok = tq.pop(t);
      if (ok) { --t_size; std::printf("tpop 1 %.17g %.17g %d\n", t.start, t.end, t.id); }
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
      mt::FFmpegJob j;
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
      mt::FFmpegJob j;
      bool ok = jq.pop(j);
      if (ok) {
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

// T workers drain N tasks pushed by the main thread, each worker adds the ids it popped (as doubles) to
// the ResultCollector; prints total count and the number of ids seen exactly once (order is free).
static int cmd_race(int n, int threads) {
  mt::TaskQueue tq;
  mt::ResultCollector rc;
  std::vector<std::thread> pool;
  for (int i = 0; i < threads; ++i)
    pool.emplace_back([&] {
      mt::ScanTask t{};
      while (tq.pop(t)) {
        std::vector<double> one{static_cast<double>(t.id), t.start, t.end};
        rc.add(std::move(one));
      }
    });
  for (int i = 0; i < n; ++i) tq.push(mt::ScanTask{i * 30.0, (i + 1) * 30.0, i});
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

int main(int argc, char **argv) {
  const std::string cmd = argc > 1 ? argv[1] : "";
  if (cmd == "sizing") return cmd_sizing();
  if (cmd == "config") return cmd_config();
  if (cmd == "memo") return cmd_memo();
  if (cmd == "layout") return cmd_layout();
  if (cmd == "queue") return cmd_queue();
  if (cmd == "race" && argc == 4) return cmd_race(std::atoi(argv[2]), std::atoi(argv[3]));
  std::fprintf(stderr, "usage: ref_host_probe config|layout|queue|race N T\n");
  return 2;
}
