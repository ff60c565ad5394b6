// pack_bench.cpp — CPU microbenchmark of the host dispatcher's copy-out (csrc/pack_simd.cpp):
// T threads, each packing its OWN DRAM-resident array of 40-byte records (first-touched by the thread
// itself, far larger than its L3 share) frame by frame into a ring of staging the size of a worker's
// pipe (3 x 16 MiB), exactly as mtgpu_batch_add_frame does.  Source GB/s per thread and in total, for
// every loop this CPU can run x {ordinary, non-temporal} stores x prefetch distances, plus the 40-byte
// memcpy of the aos40 layout and a read-only pass (the ceiling of "reading the source").
//
// Build + run (needs only libmtgpu.so; the destination is pinned with hipHostMalloc when a device is
// visible — loaded lazily from libamdhip64 — otherwise 64-byte-aligned malloc):
//   make -C scripts/micro pack_bench && scripts/micro/pack_bench --threads 1,4,16 --mb 512
// Prints one JSON object.
#include <dlfcn.h>
#include <pthread.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mtgpu.h"

namespace {

struct Barrier {                      // sense-reversing spin barrier (threads are <= cores here)
  std::atomic<int> count{0}, gen{0};
  int n;
  explicit Barrier(int n_) : n(n_) {}
  void wait() {
    const int g = gen.load();
    if (count.fetch_add(1) + 1 == n) { count.store(0); gen.fetch_add(1); }
    else while (gen.load() == g) std::this_thread::yield();
  }
};

using HostMallocFn = int (*)(void **, size_t, unsigned);
using HostFreeFn = int (*)(void *);
HostMallocFn hip_host_malloc = nullptr;
HostFreeFn hip_host_free = nullptr;

bool load_hip() {
  if (mtgpu_device_count() < 1) return false;
  void *h = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return false;
  hip_host_malloc = reinterpret_cast<HostMallocFn>(dlsym(h, "hipHostMalloc"));
  hip_host_free = reinterpret_cast<HostFreeFn>(dlsym(h, "hipHostFree"));
  return hip_host_malloc && hip_host_free;
}

struct Variant { std::string name; int flags; };   // flags < 0: special loops (-1 memcpy40, -2 read-only)

double run(const Variant &v, int threads, size_t src_bytes, size_t frame_rec, size_t ring_bytes, bool pinned, int reps,
           std::vector<unsigned char *> &src, std::vector<unsigned char *> &ring) {
  Barrier bar(threads + 1);
  std::vector<std::thread> th;
  std::atomic<uint64_t> sink{0};
  const size_t frames = src_bytes / (frame_rec * 40);
  const size_t out_per_frame = (v.flags == -1 ? 40 : 8) * frame_rec;
  for (int t = 0; t < threads; ++t)
    th.emplace_back([&, t] {
      unsigned char *s = src[t], *d = ring[t];
      bar.wait();
      uint64_t acc = 0;
      for (int r = 0; r < reps; ++r) {
        size_t pos = 0;
        for (size_t f = 0; f < frames; ++f) {
          const unsigned char *fs = s + f * frame_rec * 40;
          if (pos + out_per_frame > ring_bytes) pos = 0;
          if (v.flags == -1) std::memcpy(d + pos, fs, frame_rec * 40);
          else if (v.flags == -2) { for (size_t i = 0; i < frame_rec; ++i) { uint64_t x; std::memcpy(&x, fs + i * 40 + 6, 8); acc += x; } }
          else mtgpu_pack_records_with(v.flags, fs, frame_rec, d + pos);
          pos += out_per_frame;
        }
      }
      sink += acc;
      bar.wait();
    });
  bar.wait();
  const auto t0 = std::chrono::steady_clock::now();
  bar.wait();
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (auto &x : th) x.join();
  (void)pinned;
  return (double)threads * (double)frames * (double)frame_rec * 40.0 * reps / sec / 1e9;   // source GB/s, all threads
}

}  // namespace

int main(int argc, char **argv) {
  std::vector<int> thread_counts{1, 4, 16};
  size_t mb = 512, frame_rec = 32640, ring_mb = 48;
  int reps = 2;
  bool want_pinned = true;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&] { return i + 1 < argc ? std::string(argv[++i]) : std::string(); };
    if (a == "--threads") { thread_counts.clear(); std::string s = next(); size_t p = 0; while (p < s.size()) { thread_counts.push_back(std::atoi(s.c_str() + p)); p = s.find(',', p); if (p == std::string::npos) break; ++p; } }
    else if (a == "--mb") mb = std::strtoull(next().c_str(), nullptr, 10);
    else if (a == "--frame-records") frame_rec = std::strtoull(next().c_str(), nullptr, 10);
    else if (a == "--ring-mb") ring_mb = std::strtoull(next().c_str(), nullptr, 10);
    else if (a == "--reps") reps = std::atoi(next().c_str());
    else if (a == "--malloc") want_pinned = false;
  }
  const bool pinned = want_pinned && load_hip();
  const size_t src_bytes = mb << 20, ring_bytes = ring_mb << 20;
  int max_t = 0;
  for (int t : thread_counts) max_t = t > max_t ? t : max_t;

  std::vector<Variant> variants{{"read_only_8_of_40", -2}, {"memcpy_40B_records", -1}, {"scalar", MT_PACK_SCALAR}};
  unsigned char probe_s[40 * 16] = {0}, probe_d[8 * 16 + 64];
  for (int impl : {MT_PACK_AVX2, MT_PACK_AVX512}) {
    if (mtgpu_pack_records_with(impl, probe_s, 16, probe_d) != MT_OK) continue;
    const std::string base = impl == MT_PACK_AVX2 ? "avx2" : "avx512";
    variants.push_back({base, impl});
    variants.push_back({base + "_nt", impl | MT_PACK_NT});
    variants.push_back({base + "_nt_pf1k", impl | MT_PACK_NT | MT_PACK_PREFETCH_LINES(16)});
    variants.push_back({base + "_nt_pf4k", impl | MT_PACK_NT | MT_PACK_PREFETCH_LINES(64)});
    variants.push_back({base + "_pf1k", impl | MT_PACK_PREFETCH_LINES(16)});
  }

  // every thread first-touches its own source and ring (NUMA-local, like a decoder's own output)
  std::vector<unsigned char *> src(max_t, nullptr), ring(max_t, nullptr);
  {
    std::vector<std::thread> th;
    for (int t = 0; t < max_t; ++t)
      th.emplace_back([&, t] {
        src[t] = static_cast<unsigned char *>(aligned_alloc(4096, src_bytes));
        uint64_t x = 0x9E3779B97F4A7C15ull * (t + 1);
        for (size_t i = 0; i + 8 <= src_bytes; i += 8) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; std::memcpy(src[t] + i, &x, 8); }
        if (!pinned) { ring[t] = static_cast<unsigned char *>(aligned_alloc(4096, ring_bytes)); std::memset(ring[t], 1, ring_bytes); }
      });
    for (auto &x : th) x.join();
    if (pinned)
      for (int t = 0; t < max_t; ++t) {
        void *p = nullptr;
        if (hip_host_malloc(&p, ring_bytes, 0) != 0) { std::fprintf(stderr, "hipHostMalloc failed\n"); return 1; }
        ring[t] = static_cast<unsigned char *>(p);
        std::memset(ring[t], 1, ring_bytes);
      }
  }
  std::printf("{\"selected\": %d, \"destination\": \"%s\", \"source_mb_per_thread\": %zu, \"frame_records\": %zu, "
              "\"ring_mb\": %zu, \"reps\": %d, \"unit\": \"GB/s of 40-byte source records, all threads\", \"results\": {",
              mtgpu_pack_selected(), pinned ? "hipHostMalloc" : "malloc", mb, frame_rec, ring_mb, reps);
  bool first_t = true;
  for (int t : thread_counts) {
    std::printf("%s\"%d_threads\": {", first_t ? "" : ", ", t);
    first_t = false;
    bool first_v = true;
    for (const Variant &v : variants) {
      (void)run(v, t, src_bytes / 4, frame_rec, ring_bytes, pinned, 1, src, ring);           // warm-up
      const double gbs = run(v, t, src_bytes, frame_rec, ring_bytes, pinned, reps, src, ring);
      std::printf("%s\"%s\": %.2f", first_v ? "" : ", ", v.name.c_str(), gbs);
      std::fflush(stdout);
      first_v = false;
    }
    std::printf("}");
  }
  std::printf("}}\n");
  for (int t = 0; t < max_t; ++t) { free(src[t]); if (pinned) hip_host_free(ring[t]); else free(ring[t]); }
  return 0;
}
