// device_entry_stress.cpp — the device entry points of ONE context entered from many host threads, each on its own HIP
// stream, small batches back to back (three different batches per thread in turn, three launches in flight per stream).
// Every launch takes its work-list memory from the context's scratch ring; a launch that read another launch's list would
// answer for the wrong frames.  Expected flags are known by construction: a frame says yes iff it was given two moving
// records in neighbouring cells (VECTORS_NEEDED 1, CLUSTERS_NEEDED 1); frames without side data say no.
// Usage: device_entry_stress [threads [iterations]]   prints "ok <launches> launches" or the first mismatches; exit 0 / 1.
// Built and run by tests/test_gpu_parity.py::test_device_entry_points_under_real_thread_concurrency.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "mtgpu.h"

namespace {

struct Batch {
  std::vector<mt_mv> mv;
  std::vector<uint64_t> off;
  std::vector<uint8_t> sd, want;
  mt_mv *d_mv = nullptr;
  uint64_t *d_off = nullptr;
  uint8_t *d_sd = nullptr, *d_flags = nullptr;
};

uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

void make_batch(Batch &b, uint64_t seed) {
  const int frames = 3 + (int)(mix(seed) % 14);
  b.off.push_back(0);
  for (int f = 0; f < frames; ++f) {
    const uint64_t h = mix(seed * 1000 + (uint64_t)f);
    const bool has_sd = (h & 7) != 0;                              // one frame in eight: no side data
    const bool motion = has_sd && ((h >> 3) & 1);
    const int n = has_sd ? 500 + (int)((h >> 8) % 3000) : 0;
    for (int i = 0; i < n; ++i) {
      mt_mv v;
      std::memset(&v, 0xA5, sizeof v);                             // junk in every byte the scan must ignore
      const uint64_t g = mix(h + (uint64_t)i);
      v.dst_x = (int16_t)(g % 1920); v.dst_y = (int16_t)((g >> 16) % 1080);
      v.src_x = (int16_t)(v.dst_x - (int16_t)((g >> 32) % 3)); v.src_y = v.dst_y;      // |d|^2 <= 4 < 16
      if (motion && i < 2) {                                       // cells (50 + i, 30): neighbours, each with one vote
        v.dst_x = (int16_t)(16 * (50 + i) + 8); v.dst_y = (int16_t)(16 * 30 + 8);
        v.src_x = (int16_t)(v.dst_x - 9); v.src_y = v.dst_y;
      }
      b.mv.push_back(v);
    }
    b.off.push_back(b.mv.size());
    b.sd.push_back(has_sd ? 1 : 0);
    b.want.push_back(motion ? 1 : 0);
  }
}

#define CHECK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); std::exit(2); } } while (0)

}  // namespace

int main(int argc, char **argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 16, iters = argc > 2 ? std::atoi(argv[2]) : 150;
  mt_scan_params p;
  mtgpu_ctx *ctx = nullptr;
  if (mtgpu_params_from_config(&p, 1920, 1080, 16.0, 16, 4, 1, 1, 0.05f) != MT_OK || mtgpu_create(&p, 0, &ctx) != MT_OK) {
    std::fprintf(stderr, "mtgpu: %s\n", mtgpu_last_error());
    return 2;
  }
  std::atomic<long> launches{0}, bad{0};
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t)
    pool.emplace_back([&, t] {
      CHECK(hipSetDevice(0));
      hipStream_t st;
      CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      Batch b[3];
      for (int k = 0; k < 3; ++k) {
        make_batch(b[k], (uint64_t)t * 16 + (uint64_t)k + 1);
        const size_t nf = b[k].sd.size();
        CHECK(hipMalloc(reinterpret_cast<void **>(&b[k].d_mv), sizeof(mt_mv) * (b[k].mv.size() + 1)));
        CHECK(hipMalloc(reinterpret_cast<void **>(&b[k].d_off), sizeof(uint64_t) * (nf + 1)));
        CHECK(hipMalloc(reinterpret_cast<void **>(&b[k].d_sd), nf));
        CHECK(hipMalloc(reinterpret_cast<void **>(&b[k].d_flags), nf));
        CHECK(hipMemcpy(b[k].d_mv, b[k].mv.data(), sizeof(mt_mv) * b[k].mv.size(), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b[k].d_off, b[k].off.data(), sizeof(uint64_t) * (nf + 1), hipMemcpyHostToDevice));
        CHECK(hipMemcpy(b[k].d_sd, b[k].sd.data(), nf, hipMemcpyHostToDevice));
      }
      std::vector<uint8_t> got;
      for (int it = 0; it < iters; ++it) {
        for (int k = 0; k < 3; ++k) {
          CHECK(hipMemsetAsync(b[k].d_flags, 7, b[k].sd.size(), st));       // poisoned: every flag must be written
          if (mtgpu_scan_frames_device(ctx, b[k].d_mv, b[k].mv.size(), b[k].d_off, b[k].d_sd, (uint32_t)b[k].sd.size(),
                                       b[k].d_flags, st) != MT_OK) {
            std::fprintf(stderr, "thread %d: %s\n", t, mtgpu_last_error());
            std::exit(2);
          }
          ++launches;
        }
        CHECK(hipStreamSynchronize(st));
        for (int k = 0; k < 3; ++k) {
          got.resize(b[k].sd.size());
          CHECK(hipMemcpy(got.data(), b[k].d_flags, got.size(), hipMemcpyDeviceToHost));
          if (std::memcmp(got.data(), b[k].want.data(), got.size()) != 0 && bad++ < 5) {
            std::fprintf(stderr, "thread %d iteration %d batch %d: flags differ:", t, it, k);
            for (size_t f = 0; f < got.size(); ++f) std::fprintf(stderr, " %d/%d", got[f], b[k].want[f]);
            std::fprintf(stderr, "\n");
          }
        }
      }
      for (int k = 0; k < 3; ++k) { (void)hipFree(b[k].d_mv); (void)hipFree(b[k].d_off); (void)hipFree(b[k].d_sd); (void)hipFree(b[k].d_flags); }
      (void)hipStreamDestroy(st);
    });
  for (auto &th : pool) th.join();
  mtgpu_destroy(ctx);
  if (bad.load() != 0) { std::printf("MISMATCH in %ld of %ld launches\n", bad.load(), launches.load()); return 1; }
  std::printf("ok %ld launches\n", launches.load());
  return 0;
}
