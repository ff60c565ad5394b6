// pin_probe.cpp — what page-locking host memory costs on this box: hipHostMalloc / hipHostFree per size, from one
// thread and from T threads at once (the 64 x T workers of the host layer create their pipes together), and
// hipHostRegister of memory that is already faulted in.  Prints JSON.  hipcc -O2 -o pin_probe pin_probe.cpp
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  (void)hipSetDevice(0);
  void *warm = nullptr;
  (void)hipHostMalloc(&warm, 1 << 20, hipHostMallocDefault);
  (void)hipHostFree(warm);
  std::printf("{\"single_thread\": [");
  const size_t sizes[] = {16ull << 20, 48ull << 20, 128ull << 20, 256ull << 20, 1024ull << 20, 3072ull << 20};
  bool first = true;
  for (size_t sz : sizes) {
    void *p = nullptr;
    double t0 = now();
    hipError_t e = hipHostMalloc(&p, sz, hipHostMallocDefault);
    double t1 = now();
    if (e != hipSuccess) { std::printf("%s{\"MiB\": %zu, \"error\": \"%s\"}", first ? "" : ", ", sz >> 20, hipGetErrorString(e)); first = false; continue; }
    void *d = nullptr;
    (void)hipHostGetDevicePointer(&d, p, 0);
    double t2 = now();
    std::memset(p, 1, sz);
    double t3 = now();
    (void)hipHostFree(p);
    double t4 = now();
    std::printf("%s{\"MiB\": %zu, \"malloc_ms\": %.2f, \"GBps\": %.2f, \"devptr_ms\": %.3f, \"first_touch_ms\": %.2f, \"free_ms\": %.2f}",
                first ? "" : ", ", sz >> 20, (t1 - t0) * 1e3, sz / (t1 - t0) / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3);
    first = false;
    std::fflush(stdout);
  }
  std::printf("], \"register_prefaulted\": [");
  first = true;
  for (size_t sz : {48ull << 20, 1024ull << 20}) {
    void *p = aligned_alloc(2 << 20, sz);
    std::memset(p, 1, sz);
    double t0 = now();
    hipError_t e = hipHostRegister(p, sz, hipHostRegisterDefault);
    double t1 = now();
    if (e == hipSuccess) (void)hipHostUnregister(p);
    double t2 = now();
    free(p);
    std::printf("%s{\"MiB\": %zu, \"register_ms\": %.2f, \"GBps\": %.2f, \"unregister_ms\": %.2f, \"ok\": %d}", first ? "" : ", ", sz >> 20,
                (t1 - t0) * 1e3, sz / (t1 - t0) / 1e9, (t2 - t1) * 1e3, e == hipSuccess);
    first = false;
  }
  std::printf("], \"concurrent_48MiB\": [");
  first = true;
  for (int T : {4, 16, 64}) {
    std::vector<std::thread> th;
    std::vector<double> took(T, 0.0);
    std::vector<void *> ptr(T, nullptr);
    double t0 = now();
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        (void)hipSetDevice(0);
        double a = now();
        (void)hipHostMalloc(&ptr[t], 48ull << 20, hipHostMallocDefault);
        took[t] = now() - a;
      });
    for (auto &x : th) x.join();
    double wall = now() - t0;
    double sum = 0, mx = 0;
    for (double v : took) { sum += v; mx = v > mx ? v : mx; }
    double f0 = now();
    for (void *p : ptr) if (p) (void)hipHostFree(p);
    double f1 = now();
    std::printf("%s{\"threads\": %d, \"wall_ms\": %.1f, \"mean_call_ms\": %.1f, \"max_call_ms\": %.1f, \"GBps\": %.2f, \"free_all_ms\": %.1f}",
                first ? "" : ", ", T, wall * 1e3, sum / T * 1e3, mx * 1e3, T * 48.0 * (1 << 20) / wall / 1e9, (f1 - f0) * 1e3);
    first = false;
    std::fflush(stdout);
  }
  std::printf("], \"concurrent_pipe_shaped_setup\": [");
  // what mtgpu_pipe_create does per worker thread besides pinning: hipSetDevice in a fresh thread, 3 non-blocking
  // streams, 3 events — T threads at once
  first = true;
  for (int T : {1, 16, 64}) {
    std::vector<std::thread> th;
    std::vector<double> t_dev(T, 0.0), t_str(T, 0.0), t_evt(T, 0.0), t_pin(T, 0.0);
    std::vector<hipStream_t> st(T * 3, nullptr);
    std::vector<hipEvent_t> ev(T * 3, nullptr);
    std::vector<void *> ptr(T, nullptr);
    double t0 = now();
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        double a = now();
        (void)hipSetDevice(0);
        double b = now();
        for (int i = 0; i < 3; ++i) (void)hipStreamCreateWithFlags(&st[t * 3 + i], hipStreamNonBlocking);
        double c = now();
        for (int i = 0; i < 3; ++i) (void)hipEventCreateWithFlags(&ev[t * 3 + i], hipEventDisableTiming | hipEventReleaseToSystem);
        double d = now();
        (void)hipHostMalloc(&ptr[t], 16ull << 20, hipHostMallocDefault);
        double e = now();
        t_dev[t] = b - a; t_str[t] = c - b; t_evt[t] = d - c; t_pin[t] = e - d;
      });
    for (auto &x : th) x.join();
    double wall = now() - t0;
    auto mean = [&](const std::vector<double> &v) { double s = 0; for (double x : v) s += x; return s / v.size() * 1e3; };
    std::printf("%s{\"threads\": %d, \"wall_ms\": %.1f, \"mean_ms\": {\"hipSetDevice\": %.2f, \"3_streams\": %.2f, \"3_events\": %.2f, \"pin_16MiB\": %.2f}}",
                first ? "" : ", ", T, wall * 1e3, mean(t_dev), mean(t_str), mean(t_evt), mean(t_pin));
    first = false;
    std::fflush(stdout);
    double f0 = now();
    for (auto s2 : st) if (s2) (void)hipStreamDestroy(s2);
    for (auto e2 : ev) if (e2) (void)hipEventDestroy(e2);
    for (void *p : ptr) if (p) (void)hipHostFree(p);
    (void)f0;
  }
  std::printf("]}\n");
  return 0;
}
