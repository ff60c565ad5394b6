// integration_recipe.cpp — the adapter of INTEGRATION.md section 1 ("feed the pipe instead of calling check_frame per
// frame"), member for member, around a stand-in for the decode loop: what a maintainer pastes into MotionScanner must
// compile against include/mtgpu.h as written and give the oracle's answer.  The "decoder" hands out 90 frames at 30 fps,
// frames 30..59 carry a 2x1-cell object (the scene of examples/scan_example.c); batches are kept tiny (64 records) so that
// the loop runs through MT_ERR_CAPACITY, submit, MT_ERR_BUSY back-pressure and the final drain many times.
// Prints "motion <n> first <pts> last <pts>".   Built by tests/test_host_cpp.py (CPU: compile + link; GPU: run).
#include <cstdint>
#include <cstdio>
#include <vector>

#include "mtgpu.h"

#define LOG_ERROR(fmt, msg) std::fprintf(stderr, "mtgpu: %s\n", msg)
struct AVFrameSideData { uint8_t *data; size_t size; };          // the two fields of FFmpeg's struct the recipe reads
struct AVMotionVector { unsigned char bytes[40]; };               // layout: include/mt_types.h (mt_mv)

class MotionScanner {
  // include/motion_trim/motion_scanner.hpp — new private members
  mtgpu_ctx *gpu_ = nullptr;
  mtgpu_pipe *pipe_ = nullptr;
  mtgpu_batch *batch_ = nullptr;
  int in_flight_ = 0;

  bool collect_one(std::vector<double> &ts) {
    mtgpu_batch *done = nullptr; const uint8_t *flags; const double *pts; uint32_t n;
    --in_flight_;
    if (mtgpu_pipe_collect(pipe_, &done, &flags, &pts, nullptr, &n) != MT_OK) {
      LOG_ERROR("mtgpu: {}", mtgpu_last_error());
      if (done) mtgpu_pipe_release(pipe_, done);
      return false;
    }
    for (uint32_t i = 0; i < n; ++i)
      if (flags[i]) ts.push_back(pts[i]);
    return mtgpu_pipe_release(pipe_, done) == MT_OK;
  }
  bool next_batch(std::vector<double> &ts) {
    int rc;
    while ((rc = mtgpu_pipe_acquire(pipe_, &batch_)) == MT_ERR_BUSY)
      if (!collect_one(ts)) return false;
    return rc == MT_OK;
  }

 public:
  bool initialize(int width, int height, uint64_t records_per_batch) {
    mt_scan_params p;
    if (mtgpu_params_from_config(&p, width, height, 16.0, 16, 4, 2, 2, 0.05f) != MT_OK ||
        mtgpu_create(&p, 0, &gpu_) != MT_OK ||
        mtgpu_pipe_create(gpu_, records_per_batch, 4096, 3, &pipe_) != MT_OK) {
      LOG_ERROR("mtgpu: {}", mtgpu_last_error());
      return false;
    }
    return true;
  }
  // one decoded frame: the body that replaces lines 375-383 of src/motion_scanner.cpp
  bool on_frame(const AVFrameSideData *sd, double pts, std::vector<double> &ts) {
    static_assert(sizeof(AVMotionVector) == MT_MV_BYTES, "record layout");
    for (;;) {
      if (!batch_ && !next_batch(ts)) return false;
      int rc = mtgpu_batch_add_frame(batch_, sd ? sd->data : nullptr, sd ? sd->size : 0, sd != nullptr, pts, 0);
      if (rc == MT_OK) break;
      if (rc != MT_ERR_CAPACITY) { LOG_ERROR("mtgpu: {}", mtgpu_last_error()); return false; }
      if (mtgpu_pipe_submit(pipe_, batch_) != MT_OK) { LOG_ERROR("mtgpu: {}", mtgpu_last_error()); return false; }
      ++in_flight_;
      batch_ = nullptr;
    }
    return true;
  }
  // where scan_range returns `ts`: false = a batch was lost (the caller fails the chunk, as scan_range's callers do
  // when initialize() fails) — never a silently shorter `ts`
  bool finish(std::vector<double> &ts) {
    bool ok = true;
    if (batch_) {
      if (mtgpu_pipe_submit(pipe_, batch_) == MT_OK) ++in_flight_;
      else { LOG_ERROR("mtgpu: {}", mtgpu_last_error()); mtgpu_pipe_release(pipe_, batch_); ok = false; }   // a failed submit leaves the batch ours
      batch_ = nullptr;
    }
    while (in_flight_ > 0)
      if (!collect_one(ts)) ok = false;          // keep draining: every batch in flight is collected and released
    return ok;
  }
  ~MotionScanner() { mtgpu_pipe_destroy(pipe_); mtgpu_destroy(gpu_); }
};

int main() {
  MotionScanner s;
  if (!s.initialize(1920, 1080, 64)) return 1;
  std::vector<double> ts;
  enum { F = 90, PER = 4 };
  for (int f = 0; f < F; ++f) {
    std::vector<mt_mv> mv;
    if (f >= 30 && f < 60)
      for (int k = 0; k < PER; ++k) {
        mt_mv v{};
        v.dst_x = (int16_t)(16 * (40 + k / 2) + 8);
        v.dst_y = (int16_t)(16 * 30 + 8);
        v.src_x = (int16_t)(v.dst_x - 6);
        v.src_y = v.dst_y;
        v.w = v.h = 8;
        v.source = -1;
        mv.push_back(v);
      }
    AVFrameSideData sd{reinterpret_cast<uint8_t *>(mv.data()), mv.size() * sizeof(mt_mv)};
    const bool has_sd = f % 15 != 0;                              // every 15th frame: no MV side data at all
    if (!s.on_frame(has_sd ? &sd : nullptr, f / 30.0, ts)) return 1;
  }
  if (!s.finish(ts)) return 1;
  std::printf("motion %zu first %.6f last %.6f\n", ts.size(), ts.empty() ? -1.0 : ts.front(), ts.empty() ? -1.0 : ts.back());
  return 0;
}
