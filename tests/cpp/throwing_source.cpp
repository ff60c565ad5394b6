// throwing_source.cpp — a decoder that fails in the middle of a video must fail THAT video and nothing else:
// the exception leaves FrameSource::next() while the worker holds a half-filled staging batch and a CPU token
// (mtgpu_host::CpuGate).  With MTGPU_CPU_TOKENS=1 a token that is not given back would park every other worker
// for good.  Usage: throwing_source <threads> <throw_at_frame | -1>; prints "rc <code> error <text>" and, when
// nothing is thrown, the number of motion frames.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <vector>

#include "mtgpu_host.hpp"

using namespace mtgpu_host;

class Camera : public FrameSource {
  int pos_ = 0, throw_at_;
  std::vector<mt_mv> mv_;
 public:
  static constexpr int kW = 640, kH = 480, kFps = 25, kFrames = 1500, kTb = 12800;
  explicit Camera(int throw_at) : throw_at_(throw_at), mv_((kW / 16) * (kH / 16)) {}
  int width() const override { return kW; }
  int height() const override { return kH; }
  double duration() const override { return (double)kFrames / kFps; }
  double fps() const override { return kFps; }
  double time_base() const override { return 1.0 / kTb; }
  void seek(double seconds) override {
    long f = (long)(seconds / time_base()) / (kTb / kFps);
    pos_ = (int)(f < 0 ? 0 : f >= kFrames ? kFrames - 1 : f);
  }
  bool next(Frame &fr) override {
    if (pos_ >= kFrames) return false;
    const int f = pos_++;
    if (f == throw_at_) throw std::runtime_error("decoder lost the stream at frame " + std::to_string(f));
    fr.pts = (int64_t)f * (kTb / kFps);
    fr.has_side_data = true;
    const bool moving = (f / 100) % 2 == 1;
    for (int y = 0; y < kH / 16; ++y)
      for (int x = 0; x < kW / 16; ++x) {
        mt_mv &r = mv_[(size_t)y * (kW / 16) + x];
        r = mt_mv{};
        r.dst_x = (int16_t)(x * 16 + 8); r.dst_y = (int16_t)(y * 16 + 8);
        const bool obj = moving && x >= 10 && x < 14 && y >= 10 && y < 13;
        r.src_x = (int16_t)(r.dst_x - (obj ? 6 : 0)); r.src_y = r.dst_y;
      }
    fr.mv = mv_.data();
    fr.mv_bytes = mv_.size() * sizeof(mt_mv);
    return true;
  }
};

int main(int argc, char **argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 3;
  const int throw_at = argc > 2 ? std::atoi(argv[2]) : -1;
  PipelineResult r;
  const int rc = run_scan_pipeline([&] { return std::unique_ptr<FrameSource>(new Camera(throw_at)); }, threads, r);
  std::printf("rc %d motion %zu tokens %d error %s\n", rc, r.motion_frames, CpuGate::instance().tokens(), r.error.c_str());
  return 0;
}
