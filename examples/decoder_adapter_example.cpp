// decoder_adapter_example.cpp — how a decoder plugs into the C++ host layer (csrc/host/mtgpu_host.hpp).
//
// The reference's MotionScanner owns an FFmpeg demuxer + decoder (src/motion_scanner.cpp:62-178) and calls
// check_frame on every decoded frame (:376).  With this library the decode half stays whatever it is and is
// presented as a `FrameSource`: seek to the last keyframe at or before a time (:321-325), hand out decoded
// frames in order with their pts and MV side data (:334-354).  Everything behind that — frame filter, pinned
// staging, GPU scan, pooling of the chunks, sort / unique / gap merge / clamp / savings / cut decision — is
// run_scan_pipeline.  A libav-backed FrameSource is csrc/host/libav_source.hpp; the one below needs no FFmpeg:
// it renders a camera scene procedurally (a rectangle crossing a still 1280x720 picture twice), one record per
// 16-px macroblock, a keyframe without MV side data every 25 frames — enough to watch the whole path work.
//
//   g++ -std=c++17 -O2 -Iinclude -Imotion-estimated-video-trimmer_amd/csrc/host examples/decoder_adapter_example.cpp
//       -o decoder_adapter_example -Lmotion-estimated-video-trimmer_amd -lmtgpu -lpthread
//       -Wl,-rpath,$PWD/motion-estimated-video-trimmer_amd          (one command line)
//   ./decoder_adapter_example [threads]        prints the timestamps with motion and the segments to cut
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "mtgpu_host.hpp"

using namespace mtgpu_host;

class SyntheticCamera : public FrameSource {
 public:
  static constexpr int kWidth = 1280, kHeight = 720, kFps = 25, kGop = 25, kFrames = 1500;   // 60 s
  static constexpr int kTbDen = 12800;                                                       // time_base 1/12800
  int width() const override { return kWidth; }
  int height() const override { return kHeight; }
  double duration() const override { return (double)kFrames / kFps; }
  double fps() const override { return kFps; }
  double time_base() const override { return 1.0 / kTbDen; }

  // AVSEEK_FLAG_BACKWARD: continue from the last keyframe at or before `seconds`
  void seek(double seconds) override {
    const int64_t target = static_cast<int64_t>(seconds / time_base());
    int64_t f = target / (kTbDen / kFps);
    if (f >= kFrames) f = kFrames - 1;
    if (f < 0) f = 0;
    pos_ = (int)(f - f % kGop);
  }

  // next decoded frame; its MV bytes stay valid until the next call only (like AVFrame side data)
  bool next(Frame &fr) override {
    if (pos_ >= kFrames) return false;
    const int f = pos_++;
    fr.pts = (int64_t)f * (kTbDen / kFps);
    fr.has_side_data = (f % kGop) != 0;                 // I-frames carry no motion vectors
    fr.mv = nullptr;
    fr.mv_bytes = 0;
    if (!fr.has_side_data) return true;
    render(f);
    fr.mv = mv_.data();
    fr.mv_bytes = mv_.size() * sizeof(mt_mv);
    return true;
  }

  // The scene: a 4x3-macroblock object crosses the picture during seconds [8, 14) and [40, 43); everything else
  // is still (zero vectors), one isolated noisy block flickers all the time and must not count as motion.
  static bool object_at(int f, int &mbx, int &mby) {
    const double t = (double)f / kFps;
    if (t >= 8.0 && t < 14.0) { mbx = 4 + (int)((t - 8.0) * 10.0); mby = 20; return true; }
    if (t >= 40.0 && t < 43.0) { mbx = 60 - (int)((t - 40.0) * 12.0); mby = 9; return true; }
    return false;
  }

 private:
  void render(int f) {
    const int gw = (kWidth + 15) / 16, gh = (kHeight + 15) / 16;
    mv_.assign((size_t)gw * gh, mt_mv{});
    int ox = 0, oy = 0;
    const bool obj = object_at(f, ox, oy);
    for (int y = 0; y < gh; ++y)
      for (int x = 0; x < gw; ++x) {
        mt_mv &m = mv_[(size_t)y * gw + x];
        m.source = -1;
        m.w = m.h = 16;
        m.dst_x = (int16_t)(x * 16 + 8);
        m.dst_y = (int16_t)(y * 16 + 8);
        int dx = 0, dy = 0;
        if (obj && x >= ox && x < ox + 4 && y >= oy && y < oy + 3) { dx = 6; dy = -1; }
        if (x == 70 && y == 30 && (f & 1)) { dx = -9; dy = 4; }            // the lone noisy block
        m.src_x = (int16_t)(m.dst_x - dx);
        m.src_y = (int16_t)(m.dst_y - dy);
        m.motion_x = -dx * 4;
        m.motion_y = -dy * 4;
        m.motion_scale = 4;
      }
  }
  int pos_ = 0;
  std::vector<mt_mv> mv_;
};

int main(int argc, char **argv) {
  const int threads = argc > 1 ? std::atoi(argv[1]) : 4;
  // one record per macroblock can give a cell at most one vote; everything else as the reference's code defaults
  setenv("VECTORS_NEEDED", "1", 0);
  PipelineResult r;
  const int rc = run_scan_pipeline([] { return std::unique_ptr<FrameSource>(new SyntheticCamera()); }, threads, r);
  if (rc != 0) { std::fprintf(stderr, "error: %s\n", r.error.c_str()); return 1; }
  std::printf("chunks %d threads %d frames_scanned %llu motion_frames %zu\n", r.chunks, r.threads,
              (unsigned long long)r.frames_scanned, r.motion_frames);
  std::printf("do_cut %d time_removed %.17g saved_pct %.17g\n", r.merge.do_cut, r.merge.time_removed, r.merge.saved_pct);
  for (const mt_segment &s : r.segments) std::printf("segment %.17g %.17g\n", s.start, s.end);
  return 0;
}
