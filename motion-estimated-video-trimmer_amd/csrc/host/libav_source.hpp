// libav_source.hpp — FFmpeg-backed FrameSource for the C++ host layer.
//
// NOT COMPILED OR TESTED IN THIS IMAGE: FFmpeg headers and libraries are absent here
// (SURVEY.md "container facts"), so this file is only built where they exist:
//     g++ -std=c++17 -DMTGPU_WITH_LIBAV -I include ... -lavformat -lavcodec -lavutil
// It keeps decode where the reference keeps it — on the host, one decoder per worker
// (include/motion_trim/motion_scanner.hpp:8-13) — and hands every decoded frame's
// AV_FRAME_DATA_MOTION_VECTORS bytes to GpuMotionScanner through the FrameSource interface.
// Decoder settings follow what the reference configures (src/motion_scanner.cpp:148-172):
// +export_mvs, no loop filter, no IDCT, B-frames skipped, fast + gray flags, one slice thread.
// Difference from the reference, on purpose: the reference maps the whole file and reads it through custom
// AVIO callbacks over that buffer (MemoryLoader::read / ::seek, src/memory_io.cpp:134-166, attached at
// src/motion_scanner.cpp:71-97 with AVFMT_FLAG_CUSTOM_IO); memory_io is untouched and out of scope (SURVEY.md
// §2 #8), so this source lets libavformat open the PATH itself.  Same packets, same decoder, same side data;
// a maintainer who wants the mmap'ed input back passes their own AVIOContext in fmt_->pb before
// avformat_open_input, exactly as the reference does.
// SURVEY.md §8 row f4 stays "not built" until this file has been compiled and run against real FFmpeg.
#pragma once

#if !defined(MTGPU_WITH_LIBAV)
#error "libav_source.hpp needs FFmpeg: compile with -DMTGPU_WITH_LIBAV where libavcodec/libavformat are installed"
#else

extern "C" {
#include <libavcodec/avcodec.h>
#include <libavformat/avformat.h>
#include <libavutil/motion_vector.h>
}

#include <cstddef>
#include <stdexcept>
#include <string>

#include "mtgpu_host.hpp"

namespace mtgpu_host {

// The scan kernels read the records as mt_mv: the layouts must be identical.
static_assert(sizeof(AVMotionVector) == sizeof(mt_mv), "AVMotionVector is not 40 bytes");
static_assert(offsetof(AVMotionVector, src_x) == offsetof(mt_mv, src_x), "src_x offset");
static_assert(offsetof(AVMotionVector, src_y) == offsetof(mt_mv, src_y), "src_y offset");
static_assert(offsetof(AVMotionVector, dst_x) == offsetof(mt_mv, dst_x), "dst_x offset");
static_assert(offsetof(AVMotionVector, dst_y) == offsetof(mt_mv, dst_y), "dst_y offset");

class LibavSource : public FrameSource {
  AVFormatContext *fmt_ = nullptr;
  AVCodecContext *dec_ = nullptr;
  AVFrame *frame_ = nullptr;
  AVPacket *pkt_ = nullptr;
  int vs_ = -1;

  void fail(const char *what) { close(); throw std::runtime_error(what); }
  void close() {
    if (dec_) avcodec_free_context(&dec_);
    if (fmt_) avformat_close_input(&fmt_);
    if (frame_) av_frame_free(&frame_);
    if (pkt_) av_packet_free(&pkt_);
  }

 public:
  explicit LibavSource(const std::string &path) {
    frame_ = av_frame_alloc();
    pkt_ = av_packet_alloc();
    if (!frame_ || !pkt_) fail("av_frame_alloc / av_packet_alloc");
    if (avformat_open_input(&fmt_, path.c_str(), nullptr, nullptr) < 0) fail("avformat_open_input");
    if (avformat_find_stream_info(fmt_, nullptr) < 0) fail("avformat_find_stream_info");
    vs_ = av_find_best_stream(fmt_, AVMEDIA_TYPE_VIDEO, -1, -1, nullptr, 0);
    if (vs_ < 0) fail("no video stream");
    for (unsigned i = 0; i < fmt_->nb_streams; ++i)
      if ((int)i != vs_) fmt_->streams[i]->discard = AVDISCARD_ALL;
    const AVCodecParameters *par = fmt_->streams[vs_]->codecpar;
    const AVCodec *codec = avcodec_find_decoder(par->codec_id);
    if (!codec)      // the reference's by-name fallback for builds whose id table lacks the entry (src/motion_scanner.cpp:126-131)
      codec = avcodec_find_decoder_by_name(par->codec_id == AV_CODEC_ID_HEVC ? "hevc" : "h264");
    if (!codec) fail("no decoder");
    dec_ = avcodec_alloc_context3(codec);
    if (!dec_ || avcodec_parameters_to_context(dec_, par) < 0) fail("decoder context");
    // Decoder configuration through AVOptions (one table, applied by avcodec_open2): the scan needs
    // the exported motion vectors only, never pixels — no in-loop filter, no IDCT, no B-frames
    // (they never reach check_frame, src/motion_scanner.cpp:154), luma-only fast paths, and one
    // decoding thread per scanner because parallelism is per chunk (src/pipeline.cpp:186-197).
    static const char *const kDecoderOptions[][2] = {
        {"flags2", "+export_mvs+fast"}, {"flags", "+gray"},       {"skip_frame", "bidir"},
        {"skip_idct", "all"},           {"skip_loop_filter", "all"}, {"threads", "1"},
        {"thread_type", "slice"},
    };
    AVDictionary *opts = nullptr;
    for (const auto &kv : kDecoderOptions) av_dict_set(&opts, kv[0], kv[1], 0);
    const int rc = avcodec_open2(dec_, codec, &opts);
    av_dict_free(&opts);       // entries the decoder did not consume are dropped with the dictionary
    if (rc < 0) fail("avcodec_open2");
  }
  ~LibavSource() override { close(); }
  LibavSource(const LibavSource &) = delete;
  LibavSource &operator=(const LibavSource &) = delete;

  int width() const override { return dec_->width; }
  int height() const override { return dec_->height; }
  double duration() const override {
    return fmt_->duration != AV_NOPTS_VALUE ? fmt_->duration / (double)AV_TIME_BASE : 0.0;
  }
  double fps() const override {
    const AVRational r = fmt_->streams[vs_]->avg_frame_rate;
    return r.den > 0 ? av_q2d(r) : 25.0;
  }
  double time_base() const override { return av_q2d(fmt_->streams[vs_]->time_base); }

  void seek(double seconds) override {
    const int64_t ts = static_cast<int64_t>(seconds / time_base());
    av_seek_frame(fmt_, vs_, ts, AVSEEK_FLAG_BACKWARD);
    avcodec_flush_buffers(dec_);
  }

  bool next(Frame &out) override {
    for (;;) {
      const int got = avcodec_receive_frame(dec_, frame_);
      if (got == 0) {
        const AVFrameSideData *sd = av_frame_get_side_data(frame_, AV_FRAME_DATA_MOTION_VECTORS);
        out.pts = frame_->pts;
        out.has_side_data = sd != nullptr;
        out.mv = sd ? sd->data : nullptr;
        out.mv_bytes = sd ? (size_t)sd->size : 0;
        return true;                            // bytes stay valid until the next call
      }
      // got < 0 — EAGAIN (the decoder wants input) or a decode error: either way the reference leaves its
      // receive loop and reads the next packet (src/motion_scanner.cpp:347-351, 334); only the end of the
      // file ends the range (the decoder is never drained there either)
      bool fed = false;
      while (!fed) {
        if (av_read_frame(fmt_, pkt_) < 0) {    // end of file: the reference stops here too
          return false;
        }
        if (pkt_->stream_index == vs_) fed = avcodec_send_packet(dec_, pkt_) >= 0;
        av_packet_unref(pkt_);
      }
    }
  }
};

}  // namespace mtgpu_host
#endif  // MTGPU_WITH_LIBAV
