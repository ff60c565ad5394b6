/* pipe_example.c — what a decoder thread does instead of calling check_frame per frame
 * (reference src/motion_scanner.cpp:375-383): copy each frame's MV side data into a pinned batch,
 * submit full batches asynchronously, collect the flags in submission order, merge the motion
 * timestamps.  Plain C against include/mtgpu.h.
 *
 *   gcc -std=c11 -Iinclude examples/pipe_example.c -o pipe_example \
 *       -Lmotion-estimated-video-trimmer_amd -lmtgpu -Wl,-rpath,$PWD/motion-estimated-video-trimmer_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mtgpu.h"

#define CHECK(call)                                                        \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ != MT_OK) {                                                    \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mtgpu_last_error());   \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static double g_ts[4096];
static size_t g_nts = 0;
static int g_inflight = 0;

/* wait for the oldest batch, keep the timestamps of its motion frames (:382-383), free the batch */
static int collect(mtgpu_pipe *pipe) {
  mtgpu_batch *b = NULL;
  const uint8_t *flags = NULL;
  const double *pts = NULL;
  uint32_t n = 0;
  CHECK(mtgpu_pipe_collect(pipe, &b, &flags, &pts, NULL, &n));
  for (uint32_t i = 0; i < n; ++i)
    if (flags[i]) g_ts[g_nts++] = pts[i];
  --g_inflight;
  CHECK(mtgpu_pipe_release(pipe, b));
  return 0;
}

int main(void) {
  mt_scan_params p;
  CHECK(mtgpu_params_from_config(&p, 1920, 1080, 16.0, 16, 4, 2, 2, 0.05f));
  mtgpu_ctx *ctx = NULL;
  CHECK(mtgpu_create(&p, 0, &ctx));
  /* 3 pinned batches of up to 16 frames / 4096 records; default staging: 8-byte compact records
   * that the scan kernel reads over PCIe itself (no copy commands) */
  mtgpu_pipe *pipe = NULL;
  CHECK(mtgpu_pipe_create(ctx, 4096, 16, 3, &pipe));

  enum { F = 300, PER = 4 };                       /* 10 s at 30 fps; motion in seconds 2-3 and 8-9 */
  mtgpu_batch *cur = NULL;
  for (int f = 0; f < F; ++f) {
    mt_mv side_data[PER];                          /* stands for the AVFrame's side data: dies with the frame (:347) */
    size_t n = 0;
    const int moving = (f >= 60 && f < 90) || (f >= 240 && f < 270);
    memset(side_data, 0, sizeof side_data);
    if (moving)
      for (int k = 0; k < PER; ++k) {
        mt_mv *v = &side_data[n++];
        v->dst_x = (int16_t)(16 * (40 + k / 2) + 8);
        v->dst_y = (int16_t)(16 * 30 + 8);
        v->src_x = (int16_t)(v->dst_x - 6);
        v->src_y = v->dst_y;
      }
    const int is_keyframe = (f % 30) == 0;         /* I-frames export no MV side data (:219-221) */
    for (;;) {
      if (!cur) {
        int rc = mtgpu_pipe_acquire(pipe, &cur);
        if (rc == MT_ERR_BUSY) { if (collect(pipe)) return 1; continue; }   /* back-pressure */
        CHECK(rc);
      }
      int rc = mtgpu_batch_add_frame(cur, is_keyframe ? NULL : side_data, n * sizeof(mt_mv), !is_keyframe, f / 30.0,
                                     (uint64_t)f);
      if (rc == MT_ERR_CAPACITY) {                 /* batch full: ship it, start the next one */
        CHECK(mtgpu_pipe_submit(pipe, cur));
        cur = NULL;
        ++g_inflight;
        continue;
      }
      CHECK(rc);
      break;
    }
  }
  if (cur) { CHECK(mtgpu_pipe_submit(pipe, cur)); ++g_inflight; }
  while (g_inflight > 0)
    if (collect(pipe)) return 1;

  mt_merge_params mp = {5.0, 0.5, F / 30.0, 5.0};   /* MAX_GAP_SEC, PADDING_SEC, duration, MIN_SAVINGS_PCT */
  mt_segment seg[8];
  mt_merge_result r;
  CHECK(mtgpu_merge_segments(ctx, g_ts, g_nts, &mp, 1, seg, 8, &r));
  printf("motion frames %zu, segments %llu, do_cut %d, saved %.1f%%\n", g_nts, (unsigned long long)r.n_segments,
         r.do_cut, r.saved_pct);
  for (uint64_t i = 0; i < r.n_segments; ++i) printf("  [%.3f, %.3f]\n", seg[i].start, seg[i].end);
  mtgpu_pipe_destroy(pipe);
  mtgpu_destroy(ctx);
  /* frames 60 and 240 are keyframes: 29 + 29 motion frames; two bursts 6 s apart -> two segments */
  return (g_nts == 58 && r.n_segments == 2 && r.do_cut == 1) ? 0 : 3;
}
