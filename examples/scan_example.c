/* scan_example.c — the C ABI from plain C: build a tiny batch of AVMotionVector records,
 * scan it on the GPU, merge the motion timestamps into segments.
 *
 *   gcc -std=c11 -Iinclude examples/scan_example.c -o scan_example \
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

int main(void) {
  /* 1080p, reference code defaults (MV_THRESHOLD_SQ 16, BLOCK 16/4, VECTORS 2, CLUSTERS 2, MASK 0.05) */
  mt_scan_params p;
  CHECK(mtgpu_params_from_config(&p, 1920, 1080, 16.0, 16, 4, 2, 2, 0.05f));
  mtgpu_ctx *ctx = NULL;
  CHECK(mtgpu_create(&p, 0, &ctx));

  /* 90 frames at 30 fps; frames 30..59 carry a 2x1-cell object moving 6 px/frame */
  enum { F = 90, PER = 4 };
  mt_mv *mv = calloc((size_t)F * PER, sizeof *mv);
  uint64_t off[F + 1];
  double pts[F];
  size_t n = 0;
  off[0] = 0;
  for (int f = 0; f < F; ++f) {
    pts[f] = f / 30.0;
    if (f >= 30 && f < 60)
      for (int k = 0; k < PER; ++k) {            /* two votes in each of two adjacent cells */
        mt_mv *v = &mv[n++];
        v->dst_x = (int16_t)(16 * (40 + k / 2) + 8);
        v->dst_y = (int16_t)(16 * 30 + 8);
        v->src_x = (int16_t)(v->dst_x - 6);
        v->src_y = v->dst_y;
        v->w = v->h = 8;
        v->source = -1;
      }
    off[f + 1] = n;
  }
  uint8_t has_sd[F], flags[F];
  memset(has_sd, 1, sizeof has_sd);              /* every frame had MV side data (some with 0 records) */
  CHECK(mtgpu_scan_frames(ctx, mv, off, has_sd, F, flags));

  double ts[F];
  size_t m = 0;
  for (int f = 0; f < F; ++f)
    if (flags[f]) ts[m++] = pts[f];

  mt_merge_params mp = {5.0, 0.5, F / 30.0, 5.0};   /* MAX_GAP_SEC, PADDING_SEC, duration, MIN_SAVINGS_PCT */
  mt_segment seg[8];
  mt_merge_result r;
  CHECK(mtgpu_merge_segments(ctx, ts, m, &mp, 1, seg, 8, &r));
  printf("motion frames %zu, segments %llu, do_cut %d, saved %.1f%%\n", m,
         (unsigned long long)r.n_segments, r.do_cut, r.saved_pct);
  for (uint64_t i = 0; i < r.n_segments; ++i) printf("  [%.3f, %.3f]\n", seg[i].start, seg[i].end);
  mtgpu_destroy(ctx);
  free(mv);
  return (m == 30 && r.n_segments == 1 && r.do_cut == 1) ? 0 : 3;
}
