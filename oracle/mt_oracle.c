#define _GNU_SOURCE
/*
 * mt_oracle.c — CPU ORACLE for the MV-scan hot path.  TEST INFRASTRUCTURE ONLY
 * (see mt_oracle.h: who may load it, and why parity is "unpinned").
 *
 * Scalar, single-pass restatement of the reference algorithm; citations are
 * file:line in the reference tree.  Build: `make -C oracle` (gcc -O3,
 * -ffp-contract=off so the double arithmetic of the merge is never fused).
 *
 * Defined domain (where the reference itself has undefined behaviour we pick a
 * definition, state it here, and keep golden vectors away from it):
 *  - mag_sq is computed exactly in 64-bit.  The reference computes it in `int`
 *    (src/motion_scanner.cpp:246-248), which overflows when |dx| or |dy| > 32767;
 *    for |dx|,|dy| <= 32767 both agree.
 *  - neighbours outside the grid (only reachable when vertical_margin == 0) are
 *    inactive.  The reference reads out of bounds there (motion_scanner.cpp:285-286).
 *  - timestamps are not NaN.
 */
#include "mt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

const char *mto_version(void) { return "mt_oracle 1 (CPU restatement, test-only)"; }

/* ------------------------------------------------------------------ a2 --- */

int mto_params_from_config(mt_scan_params *out, int width, int height,
                           double mv_threshold_sq, int block_size, int block_shift,
                           int vectors_needed, int clusters_needed, float vertical_mask) {
  if (!out) return MT_ERR_INVALID;
  if (block_shift < 0 || block_shift > 31) return MT_ERR_INVALID;
  /* motion_scanner.cpp:190-193: (dim + BLOCK_SIZE - 1) >> block_shift, cast to int16. */
  long gw_l = ((long)width + block_size - 1) >> block_shift;
  long gh_l = ((long)height + block_size - 1) >> block_shift;
  if (gw_l < 1 || gh_l < 1 || gw_l > 32767 || gh_l > 32767) return MT_ERR_INVALID;
  int16_t gw = (int16_t)gw_l, gh = (int16_t)gh_l;

  memset(out, 0, sizeof *out);
  out->mv_threshold_sq = mv_threshold_sq;                /* :184 */
  out->block_shift = block_shift;                        /* :185 */
  out->vectors_needed = (uint8_t)vectors_needed;         /* :186 + config.hpp:75 */
  out->clusters_needed = clusters_needed;                /* :187 */
  out->grid_w = gw;
  out->grid_h = gh;
  /* :196 — int16 * float is evaluated in float32, then truncated toward zero. */
  float m = (float)gh * vertical_mask;
  out->vertical_margin = (int)m;
  return MT_OK;
}

/* ------------------------------------------------------------ a3..a5 --- */

static int params_ok(const mt_scan_params *p) {
  /* vertical_margin < 0 (a negative VERTICAL_MASK) makes the reference index before the grid
   * (:237, :285): outside the defined domain, rejected like the product does (validate_params). */
  return p && p->grid_w >= 1 && p->grid_h >= 1 && p->grid_w <= 32767 && p->grid_h <= 32767 &&
         p->block_shift >= 0 && p->block_shift <= 31 && p->vertical_margin >= 0;
}

/* Phase 1, motion_scanner.cpp:229-268. */
static void vote(const mt_scan_params *p, const mt_mv *mvs, int64_t count, uint8_t *grid) {
  const int gw = p->grid_w, gh = p->grid_h;
  const int y_lo = p->vertical_margin;      /* :237 */
  const int y_hi = gh - p->vertical_margin; /* :238 */
  const double thr = p->mv_threshold_sq;
  const int sh = p->block_shift;

  memset(grid, 0, (size_t)gw * (size_t)gh); /* :229 */
  for (int64_t i = 0; i < count; ++i) {
    const mt_mv *v = mvs + i;
    int64_t dx = (int64_t)v->dst_x - v->src_x; /* :246 */
    int64_t dy = (int64_t)v->dst_y - v->src_y; /* :247 */
    int64_t mag = dx * dx + dy * dy;           /* :248 (exact, see header note) */
    if ((double)mag < thr) continue;           /* :251 — NaN thr keeps everything */
    int gx = (int)v->dst_x >> sh;              /* :255 arithmetic shift of the promoted int16 */
    int gy = (int)v->dst_y >> sh;              /* :256 */
    if (gx < 0 || gx >= gw || gy < y_lo || gy >= y_hi) continue; /* :262 */
    uint8_t *c = grid + (size_t)gy * gw + gx;
    if (*c != 255) ++*c;                       /* :265-266 saturating */
  }
}

static inline int cell_on(const mt_scan_params *p, const uint8_t *grid, int x, int y) {
  if (x < 0 || y < 0 || x >= p->grid_w || y >= p->grid_h) return 0; /* defined: outside = inactive */
  return grid[(size_t)y * p->grid_w + x] >= p->vectors_needed;     /* :282, 285-286 */
}

/* Phase 2, motion_scanner.cpp:272-294.  early_exit reproduces the `return true`
 * inside the loop; without it the full centre count is returned in *centres. */
static int clusters(const mt_scan_params *p, const uint8_t *grid, int early_exit,
                    int64_t *centres) {
  const int gw = p->grid_w;
  const int y_lo = p->vertical_margin, y_hi = p->grid_h - p->vertical_margin;
  int64_t n = 0;
  for (int y = y_lo; y < y_hi; ++y) {
    for (int x = 1; x < gw - 1; ++x) {     /* :280 */
      if (!cell_on(p, grid, x, y)) continue;
      int nb = cell_on(p, grid, x - 1, y) | cell_on(p, grid, x + 1, y) |
               cell_on(p, grid, x, y - 1) | cell_on(p, grid, x, y + 1);
      if (!nb) continue;
      ++n;                                  /* :288 ++clusters >= clust_need -> true */
      if (early_exit && n >= p->clusters_needed) {
        if (centres) *centres = n;
        return 1;
      }
    }
  }
  if (centres) *centres = n;
  if (early_exit) return 0;                 /* :294 */
  int64_t need = p->clusters_needed < 1 ? 1 : p->clusters_needed;
  return n >= need;
}

int mto_check_frame(const mt_scan_params *p, const mt_mv *mvs, int64_t count,
                    int has_side_data, uint8_t *grid) {
  if (!params_ok(p) || !grid || count < 0) return -MT_ERR_INVALID;
  if (!has_side_data) return 0;             /* :219-221 */
  vote(p, mvs, count, grid);
  return clusters(p, grid, 1, NULL);
}

int mto_check_frame_count(const mt_scan_params *p, const mt_mv *mvs, int64_t count,
                          int has_side_data, uint8_t *grid, int64_t *centres) {
  if (!params_ok(p) || !grid || count < 0) return -MT_ERR_INVALID;
  if (centres) *centres = 0;
  if (!has_side_data) return 0;
  vote(p, mvs, count, grid);
  return clusters(p, grid, 0, centres);
}

static int scan_range_frames(const mt_scan_params *p, const mt_mv *mv, const uint64_t *frame_off,
                             const uint8_t *has_sd, uint32_t f0, uint32_t f1, uint8_t *flags,
                             uint8_t *grid) {
  for (uint32_t f = f0; f < f1; ++f) {
    uint64_t a = frame_off[f], b = frame_off[f + 1];
    if (b < a) return MT_ERR_INVALID;
    int sd = has_sd ? (has_sd[f] != 0) : (b > a);
    int r = mto_check_frame(p, mv + a, (int64_t)(b - a), sd, grid);
    if (r < 0) return -r;
    flags[f] = (uint8_t)r;
  }
  return MT_OK;
}

int mto_scan_frames(const mt_scan_params *p, const mt_mv *mv, const uint64_t *frame_off,
                    const uint8_t *has_sd, uint32_t n_frames, uint8_t *flags) {
  if (!params_ok(p)) return MT_ERR_INVALID;
  if (n_frames == 0) return MT_OK;
  if (!frame_off || !flags) return MT_ERR_INVALID;
  uint8_t *grid = (uint8_t *)malloc((size_t)p->grid_w * p->grid_h);
  if (!grid) return MT_ERR_NOMEM;
  int rc = scan_range_frames(p, mv, frame_off, has_sd, 0, n_frames, flags, grid);
  free(grid);
  return rc;
}

typedef struct {
  const mt_scan_params *p;
  const mt_mv *mv;
  const uint64_t *frame_off;
  const uint8_t *has_sd;
  uint8_t *flags;
  uint32_t f0, f1;
  int rc;
} mt_job;

static void *mt_worker(void *arg) {
  mt_job *j = (mt_job *)arg;
  uint8_t *grid = (uint8_t *)malloc((size_t)j->p->grid_w * j->p->grid_h);
  if (!grid) { j->rc = MT_ERR_NOMEM; return NULL; }
  j->rc = scan_range_frames(j->p, j->mv, j->frame_off, j->has_sd, j->f0, j->f1, j->flags, grid);
  free(grid);
  return NULL;
}

int mto_scan_frames_mt(const mt_scan_params *p, const mt_mv *mv, const uint64_t *frame_off,
                       const uint8_t *has_sd, uint32_t n_frames, uint8_t *flags, int nthreads) {
  if (nthreads <= 1 || n_frames < 2) return mto_scan_frames(p, mv, frame_off, has_sd, n_frames, flags);
  if (!params_ok(p) || !frame_off || !flags) return MT_ERR_INVALID;
  if ((uint32_t)nthreads > n_frames) nthreads = (int)n_frames;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
  mt_job *jobs = (mt_job *)malloc(sizeof(mt_job) * nthreads);
  if (!th || !jobs) { free(th); free(jobs); return MT_ERR_NOMEM; }
  /* Static split balanced by RECORDS, so every thread streams about the same bytes. */
  uint64_t total = frame_off[n_frames] - frame_off[0];
  uint32_t f = 0;
  for (int t = 0; t < nthreads; ++t) {
    uint64_t target = frame_off[0] + (total * (uint64_t)(t + 1)) / (uint64_t)nthreads;
    uint32_t e = f;
    if (t == nthreads - 1) e = n_frames;
    else while (e < n_frames && frame_off[e + 1] <= target) ++e;
    if (e == f && f < n_frames && (n_frames - f) > (uint32_t)(nthreads - 1 - t)) e = f + 1;
    jobs[t] = (mt_job){p, mv, frame_off, has_sd, flags, f, e, MT_OK};
    f = e;
  }
  for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, mt_worker, &jobs[t]);
  int rc = MT_OK;
  for (int t = 0; t < nthreads; ++t) {
    pthread_join(th[t], NULL);
    if (jobs[t].rc != MT_OK) rc = jobs[t].rc;
  }
  free(th);
  free(jobs);
  return rc;
}

/* Timed CPU baseline (bench.py's cpu_baseline leg only).  The reference's model is one scanner per worker
 * thread, each analysing frames its own decoder has just written into its own memory (src/pipeline.cpp:186-197,
 * timer bracket src/motion_scanner.cpp:375-380).  So each of `nthreads` pthreads first COPIES its share of the
 * frames into a buffer it allocates itself (first touch: its own NUMA node), all threads meet at a barrier, every
 * thread scans its share `reps` times, and the wall time between the two barriers comes back in *seconds (thread
 * creation, the copies and the join are outside it).  flags = the results of the last pass. */
typedef struct {
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int go;                        /* 0: wait, 1: run (the barrier is initialised), -1: give up */
} mt_gate;

typedef struct {
  mt_job j;
  int reps;
  mt_gate *gate;
  pthread_barrier_t *bar;
  double *t0, *t1;
} mt_bench_job;

static double now_sec(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *mt_bench_worker(void *arg) {
  mt_bench_job *b = (mt_bench_job *)arg;
  mt_job *j = &b->j;
  /* nobody touches the barrier before the main thread knows that every party exists */
  pthread_mutex_lock(&b->gate->mu);
  while (b->gate->go == 0) pthread_cond_wait(&b->gate->cv, &b->gate->mu);
  const int go = b->gate->go;
  pthread_mutex_unlock(&b->gate->mu);
  if (go < 0) { j->rc = MT_ERR_NOMEM; return NULL; }
  const uint32_t nf = j->f1 - j->f0;
  const uint64_t r0 = j->frame_off[j->f0], r1 = j->frame_off[j->f1];
  uint8_t *grid = (uint8_t *)malloc((size_t)j->p->grid_w * j->p->grid_h);
  mt_mv *lmv = (mt_mv *)malloc((size_t)(r1 - r0) * sizeof(mt_mv) + 64);
  uint64_t *loff = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)nf + 1));
  uint8_t *lsd = (j->has_sd && nf) ? (uint8_t *)malloc(nf) : NULL;
  uint8_t *lfl = (uint8_t *)malloc((size_t)nf + 1);
  const int ok = grid && lmv && loff && lfl && (!j->has_sd || !nf || lsd);
  if (ok) {
    memcpy(lmv, j->mv + r0, (size_t)(r1 - r0) * sizeof(mt_mv));
    for (uint32_t f = 0; f <= nf; ++f) loff[f] = j->frame_off[j->f0 + f] - r0;
    if (lsd) memcpy(lsd, j->has_sd + j->f0, nf);
  }
  j->rc = ok ? MT_OK : MT_ERR_NOMEM;
  if (pthread_barrier_wait(b->bar) == PTHREAD_BARRIER_SERIAL_THREAD) *b->t0 = now_sec();
  pthread_barrier_wait(b->bar);                      /* nobody starts before t0 is taken */
  if (ok)
    for (int r = 0; r < b->reps && j->rc == MT_OK; ++r)
      j->rc = scan_range_frames(j->p, lmv, loff, lsd, 0, nf, lfl, grid);
  if (pthread_barrier_wait(b->bar) == PTHREAD_BARRIER_SERIAL_THREAD) *b->t1 = now_sec();
  if (ok && j->rc == MT_OK) memcpy(j->flags + j->f0, lfl, nf);
  free(grid); free(lmv); free(loff); free(lsd); free(lfl);
  return NULL;
}

int mto_bench_scan(const mt_scan_params *p, const mt_mv *mv, const uint64_t *frame_off, const uint8_t *has_sd,
                   uint32_t n_frames, uint8_t *flags, int nthreads, int reps, double *seconds) {
  if (!params_ok(p) || !frame_off || !flags || !seconds || n_frames == 0 || reps < 1) return MT_ERR_INVALID;
  if (nthreads < 1) nthreads = 1;
  if ((uint32_t)nthreads > n_frames) nthreads = (int)n_frames;
  pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
  mt_bench_job *jobs = (mt_bench_job *)malloc(sizeof(mt_bench_job) * nthreads);
  if (!th || !jobs) { free(th); free(jobs); return MT_ERR_NOMEM; }
  mt_gate gate;
  pthread_mutex_init(&gate.mu, NULL);
  pthread_cond_init(&gate.cv, NULL);
  gate.go = 0;
  pthread_barrier_t bar;
  double t0 = 0.0, t1 = 0.0;
  const uint64_t total = frame_off[n_frames] - frame_off[0];
  uint32_t f = 0;
  for (int t = 0; t < nthreads; ++t) {               /* same record-balanced static split as mto_scan_frames_mt */
    uint64_t target = frame_off[0] + (total * (uint64_t)(t + 1)) / (uint64_t)nthreads;
    uint32_t e = f;
    if (t == nthreads - 1) e = n_frames;
    else while (e < n_frames && frame_off[e + 1] <= target) ++e;
    if (e == f && f < n_frames && (n_frames - f) > (uint32_t)(nthreads - 1 - t)) e = f + 1;
    jobs[t].j = (mt_job){p, mv, frame_off, has_sd, flags, f, e, MT_OK};
    jobs[t].reps = reps; jobs[t].gate = &gate; jobs[t].bar = &bar; jobs[t].t0 = &t0; jobs[t].t1 = &t1;
    f = e;
  }
  int started = 0, rc = MT_OK;
  for (; started < nthreads; ++started)
    if (pthread_create(&th[started], NULL, mt_bench_worker, &jobs[started]) != 0) break;
  /* every party exists: open the gate; a thread short (its frames would stay unscanned): everybody goes home */
  const int all = started == nthreads && pthread_barrier_init(&bar, NULL, (unsigned)nthreads) == 0;
  pthread_mutex_lock(&gate.mu);
  gate.go = all ? 1 : -1;
  pthread_cond_broadcast(&gate.cv);
  pthread_mutex_unlock(&gate.mu);
  for (int t = 0; t < started; ++t) {
    pthread_join(th[t], NULL);
    if (jobs[t].j.rc != MT_OK) rc = jobs[t].j.rc;
  }
  if (all) pthread_barrier_destroy(&bar);
  else rc = MT_ERR_NOMEM;
  pthread_mutex_destroy(&gate.mu);
  pthread_cond_destroy(&gate.cv);
  free(th);
  free(jobs);
  *seconds = t1 - t0;
  return rc;
}

/* ------------------------------------------------------------ a6, a7 --- */

int mto_frame_skip(double video_fps, double target_fps) {
  /* motion_scanner.cpp:311-313 */
  return (target_fps > 0 && target_fps < video_fps) ? (int)(video_fps / target_fps) : 1;
}

int64_t mto_filter_frames(const int64_t *frame_pts, int64_t n, double time_base,
                          double start, double end, int frame_skip,
                          uint8_t *analysed, double *pts_sec) {
  int frame_count = 0;                               /* :314 restarts per call */
  for (int64_t i = 0; i < n; ++i) {
    if (analysed) analysed[i] = 0;
    double pts = (double)frame_pts[i] * time_base;   /* :361 */
    if (pts_sec) pts_sec[i] = pts;
    if (++frame_count % frame_skip != 0) continue;   /* :357 */
    if (pts < start) continue;                       /* :364-365 */
    if (pts >= end) return i;                        /* :368-371 returns before analysing */
    if (analysed) analysed[i] = 1;                   /* :376 */
  }
  return n;
}

int64_t mto_chunks(double duration, double chunk_sec, mt_segment *out, int64_t cap) {
  int64_t k = 0;
  for (double t = 0; t < duration; t += chunk_sec) { /* pipeline.cpp:164 */
    double e = t + chunk_sec;
    if (duration < e) e = duration;                  /* :165 std::min */
    if (out && k < cap) { out[k].start = t; out[k].end = e; }
    ++k;
    if (!(chunk_sec > 0)) break;                     /* the reference would spin forever */
  }
  return k;
}

/* ------------------------------------------------------------ a8, a9 --- */

static int cmp_double(const void *a, const void *b) {
  double x = *(const double *)a, y = *(const double *)b;
  return (x > y) - (x < y);
}

int64_t mto_sort_unique(double *ts, int64_t n) {
  if (n <= 0) return 0;
  for (int64_t i = 0; i < n; ++i) if (ts[i] != ts[i]) return -1;
  qsort(ts, (size_t)n, sizeof(double), cmp_double);  /* pipeline.cpp:302 */
  int64_t m = 1;
  for (int64_t i = 1; i < n; ++i)                    /* :303-304 std::unique: operator== */
    if (!(ts[i] == ts[m - 1])) ts[m++] = ts[i];
  return m;
}

int mto_merge_segments(const double *ts, int64_t n, const mt_merge_params *mp,
                       int job_semantics, mt_segment *out, int64_t cap,
                       mt_merge_result *res) {
  if (!mp || !res || n < 0 || (n > 0 && !ts)) return MT_ERR_INVALID;
  memset(res, 0, sizeof *res);
  res->n_timestamps = (uint64_t)n;
  if (n == 0) {                                      /* pipeline.cpp:308-319 */
    res->do_cut = -1;
    return MT_OK;
  }
  const double gap = mp->max_gap_sec, pad = mp->padding_sec, dur = mp->duration;
  int64_t k = 0;
  double out_dur = 0;                                /* :349 */
  double seg_start = ts[0], last = ts[0];            /* :328-329 */
  for (int64_t i = 1; i <= n; ++i) {
    int close_seg = (i == n);
    if (!close_seg) {
      if (ts[i] != ts[i] || !(ts[i] > ts[i - 1])) return MT_ERR_INVALID; /* must be sorted+unique */
      double g = ts[i] - last;                       /* :332 */
      close_seg = g > gap;                           /* :333 strict */
    }
    if (close_seg) {
      double s = seg_start - pad;                    /* :337 / :343 */
      s = (0.0 < s) ? s : 0.0;                       /* std::max(0.0, s) == (a < b) ? b : a */
      double e = last + pad;                         /* :338 / :344 */
      if (dur < e) e = dur;                          /* :351 std::min(s.end, duration) */
      if (e < s) s = e;                              /* :352 std::min(s.start, s.end) */
      out_dur += (e - s);                            /* :353, in segment order */
      if (out && k < cap) { out[k].start = s; out[k].end = e; }
      ++k;
      if (i < n) seg_start = ts[i];                  /* :339 */
    }
    if (i < n) last = ts[i];                         /* :341 */
  }
  res->time_removed = dur - out_dur;                 /* :355 */
  res->saved_pct = (dur > 0) ? res->time_removed / dur * 100.0 : 0.0; /* :356 */
  res->do_cut = (res->saved_pct > mp->min_savings_pct) ? 1 : 0;       /* :358 */
  if (job_semantics && !res->do_cut) {               /* :387-388 full copy */
    if (out && cap >= 1) { out[0].start = 0.0; out[0].end = dur; }
    k = 1;
  }
  res->n_segments = (uint64_t)k;
  return (out == NULL || k <= cap) ? MT_OK : MT_ERR_CAPACITY;
}

/* ------------------------------------------------- config 0 plumbing --- */

int mto_motion_scalar(const mt_mv *mv, const uint64_t *frame_off, const double *pts_sec,
                      uint32_t n_frames, double *acc, int64_t n_sec) {
  if (!frame_off || !pts_sec || !acc || n_sec < 0) return MT_ERR_INVALID;
  for (int64_t s = 0; s < n_sec; ++s) acc[s] = 0.0;
  for (uint32_t f = 0; f < n_frames; ++f) {
    if (pts_sec[f] < 0) continue;                    /* JSON null, motion_scalar.cpp:62-63 */
    int sec = (int)floor(pts_sec[f]);                /* :66 */
    if (sec < 0 || sec >= n_sec) continue;
    for (uint64_t i = frame_off[f]; i < frame_off[f + 1]; ++i) {
      const mt_mv *v = mv + i;
      int scale = v->motion_scale;
      if (scale == 0) continue;                      /* :75-76 */
      double dx = (double)v->motion_x / scale;       /* :78-79 */
      double dy = (double)v->motion_y / scale;
      double mag = sqrt(dx * dx + dy * dy);          /* :81 */
      acc[sec] += mag * v->w * v->h;                 /* :82 */
    }
  }
  return MT_OK;
}
