/*
 * mt_oracle.h — CPU ORACLE for the MV-scan hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's algorithm for the path named
 * by BASELINE.json `north_star`: MotionScanner::check_frame, the
 * cfg/grid derivation, the scan_range frame filter, the chunking, sort+unique and
 * the gap-bounded segment merge.  Each function cites the reference file:line it
 * follows (paths relative to the reference tree).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * product (include/mtgpu.h, libmtgpu.so) never links, loads or calls it.
 *
 * PARITY UNPINNED by the reference's own tests: the reference ships no tests,
 * golden vectors or fixtures, and its translation units for this
 * path include <libavcodec/avcodec.h>, which this image lacks, so the reference
 * object code cannot be built here without writing stand-in headers (not done).
 * What pins this file instead: hand-derived known-answer vectors in
 * tests/golden/ (each derivable by reading the cited reference lines), the
 * behaviour the reference documents in the comments of its config/motion_trim.env
 * (tests/golden/reference_documented_examples.json), and the two segment values
 * SURVEY.md §8c recorded from a run of the reference's own object code during
 * the survey (tests/golden/survey_segments.json).
 */
#ifndef MT_ORACLE_H
#define MT_ORACLE_H

#include "../include/mt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* src/motion_scanner.cpp:184-199 + include/motion_trim/config.hpp:56-89.
 * `vectors_needed` is the raw int config value (cast to uint8 like config.hpp:75).
 * Returns MT_ERR_INVALID outside the domain where the reference is defined
 * (block_shift outside [0,31], non-positive grid, grid dims > INT16_MAX). */
int mto_params_from_config(mt_scan_params *out, int width, int height,
                           double mv_threshold_sq, int block_size, int block_shift,
                           int vectors_needed, int clusters_needed, float vertical_mask);

/* src/motion_scanner.cpp:217-295.  `grid` is caller scratch of grid_w*grid_h bytes.
 * has_side_data == 0 restates the `if (!sd) return false` of lines 219-221. */
int mto_check_frame(const mt_scan_params *p, const mt_mv *mvs, int64_t count,
                    int has_side_data, uint8_t *grid);

/* The same, but also reports the number of cluster-centre cells over the whole
 * frame (no early exit) — used to cross-check the early-exit equivalence
 * result == (centres >= max(1, clusters_needed)). */
int mto_check_frame_count(const mt_scan_params *p, const mt_mv *mvs, int64_t count,
                          int has_side_data, uint8_t *grid, int64_t *centres);

/* A batch of frames in the product's CSR layout: frame f owns records
 * [frame_off[f], frame_off[f+1]).  has_sd may be NULL: then a frame has side data
 * iff it owns at least one record.  flags[f] = check_frame(f) as 0/1. */
int mto_scan_frames(const mt_scan_params *p, const mt_mv *mv, const uint64_t *frame_off,
                    const uint8_t *has_sd, uint32_t n_frames, uint8_t *flags);

/* Same, frames split statically over `nthreads` pthreads, one private grid per
 * thread (the reference's model: one MotionScanner per worker, src/pipeline.cpp:186-197). */
int mto_scan_frames_mt(const mt_scan_params *p, const mt_mv *mv, const uint64_t *frame_off,
                       const uint8_t *has_sd, uint32_t n_frames, uint8_t *flags, int nthreads);

/* bench.py's cpu_baseline leg only: every thread copies its share of the frames into memory it allocates itself
 * (NUMA-local, like a worker's own decoder output), then all threads scan their shares `reps` times between two
 * barriers; *seconds = that wall time.  flags = results of the last pass. */
int mto_bench_scan(const mt_scan_params *p, const mt_mv *mv, const uint64_t *frame_off, const uint8_t *has_sd,
                   uint32_t n_frames, uint8_t *flags, int nthreads, int reps, double *seconds);

/* src/motion_scanner.cpp:307-313. */
int mto_frame_skip(double video_fps, double target_fps);

/* src/motion_scanner.cpp:314, 357-371: which decoded frames of one scan_range(start,end)
 * call reach check_frame.  frame_pts[i] are the AVFrame::pts of the frames the decoder
 * returns after the seek, in decode order; writes analysed[i] = 1/0 and pts_sec[i]
 * (= pts * time_base, line 361) and returns how many leading frames were consumed
 * before the `pts >= end` return (n if the stream ran out first). */
int64_t mto_filter_frames(const int64_t *frame_pts, int64_t n, double time_base,
                          double start, double end, int frame_skip,
                          uint8_t *analysed, double *pts_sec);

/* src/pipeline.cpp:141-142, 163-167: chunk boundaries.  Writes up to cap {start,end}
 * pairs, returns the number of chunks the loop creates. */
int64_t mto_chunks(double duration, double chunk_sec, mt_segment *out, int64_t cap);

/* src/pipeline.cpp:302-304: std::sort + std::unique in place; returns the new length,
 * or -1 if a NaN is present (outside the defined domain). */
int64_t mto_sort_unique(double *ts, int64_t n);

/* src/pipeline.cpp:308-358, 387-388 on sorted, de-duplicated timestamps.
 * job_semantics == 0: `out` receives the merged, clamped segments (pipeline.cpp:328-354).
 * job_semantics == 1: `out` receives what the FFmpegJob would carry: those segments if
 *   res->do_cut == 1, the single segment {0,duration} if 0, nothing if -1.
 * res->n_segments is always the number written (or needed, with MT_ERR_CAPACITY). */
int mto_merge_segments(const double *ts, int64_t n, const mt_merge_params *mp,
                       int job_semantics, mt_segment *out, int64_t cap,
                       mt_merge_result *res);

/* tools/motion_scalar.cpp:61-84 — the per-second "motion scalar" of BASELINE.json
 * config 0 (plumbing only, not the hot path): sec = floor(pts); for every MV with
 * motion_scale != 0: acc[sec] += sqrt(dx*dx+dy*dy) * w * h, dx = motion_x/scale.
 * pts_sec < 0 stands for the JSON null (tools/extract_mvs.cpp:137-140).
 * acc has n_sec entries (seconds 0..n_sec-1); frames outside are ignored. */
int mto_motion_scalar(const mt_mv *mv, const uint64_t *frame_off, const double *pts_sec,
                      uint32_t n_frames, double *acc, int64_t n_sec);

const char *mto_version(void);

#ifdef __cplusplus
}
#endif
#endif
