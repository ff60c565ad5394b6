/*
 * mt_types.h — plain-C record types shared by the MI355X MV-scan library
 * (include/mtgpu.h) and the CPU oracle (oracle/mt_oracle.h).
 *
 * Every type here restates a record of the reference
 * (Vaibhav-20022002/Motion-Estimated-Video-Trimmer); the file:line citations
 * are relative to the reference tree.
 */
#ifndef MT_TYPES_H
#define MT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * mt_mv — one exported motion vector, byte-compatible with FFmpeg's
 * `AVMotionVector` (libavutil/motion_vector.h, FFmpeg 8.0; third-party, not in
 * the reference tree).  The reference reads it at src/motion_scanner.cpp:224-226
 * (cast of the side-data bytes) and 243-256 (fields src_x/src_y/dst_x/dst_y);
 * every field is enumerated at tools/extract_mvs.cpp:148-164.
 *
 * The scan reads ONLY src_x, src_y, dst_x, dst_y (bytes 6..13 of each 40).
 */
typedef struct mt_mv {
  int32_t source;        /*  0  <0: past reference, >0: future reference        */
  uint8_t w, h;          /*  4,5  block width/height                            */
  int16_t src_x, src_y;  /*  6,8  absolute source position                      */
  int16_t dst_x, dst_y;  /* 10,12 absolute destination position                 */
  uint16_t _pad0;        /* 14  (natural padding before the u64)                */
  uint64_t flags;        /* 16                                                   */
  int32_t motion_x;      /* 24                                                   */
  int32_t motion_y;      /* 28                                                   */
  uint16_t motion_scale; /* 32                                                   */
  uint16_t _pad1[3];     /* 34  tail padding to 40                              */
} mt_mv;

#define MT_MV_BYTES 40

/*
 * Compact record: the 8 bytes of an AVMotionVector that check_frame reads (src_x, src_y,
 * dst_x, dst_y = bytes 6..13; src/motion_scanner.cpp:246-256), packed back to back.  What the
 * host dispatcher stages and ships over PCIe instead of the 40-byte record (5x fewer bytes).
 */
typedef struct mt_mv_compact {
  int16_t src_x, src_y;
  int16_t dst_x, dst_y;
} mt_mv_compact;
#define MT_COMPACT_BYTES 8

#if defined(__cplusplus)
static_assert(sizeof(mt_mv) == MT_MV_BYTES, "mt_mv must match AVMotionVector (40 B)");
static_assert(offsetof(mt_mv, src_x) == 6 && offsetof(mt_mv, src_y) == 8, "mt_mv layout");
static_assert(offsetof(mt_mv, dst_x) == 10 && offsetof(mt_mv, dst_y) == 12, "mt_mv layout");
static_assert(offsetof(mt_mv, flags) == 16 && offsetof(mt_mv, motion_x) == 24, "mt_mv layout");
static_assert(offsetof(mt_mv, motion_scale) == 32, "mt_mv layout");
static_assert(sizeof(mt_mv_compact) == 8, "mt_mv_compact is bytes 6..13 of mt_mv");
#else
_Static_assert(sizeof(mt_mv_compact) == 8, "mt_mv_compact is bytes 6..13 of mt_mv");
_Static_assert(sizeof(mt_mv) == MT_MV_BYTES, "mt_mv must match AVMotionVector (40 B)");
_Static_assert(offsetof(mt_mv, src_x) == 6 && offsetof(mt_mv, dst_y) == 12, "mt_mv layout");
_Static_assert(offsetof(mt_mv, flags) == 16 && offsetof(mt_mv, motion_scale) == 32, "mt_mv layout");
#endif

/*
 * mt_scan_params — the per-scanner parameter block: `MotionScanner::cfg` plus
 * the grid dimensions (include/motion_trim/motion_scanner.hpp:76-77, 86-93),
 * derived in MotionScanner::initialize() (src/motion_scanner.cpp:184-199).
 * Build it with mtgpu_params_from_config() / mto_params_from_config().
 */
typedef struct mt_scan_params {
  double mv_threshold_sq;  /* cfg.mv_threshold_sq: keep an MV iff !(mag_sq < this)     */
  int32_t block_shift;     /* cfg.block_shift: gx = dst_x >> block_shift               */
  int32_t clusters_needed; /* cfg.clusters_needed                                      */
  int32_t vertical_margin; /* cfg.vertical_margin: rows masked at top and at bottom    */
  uint8_t vectors_needed;  /* cfg.vectors_needed (uint8: the int config wraps mod 256) */
  uint8_t _pad[3];
  int32_t grid_w;          /* grid_w (int16 in the reference: 1..32767)                */
  int32_t grid_h;          /* grid_h                                                   */
} mt_scan_params;

/*
 * mt_segment — `TimeSegment{double start, end}` (include/motion_trim/types.hpp:56-59),
 * the record the cut executor consumes (include/motion_trim/ffmpeg_queue.hpp:32-38).
 * Same size (16) and field offsets (0, 8) as the reference's struct, which is declared
 * alignas(16); mt_segment deliberately asks only for the natural 8-byte alignment, so every
 * TimeSegment array is a valid mt_segment array (INTEGRATION.md casts in that direction) and
 * segment lists may sit at any 8-byte offset of a packed buffer (mtgpu_gather_segments).
 * Checked against the compiled reference header in tests/test_reference_host.py.
 */
typedef struct mt_segment {
  double start;
  double end;
} mt_segment;

/*
 * mt_merge_params — the constants of the merge/cut decision
 * (src/pipeline.cpp:333, 337-338, 343-344, 351-358; defaults config.hpp:93, 99, 123).
 */
typedef struct mt_merge_params {
  double max_gap_sec;     /* MAX_GAP_SEC: a gap strictly greater than this splits */
  double padding_sec;     /* PADDING_SEC                                           */
  double duration;        /* stream duration used for the clamp and the savings    */
  double min_savings_pct; /* MIN_SAVINGS_PCT: cut iff saved_pct > this             */
} mt_merge_params;

/*
 * mt_merge_result — what ProcessingPipeline::run() leaves behind after the merge
 * (src/pipeline.cpp:349-358, 387-388; getters include/motion_trim/pipeline.hpp:132-142).
 */
typedef struct mt_merge_result {
  uint64_t n_timestamps;  /* motion timestamps after sort + unique (pipeline.cpp:302-304)   */
  uint64_t n_segments;    /* segments of the merge (0 iff no motion, pipeline.cpp:308-319)  */
  double time_removed;    /* duration - out_dur                (pipeline.cpp:355)           */
  double saved_pct;       /* time_removed / duration * 100     (pipeline.cpp:356)           */
  int32_t do_cut;         /* 1: saved_pct > min_savings_pct -> cut to the segments;          *
                           * 0: full copy {0,duration} (pipeline.cpp:358, 387-388);          *
                           * -1: no motion, run() returns before any job (308-319)           */
  int32_t status;         /* MT_OK, or MT_ERR_INVALID if a timestamp was NaN (device merges     *
                           * report per-stream problems here); every other field is then 0   *
                           * and do_cut -1, whatever the list length (std::sort of NaN is    *
                           * undefined in the reference: outside the defined domain)         */
} mt_merge_result;

/* Status codes shared by both libraries (0 = ok; the reference itself only
 * returns bool/int + a log line, src/motion_scanner.cpp:62-202). */
enum {
  MT_OK = 0,
  MT_ERR_INVALID = 1,   /* bad argument / parameter outside the defined domain */
  MT_ERR_CAPACITY = 2,  /* output buffer too small (n_out still reports need)  */
  MT_ERR_DEVICE = 3,    /* HIP runtime error (see *_last_error())              */
  MT_ERR_NOMEM = 4,
  MT_ERR_BUSY = 5,      /* every staging buffer of a pipe is in flight: collect first */
  MT_ERR_UNSUPPORTED = 6 /* the host CPU lacks the instruction set that was asked for      */
};

#ifdef __cplusplus
}
#endif
#endif /* MT_TYPES_H */
