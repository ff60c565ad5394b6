// mtgpu_api.hip — implementation of the C ABI declared in include/mtgpu.h:
// parameter derivation, launch planning, host<->device staging.  The compute is
// in scan_kernels.hip / merge_kernels.hip; nothing here computes a result on
// the CPU (no fallback: without a usable device every compute call fails).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cctype>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/mtgpu.h"
#include "api_internal.h"
#include "knobs.h"
#include "pack_simd.h"
#include "merge_kernels.h"
#include "scan_kernels.h"

namespace {
thread_local char g_err[512] = "";
}

namespace mtgpu {

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(hipError_t e, const char *what) {
  return fail(MT_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace mtgpu

namespace {

using mtgpu::fail;
using mtgpu::hip_fail;

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return hip_fail(_e, #expr);   \
  } while (0)

// Grow-only device buffer used by the host-pointer entry points.
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  int reserve(size_t bytes) {
    if (bytes <= cap) return MT_OK;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) { p = nullptr; return hip_fail(e, "hipMalloc(staging)"); }
    cap = want;
    return MT_OK;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct mtgpu_ctx {
  mt_scan_params params;
  mtgpu::ScanK k;
  mtgpu_plan plan;
  int device;            // physical HIP device
  int logical_device = 0;  // what the caller asked for (differs only under MTGPU_ALIAS_DEVICES)
  hipStream_t stream;    // private stream of the host-pointer entry points
  int variant = 0;       // MTGPU_VARIANT experiment knob
  std::atomic<int> slices_request{0};  // 0 = auto, else 1/2/4/8 (mtgpu_set_slices, MTGPU_FORCE_SLICES): read once per launch,
                                       // may be set while other threads scan through this context
  int wide_chunk_rows = 0, wide_lds_bytes = 0;   // single-workgroup-per-CU layout (see make_plan)
  int item_chunk = 0;    // MTGPU_ITEM_CHUNK (tests): workgroups per kernel launch, 0 = 2^30
  int lds_max = 0;       // device limit of LDS per workgroup
  int group_request = 0; // MTGPU_GROUP: frames per workgroup, 0 = automatic
  int check_offsets = 0; // MTGPU_CHECK_OFFSETS=1: the device entry points verify "frame_off non-decreasing" first (one sync per call)
  int min_lds_kb = 0;    // MTGPU_MIN_LDS_KB: launch with at least this much LDS (caps workgroups per CU), 0 = automatic
  // Launch scratch (work lists of the device entry points, spill queues, slice tiles, merge workspaces): a ring of
  // device blocks, each with the event of its last user.  Whoever takes a block makes ITS stream wait for that event,
  // so a block never has two users at once, whatever streams and threads the launches come from.  (Through round 5
  // this was a hipMemPool with hipMallocFromPoolAsync / hipFreeAsync.  Round 6 made every launch use scratch, and 16
  // host threads on their own streams then saw launches trample each other's work lists — one wrong flag in a few
  // hundred launches, also with a mutex around the pool calls, also with one stream per batch; with memory that a
  // launch demonstrably owns, none: profiles/r06_host_stress_*.log.  The library no longer relies on cross-stream
  // reuse inside the runtime's stream-ordered allocator.)
  struct Scratch {
    static constexpr int kSlots = 16;
    static constexpr size_t kBig = 64u << 20;   // requests from here on wait for a block that is large enough rather than grow another
    struct Slot {
      void *p = nullptr; size_t cap = 0;
      hipEvent_t ev = nullptr;          // recorded behind the last launch that used the block
      hipStream_t last = nullptr;       // ... and the stream it ran on (a preference only: the event is waited for regardless)
      bool recorded = false;
      bool busy = false;                // a host thread is queueing work that uses the block (guarded by mu; the other
    };                                  // fields of a busy slot belong to that thread)
    std::mutex mu;
    std::condition_variable cv;
    Slot slot[kSlots];
    uint64_t reserved = 0, reserved_high = 0;
  } scratch;
  uint64_t merge_large_min = 4096;   // timestamp lists at least this long take the multi-workgroup merge (MTGPU_MERGE_LARGE_MIN)
  // Streams of the pipes (host dispatcher): ONE small pool per context, handed to staging batches round-robin,
  // instead of a stream per batch.  Creating a HIP stream costs ~3.5 ms and the runtime serialises it: 64 workers
  // x 3 batches = 192 streams took 0.68 s of wall time and 0.35 s per worker — THE cost of creating a pipe
  // (profiles/r04_pin_probe2.json) — while the runtime maps streams onto a handful of hardware queues anyway.
  static constexpr int kMaxPipeStreams = 32;
  hipStream_t pipe_streams[kMaxPipeStreams] = {};
  int n_pipe_streams = 8;                // MTGPU_PIPE_STREAMS (0: every batch creates its own stream, as before round 4)
  unsigned pipe_rr = 0;                  // next slot (guarded by pipe_mu)
  std::mutex pipe_mu;
  std::mutex mu;         // guards the staging buffers below
  DevBuf d_mv, d_off, d_sd, d_flags, d_misc;
  // launch timing (mtgpu_profile_enable): a ring of event triples {before planning, planned, scanned}
  struct Profile {
    static constexpr int kRing = 64;
    std::atomic<int> on{0};
    std::mutex mu;                       // held across a profiled launch
    hipEvent_t ev[kRing][3] = {};
    bool created = false;
    int tail = 0, count = 0;             // outstanding triples: tail .. tail + count - 1 (mod kRing)
    double plan_ms = 0.0, scan_ms = 0.0;
    uint32_t launches = 0;
    // waits for the oldest outstanding triple and adds its times up (mu held)
    hipError_t drain_one() {
      hipEvent_t *t = ev[tail];
      hipError_t e = hipEventSynchronize(t[2]);
      if (e != hipSuccess) return e;
      float a = 0.f, b = 0.f;
      if ((e = hipEventElapsedTime(&a, t[0], t[1])) != hipSuccess) return e;
      if ((e = hipEventElapsedTime(&b, t[1], t[2])) != hipSuccess) return e;
      plan_ms += a; scan_ms += b; ++launches;
      tail = (tail + 1) % kRing; --count;
      return hipSuccess;
    }
  } prof;
};

namespace {

// keep an MV iff !(mag_sq < T) with mag_sq a non-negative integer (src/motion_scanner.cpp:251)
//   <=>  mag_sq >= ceil(T)   (T NaN or <= 0: keep all;  T above any reachable mag_sq: keep none)
unsigned long long threshold_to_int(double t) {
  if (!(t > 0.0)) return 0ull;                       // NaN, 0, negative
  if (t > 17179869184.0) return ~0ull;               // > 2^34 > 2 * 65535^2: unreachable
  return (unsigned long long)std::ceil(t);
}

// The synchronous entry points queue copies from / to CALLER memory: whatever way they return, the
// stream has drained first, so the caller may free its buffers right away (also after an error).
struct DrainOnExit {
  hipStream_t st;
  ~DrainOnExit() { (void)hipStreamSynchronize(st); }
};

int validate_params(const mt_scan_params *p) {
  if (!p) return fail(MT_ERR_INVALID, "params is NULL");
  if (p->grid_w < 1 || p->grid_h < 1 || p->grid_w > 32767 || p->grid_h > 32767)
    return fail(MT_ERR_INVALID, "grid %dx%d outside [1,32767] (int16 in the reference)", p->grid_w, p->grid_h);
  if (p->block_shift < 0 || p->block_shift > 31)
    return fail(MT_ERR_INVALID, "block_shift %d outside [0,31]", p->block_shift);
  if (p->vertical_margin < 0)
    return fail(MT_ERR_INVALID, "vertical_margin %d < 0 (the reference would index before the grid)", p->vertical_margin);
  return MT_OK;
}

// LDS bytes of one workgroup: packed counters for (band_rows + 2) rows, a mask buffer for
// (chunk_rows + 2) rows, the centre total.
size_t lds_need(int band_rows, int chunk_rows, int gw, int W, int fb, int *cnt_words) {
  const size_t fields = (size_t)(band_rows + 2) * (size_t)gw;
  size_t cw = (fields * (size_t)fb + 31u) / 32u;
  cw = (cw + 3u) & ~(size_t)3u;
  if (cnt_words) *cnt_words = (int)cw;
  return cw * 4u + (size_t)(chunk_rows + 2) * (size_t)W * 8u + 16u;
}

using mtgpu::env_int;
using mtgpu::exp_int;

int make_plan(mtgpu_ctx *c, int lds_max, int cu_count) {
  const mt_scan_params &p = c->params;
  mtgpu::ScanK &k = c->k;
  k.thr = threshold_to_int(p.mv_threshold_sq);
  k.shift = p.block_shift;
  k.gw = p.grid_w;
  k.gh = p.grid_h;
  k.y_lo = p.vertical_margin;                                // :237
  k.y_hi = p.grid_h - p.vertical_margin;                     // :238
  if (k.y_hi < k.y_lo) k.y_hi = k.y_lo;                      // empty analysed range
  k.vec_need = p.vectors_needed;
  k.clust_need = p.clusters_needed < 1 ? 1u : (unsigned int)p.clusters_needed;
  k.W = (p.grid_w + 63) / 64;
  const int R = (k.y_hi - k.y_lo) < 1 ? 1 : (k.y_hi - k.y_lo);
  const size_t mask_row = (size_t)k.W * 8u;

  // Counter form (scan_kernels.hip).  Plain 32-bit adds whenever the whole tile fits LDS (the
  // fastest and contention-proof form); otherwise packed thermometer fields, the narrowest of
  // 1/2/4/8 bits >= vectors_needed; binary 8-bit CAS fields only for vectors_needed > 8.
  // MTGPU_FORCE_FB = 32 | 1 | 2 | 4 | 8 | 108 (8-bit CAS) overrides, for tests and experiments.
  const unsigned int vn = k.vec_need;
  int packed_fb = vn <= 1u ? 1 : (vn <= 2u ? 2 : (vn <= 4u ? 4 : 8));
  int packed_mode = vn <= 8u ? 1 : 2;
  int fb = 32, mode = 0;
  if (lds_need(R, R, k.gw, k.W, 32, nullptr) > (size_t)lds_max) { fb = packed_fb; mode = packed_mode; }
  const int force = env_int("MTGPU_FORCE_FB", 0);
  if (force == 32) { fb = 32; mode = 0; }
  else if (force == 108) { fb = 8; mode = 2; }
  else if ((force == 1 || force == 2 || force == 4 || force == 8) && (unsigned int)force >= vn) { fb = force; mode = 1; }
  else if (force == 1 || force == 2 || force == 4 || force == 8) { fb = packed_fb; mode = packed_mode; }
  k.mode = mode;
  k.active_min = mode == 1 ? ((1u << vn) - 1u) : vn;         // thermometer full / binary count

  int band_rows = R, chunk_rows = R;
  // MTGPU_MAX_TILE_KB (experiments): largest single tile; beyond it the grid is cut into row bands
  long single_max = lds_max;
  {
    const int tkb = exp_int("MTGPU_MAX_TILE_KB", 0);
    if (tkb > 0 && (long)tkb * 1024 < single_max) single_max = (long)tkb * 1024;
  }
  if (lds_need(R, R, k.gw, k.W, fb, nullptr) > (size_t)single_max) {
    // 1st choice: whole grid in one tile, phase 2 in row chunks through a smaller mask buffer
    const size_t cnt_only = lds_need(R, -2, k.gw, k.W, fb, nullptr);   // counters + total
    const long room = single_max - (long)cnt_only;
    long ch = room / (long)mask_row - 2;
    if (ch > R) ch = R;
    if (ch >= 8 || ch >= R) {
      chunk_rows = (int)ch;
    } else {
      // 2nd choice: row bands walked by ONE workgroup per frame (scan_kernels.hip, SPILL): the
      // records are still read once; later bands replay the queued votes.  As few, as large
      // bands as LDS allows: one workgroup per CU then owns the CU's whole bandwidth, so a frame
      // is finished (and the next one started) in half the time two co-resident workgroups would
      // need — with 20 MB work units that halves the idle tail of a launch (960x540, 1024 frames:
      // 2 bands 6.8 TB/s vs 4 bands 6.3-6.6), and every band less is one queue replay less on
      // vote-heavy input (pan: 5.1 vs 4.1 TB/s).  MTGPU_BAND_LDS_KB overrides the tile limit.
      long tile_max = (long)lds_max;
      const int fkb = exp_int("MTGPU_BAND_LDS_KB", 0);
      if (fkb > 0) tile_max = (long)fkb * 1024 < (long)lds_max ? (long)fkb * 1024 : (long)lds_max;
      const size_t per_row = ((size_t)k.gw * (size_t)fb + 7u) / 8u + mask_row;
      long r = 0;
      for (int pass = 0; pass < 2 && r < 1; ++pass) {             // the tile limit first, all of LDS if a row is that wide
        const long lim = pass == 0 ? tile_max : (long)lds_max;
        r = (lim - 64) / (long)per_row - 2;
        while (r >= 1 && lds_need((int)r, (int)r, k.gw, k.W, fb, nullptr) > (size_t)lim) --r;
      }
      if (r < 1)
        return fail(MT_ERR_CAPACITY, "grid width %d: three counter rows do not fit %d bytes of LDS", k.gw, lds_max);
      if (r > R) r = R;
      const long nb = ((long)R + r - 1) / r;                      // balance the bands
      band_rows = chunk_rows = (int)(((long)R + nb - 1) / nb);
    }
  }
  // Two workgroups per CU hide each other's zero / cluster-test phases (measured +4 % on the
  // 960x540 grid): when only the mask buffer pushes a tile over half of LDS, chunk the cluster
  // test so that the tile fits 80 KB.
  // That only helps when a launch has more work items than CUs; with fewer, one workgroup per
  // CU runs anyway and the single-chunk layout is faster (5.87 vs 5.50 TB/s at 256 frames), so
  // both layouts are kept and launch_scan_on picks per launch.
  c->wide_chunk_rows = chunk_rows;
  c->wide_lds_bytes = (int)lds_need(band_rows, chunk_rows, k.gw, k.W, fb, nullptr);
  {
    const size_t half = 80u * 1024u;
    const size_t cnt_only = lds_need(band_rows, -2, k.gw, k.W, fb, nullptr);
    if (lds_need(band_rows, chunk_rows, k.gw, k.W, fb, nullptr) > half && cnt_only + 10u * mask_row <= half) {
      const long ch = (long)((half - cnt_only) / mask_row) - 2;
      if (ch >= 8 && ch < chunk_rows) chunk_rows = (int)ch;
    }
  }
  const int fchunk = exp_int("MTGPU_FORCE_CHUNK", 0);         // experiments: smaller mask buffer
  if (fchunk >= 1 && fchunk < chunk_rows) { chunk_rows = fchunk; c->wide_chunk_rows = fchunk; c->wide_lds_bytes = (int)lds_need(band_rows, fchunk, k.gw, k.W, fb, nullptr); }
  k.fb = fb;
  k.band_rows = band_rows;
  k.chunk_rows = chunk_rows;
  k.bands = (R + band_rows - 1) / band_rows;
  k.mask_rows = chunk_rows + 2;
  const size_t lds = lds_need(band_rows, chunk_rows, k.gw, k.W, fb, &k.cnt_words);
  // Workgroup size: 512 threads x 4 loads in flight per lane keep a CU's memory queue full at
  // 4 workgroups/CU (measured +2.5 % over 256 on 1080p); tiles above 48 KB run 1-2
  // workgroups/CU and take 16 waves each.
  int block = lds <= 48u * 1024u ? 512 : 1024;
  // MTGPU_FORCE_BLOCK (tests): 512 | 1024.  The default build instantiates what the planner can choose plus that
  // switch: 512- and 1024-thread workgroups for single tiles, 1024 for banded plans (their tiles always exceed 48 KB);
  // 256-thread workgroups and 512-thread banded ones exist only in the experiments build.
  const int fblock = env_int("MTGPU_FORCE_BLOCK", 0);
  if (fblock == 512 || fblock == 1024 || (mtgpu::kExperiments && fblock == 256)) block = fblock;
  if (!mtgpu::kExperiments && k.bands > 1) block = 1024;
  c->plan.block_threads = block;
  c->plan.bands = k.bands;
  c->plan.band_rows = k.band_rows;
  c->plan.lds_bytes = (int)lds;
  c->plan.counter_bits = fb;
  c->plan.device = c->logical_device;
  c->plan.cu_count = cu_count;
  c->plan.chunk_rows = chunk_rows;
  c->variant = exp_int("MTGPU_VARIANT", 0);
  {
    const int fs = env_int("MTGPU_FORCE_SLICES", 0);
    c->slices_request.store((fs == 1 || fs == 2 || fs == 4 || fs == 8) ? fs : 0, std::memory_order_relaxed);
  }
  k.slices = 1;
  k.group = 1;
  k.sys_flags = 0;                                            // per launch: launch_scan_on
  k.resident = std::min(std::max(exp_int("MTGPU_RESIDENT", 0), 0), 16);   // experiments: ticketed resident workgroups per CU
  k.align_lines = exp_int("MTGPU_ALIGN", 1) != 0 ? 1 : 0;     // experiments: 0 = streams start wherever the frame starts
  k.prefetch = exp_int("MTGPU_PREFETCH", 1) != 0 ? 1 : 0;      // experiments: 0 switches the next-frame prefetch off
  c->group_request = env_int("MTGPU_GROUP", 0);
  c->min_lds_kb = exp_int("MTGPU_MIN_LDS_KB", 0);
  c->check_offsets = env_int("MTGPU_CHECK_OFFSETS", 0) != 0;
  c->n_pipe_streams = std::min(std::max(exp_int("MTGPU_PIPE_STREAMS", 8), 0), (int)mtgpu_ctx::kMaxPipeStreams);
  c->item_chunk = env_int("MTGPU_ITEM_CHUNK", 0);
  if (c->item_chunk < 0) c->item_chunk = 0;
  {
    const int lm = env_int("MTGPU_MERGE_LARGE_MIN", 0);
    if (lm > 0) c->merge_large_min = (uint64_t)lm;
  }
  c->plan.counter_mode = mode;
  c->plan._pad = 0;
  return MT_OK;
}

// Slices per frame for this launch.  Measured (scripts/ab_scan.py, AB_SET=slices): the hand-off
// (tile write + agent-scope release/acquire + tile reads) costs several microseconds per
// workgroup, so splitting pays only for batches that leave most CUs idle AND whose frames are
// large: 64 frames of the 960x540 grid +47 % with 4 slices, 64 4K frames +8 % with 2; every
// other shape loses (256 1080p frames: -40 % with 2).  Auto therefore never exceeds one
// workgroup per CU and keeps >= 32768 records (1.3 MB) per slice.
int choose_slices(const mtgpu_ctx *c, uint64_t n_records, uint32_t n_frames) {
  if (c->k.bands != 1) return 1;
  int s = c->slices_request.load(std::memory_order_relaxed);
  if (s == 0) {                                               // auto
    const uint64_t cus = (uint64_t)(c->plan.cu_count > 0 ? c->plan.cu_count : 256);
    const uint64_t avg = n_records / (n_frames ? n_frames : 1);
    s = 1;
    while (s < 8 && (uint64_t)n_frames * (uint64_t)(2 * s) <= cus && avg / (uint64_t)(2 * s) >= 32768ull) s *= 2;
  }
  return (s == 2 || s == 4 || s == 8) ? s : 1;
}

// Frames per workgroup.  Very short workgroups are started more slowly than they finish (the
// dispatcher starts ~19 per microsecond), so few are alive per CU and nothing overlaps their
// zeroing / cluster test: frames below ~128 KB are scanned two or more per workgroup.  Measured
// after the 32-bit kernels dropped to 58-61 VGPRs (four workgroups per CU): 65 KB frames (1080p,
// one compact record per cell) +5 % with 2-4 per workgroup; 252 KB and 326 KB frames are 2-3 %
// FASTER one per workgroup (the earlier 1 MB target dated from two workgroups per CU).
// Only for batches that fill the chip several times over.
int choose_group(const mtgpu_ctx *c, uint64_t n_records, uint32_t n_frames, int rec_bytes, int slices) {
  if (slices != 1 || n_frames == 0) return 1;
  int g = c->group_request;
  if (g <= 0) {
    const uint64_t avg = n_records * (uint64_t)rec_bytes / n_frames;
    const uint64_t cus = (uint64_t)(c->plan.cu_count > 0 ? c->plan.cu_count : 256);
    g = 1;
    // round 3, 40-byte records at 21 GB per launch: 48 KB frames (480p, one record per cell) 5.98 / 6.28 / 6.48 /
    // 6.70 TB/s with 1 / 2 / 4 / 8 frames per workgroup, 144 KB frames (720p) 7.01 / 7.15 / 7.19 / 7.13, 192 KB
    // frames (480p dense8x8) 7.22 / 7.08 / 7.04: frames up to ~160 KB are grouped up to ~600 KB per workgroup
    if (avg <= 160ull * 1024ull)
      while (g < 8 && avg * (uint64_t)(2 * g) <= 600ull * 1024ull && (uint64_t)n_frames >= cus * 8ull * (uint64_t)(2 * g)) g *= 2;
    // Compact records on a tile that leaves ONE workgroup per CU (4K: 124 KB of 32-bit counters): nothing overlaps
    // that workgroup's zeroing and cluster test (5 of 38 us per ~1 MB frame), so it scans 2-4 frames in a row
    // and issues the next frame's first streaming step before its cluster test (scan_kernels.hip, NextStep).
    // Round 3, 4K dense8x8 compact: 1024 frames 6.13 -> 6.39 TB/s, 4096 frames 6.54 -> 6.73; 4 MB frames
    // (960x540) and four-per-CU tiles (1080p) gain nothing or lose, 40-byte records lose 2 % (no prefetch there).
    if (g == 1 && rec_bytes == MT_COMPACT_BYTES && c->k.prefetch && c->plan.lds_bytes > 80 * 1024 && avg <= (2ull << 20)) {
      if ((uint64_t)n_frames >= cus * 4ull) g = 4;
      else if ((uint64_t)n_frames >= cus * 2ull) g = 2;
    }
    // Round 6 (a workgroup that owns several list entries parks them in LDS when it starts and issues the next frame's
    // first loads before this frame's cluster test): profiles/r06_group_ab.log, wall clock per call —
    //   compact records on tiles that share a CU (1080p: 261 KB frames, 37 us per workgroup, of which ~3 us are
    //   start-up): two frames per workgroup 635 -> 594 us at 16 384 frames, 174 -> 160 us at 4096 (four: 601 / 161);
    //   40-byte records on a one-workgroup-per-CU tile (4K: 5.2 MB frames): 2895 -> 2881 us with two on equal frames,
    //   but -3 % on ragged ones (profiles/r06_group_ab.log, second part: 10 MB work units leave a long tail) — not taken;
    //   40-byte records on shared tiles (1080p 1.3 MB frames, 326 KB frames): nothing or a loss — they stay at one.
    // Only where every workgroup slot of the chip is still filled twice over afterwards.
    const uint64_t per_cu = c->plan.lds_bytes <= 40 * 1024 ? 4u : (c->plan.lds_bytes <= 80 * 1024 ? 2u : 1u);
    if (g == 1 && c->k.bands == 1 && rec_bytes == MT_COMPACT_BYTES && c->k.prefetch && avg <= (2ull << 20) &&
        (uint64_t)n_frames >= cus * per_cu * 4ull)
      g = 2;
  }
  if (g > 64) g = 64;
  return g < 1 ? 1 : g;
}

// A scratch block of at least `bytes` for work queued on `st` (see mtgpu_ctx::Scratch).  The block is the caller's
// until scratch_release, which must follow the LAST launch that uses it.  Preference: a block that is large enough and
// whose last user has finished; an unused or small block to grow (small requests only); a block that is large enough
// but still in use (the stream then waits for it).  Growing synchronises with the block's last user and calls
// hipFree / hipMalloc: it happens while a context warms up, not in steady state.
int scratch_acquire(mtgpu_ctx *c, size_t bytes, hipStream_t st, int *slot_out, void **p_out) {
  using Scratch = mtgpu_ctx::Scratch;
  Scratch &sc = c->scratch;
  int pick = -1;
  {
    std::unique_lock<std::mutex> lock(sc.mu);
    for (;;) {
      // 1. a block this stream used last: stream order alone makes it safe, nothing to ask the runtime
      for (int i = 0; i < Scratch::kSlots && pick < 0; ++i) {
        const Scratch::Slot &s = sc.slot[i];
        if (!s.busy && s.cap >= bytes && s.recorded && s.last == st) pick = i;
      }
      // 2. a block that is large enough and whose last user has finished
      int fit_busy = -1, small_done = -1, small_busy = -1, unused = -1;
      for (int i = 0; i < Scratch::kSlots && pick < 0; ++i) {
        const Scratch::Slot &s = sc.slot[i];
        if (s.busy) continue;
        if (!s.recorded && s.cap >= bytes) { pick = i; break; }
        if (!s.recorded) { if (unused < 0 || s.cap > sc.slot[unused].cap) unused = i; continue; }
        const bool done = hipEventQuery(s.ev) == hipSuccess;
        if (s.cap >= bytes) { if (done) pick = i; else if (fit_busy < 0) fit_busy = i; }
        else if (done) { if (small_done < 0 || s.cap < sc.slot[small_done].cap) small_done = i; }
        else if (small_busy < 0) small_busy = i;
      }
      (void)hipGetLastError();                                  // (hipEventQuery's "not ready" is not an error of ours)
      // 3. small requests grow an idle block rather than wait; large ones (spill queues: gigabytes) wait for one that fits
      if (pick < 0 && bytes < Scratch::kBig) pick = unused >= 0 ? unused : small_done;
      if (pick < 0) pick = fit_busy;
      if (pick < 0) pick = unused >= 0 ? unused : (small_done >= 0 ? small_done : small_busy);
      if (pick >= 0) break;
      sc.cv.wait(lock);                                          // every block is in another thread's hands this microsecond
    }
    sc.slot[pick].busy = true;
  }
  mtgpu_ctx::Scratch::Slot &s = sc.slot[pick];
  hipError_t e = hipSuccess;
  if (!s.ev) e = hipEventCreateWithFlags(&s.ev, hipEventDisableTiming);
  if (e == hipSuccess && s.cap < bytes) {
    if (s.recorded) e = hipEventSynchronize(s.ev);
    if (e == hipSuccess && s.p) { e = hipFree(s.p); s.p = nullptr; }
    const size_t old = s.cap;
    s.cap = 0;
    s.recorded = false;
    void *np = nullptr;
    const size_t want = bytes + bytes / 4 + 256;
    if (e == hipSuccess) e = hipMalloc(&np, want);
    if (e == hipSuccess) { s.p = np; s.cap = want; }
    std::lock_guard<std::mutex> lock(sc.mu);
    sc.reserved += s.cap;
    sc.reserved -= old;
    if (sc.reserved > sc.reserved_high) sc.reserved_high = sc.reserved;
  } else if (e == hipSuccess && s.recorded) {
    e = hipStreamWaitEvent(st, s.ev, 0);                        // after the block's last user, on whatever stream that was
  }
  if (e != hipSuccess) {
    { std::lock_guard<std::mutex> lock(sc.mu); s.busy = false; }
    sc.cv.notify_one();
    return hip_fail(e, "launch scratch");
  }
  *slot_out = pick;
  *p_out = s.p;
  return MT_OK;
}

// The caller has queued its last use of the block on `st`.
void scratch_release(mtgpu_ctx *c, int slot, hipStream_t st) {
  mtgpu_ctx::Scratch &sc = c->scratch;
  mtgpu_ctx::Scratch::Slot &s = sc.slot[slot];
  if (hipEventRecord(s.ev, st) == hipSuccess) {
    s.recorded = true;
    s.last = st;
  } else {                                                      // cannot tell when the work ends: wait for it here
    (void)hipGetLastError();
    (void)hipStreamSynchronize(st);
    s.recorded = false;
  }
  { std::lock_guard<std::mutex> lock(sc.mu); s.busy = false; }
  sc.cv.notify_one();
}

// Launches the scan on `st`; scratch (band centre counts, slice tiles + tickets) is allocated
// and freed stream-ordered, so concurrent callers share nothing.
// flags_sys: 1 = d_flags is not device memory (system-scope result stores), 0 = device memory, -1 = ask the runtime
int launch_scan_on(mtgpu_ctx *c, const void *d_mv, uint64_t n_records, const uint64_t *d_off,
                   const uint8_t *d_sd, uint32_t n_frames, uint8_t *d_flags, hipStream_t st, int rec_bytes = MT_MV_BYTES,
                   int flags_sys = 0, uint64_t rebase = 0, void *own_plan_ws = nullptr, size_t own_plan_ws_bytes = 0) {
  if (flags_sys < 0) {
    // a caller's pointer (the *_device entry points): device memory takes plain stores; anything else the runtime
    // knows of, or does not know at all, takes the system-scope ones
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof at);
    const hipError_t qe = hipPointerGetAttributes(&at, d_flags);
    flags_sys = (qe == hipSuccess && at.type == hipMemoryTypeDevice) ? 0 : 1;
    if (qe != hipSuccess) (void)hipGetLastError();      // only the failed query's own error is cleared
  }
  mtgpu::ScanLaunch L;
  L.rec_bytes = rec_bytes;
  L.lds_max = c->lds_max;
  L.device = c->device;
  L.cu_count = c->plan.cu_count;
  L.mv = static_cast<const unsigned char *>(d_mv);
  L.n_records = n_records;
  L.rebase = rebase;
  L.frame_off = reinterpret_cast<const unsigned long long *>(d_off);
  L.has_sd = d_sd;
  L.n_frames = n_frames;
  L.flags = d_flags;
  L.spill_q = nullptr;
  L.slice_ws = nullptr;
  L.tickets = nullptr;
  L.plan_ws = nullptr;
  L.k = c->k;
  L.k.sys_flags = flags_sys;
  L.k.slices = choose_slices(c, n_records - rebase, n_frames);
  L.k.group = choose_group(c, n_records - rebase, n_frames, rec_bytes, L.k.slices);
  if ((uint64_t)n_frames * (uint64_t)L.k.slices >= (1ull << 32))
    return fail(MT_ERR_INVALID, "%u frames x %d slices: work items must stay below 2^32 per call", n_frames, L.k.slices);
  L.block = c->plan.block_threads;
  L.variant = c->variant;
  L.item_chunk = (unsigned long long)c->item_chunk;
  L.lds_bytes = c->plan.lds_bytes;
  if (c->wide_lds_bytes > c->plan.lds_bytes &&
      (uint64_t)n_frames * (uint64_t)L.k.bands * (uint64_t)L.k.slices <= (uint64_t)c->plan.cu_count) {
    L.k.chunk_rows = c->wide_chunk_rows;        // at most one workgroup per CU: bigger mask buffer, fewer chunks
    L.k.mask_rows = c->wide_chunk_rows + 2;
    L.lds_bytes = c->wide_lds_bytes;
  }
  // several frames per workgroup: their list entries are parked in LDS behind the tile (scan_kernels.hip, stage_items)
  L.k.stage_word = 0;
  if (L.k.group > 1) {
    const int at = (L.lds_bytes + 16 + 31) & ~31;             // (+16: the experiments build's ticket word sits right behind the tile)
    if (at + mtgpu::kStageBytes <= c->lds_max) {
      L.k.stage_word = at / 4;
      L.lds_bytes = at + mtgpu::kStageBytes;
    } else {
      L.k.group = 1;                                          // a tile that fills LDS to the last 2 KB: one frame per workgroup
    }
  }
  if (c->min_lds_kb > 0) L.lds_bytes = std::max(L.lds_bytes, std::min(c->min_lds_kb * 1024, c->lds_max));
  L.stream = st;
  // launch scratch, one stream-ordered block: [work list + planning counts | spill queue or slice tiles + tickets]
  // the caller's own block for the work list (a pipe's batch), if it is large enough: the pool then serves only
  // the spill queue / the slice tiles, i.e. nothing at all for single-tile plans
  const bool own_plan = own_plan_ws && ((uintptr_t)own_plan_ws & 255u) == 0u && own_plan_ws_bytes >= mtgpu::plan_scratch_bytes(n_frames);
  const size_t plan_bytes = own_plan ? 0u : ((mtgpu::plan_scratch_bytes(n_frames) + 255u) & ~(size_t)255u);
  size_t bytes = plan_bytes;
  if (L.k.bands > 1) bytes += sizeof(unsigned int) * ((size_t)(n_records - rebase) + 4);   // spill queue: a slot per record
  if (L.k.slices > 1)
    bytes += sizeof(unsigned int) * ((size_t)n_frames * (size_t)L.k.slices * (size_t)L.k.cnt_words + (size_t)n_frames + 4);
  void *scratch = nullptr;
  int slot = -1;
  hipError_t e = hipSuccess;
  if (bytes) {
    const int rc0 = scratch_acquire(c, bytes, st, &slot, &scratch);
    if (rc0 != MT_OK) return rc0;
  }
  L.plan_ws = own_plan ? own_plan_ws : scratch;
  unsigned int *rest = reinterpret_cast<unsigned int *>(static_cast<unsigned char *>(scratch) + plan_bytes);
  if (L.k.bands > 1) L.spill_q = rest;
  if (L.k.slices > 1) {
    L.slice_ws = rest;
    L.tickets = L.slice_ws + (((size_t)n_frames * (size_t)L.k.slices * (size_t)L.k.cnt_words + 3) & ~(size_t)3);
  }
  L.ev_planned = nullptr;
  int rc = MT_OK;
  if (c->prof.on.load(std::memory_order_relaxed)) {
    mtgpu_ctx::Profile &pf = c->prof;
    std::lock_guard<std::mutex> lock(pf.mu);
    if (pf.created && pf.count == mtgpu_ctx::Profile::kRing) e = pf.drain_one();
    if (e == hipSuccess && pf.created) {
      hipEvent_t *t = pf.ev[(pf.tail + pf.count) % mtgpu_ctx::Profile::kRing];
      e = hipEventRecord(t[0], st);
      L.ev_planned = t[1];
      if (e == hipSuccess) e = mtgpu::launch_scan(L);
      if (e == hipSuccess) e = hipEventRecord(t[2], st);
      if (e == hipSuccess) ++pf.count;
    } else if (e == hipSuccess) {
      e = mtgpu::launch_scan(L);
    }
  } else {
    e = mtgpu::launch_scan(L);
  }
  if (e != hipSuccess) rc = hip_fail(e, "scan launch");
  if (slot >= 0) scratch_release(c, slot, st);
  return rc;
}

}  // namespace

namespace {
// MTGPU_ALIAS_DEVICES=N (tests, rehearsals on a box with fewer GPUs than the target node): the library presents N
// LOGICAL devices; logical device d runs on physical device d % (real count).  Everything above the C ABI — the
// host layer's stream -> device assignment, one shared context and scratch pool per (device, parameters), pipes
// per device — then takes its multi-device paths on a 1-GPU box.  Read once; 0 / unset = no aliasing.
int alias_devices() {
  static const int n = [] { const char *e = getenv("MTGPU_ALIAS_DEVICES"); return e ? std::max(0, atoi(e)) : 0; }();
  return n;
}
}  // namespace

namespace mtgpu {
// (the host copy-out loop, pack_records, lives in pack_simd.cpp: a plain host TU with per-CPU dispatch)
int ctx_device(const mtgpu_ctx *c) { return c->device; }
// A stream for one staging batch from the context's pool (created on first use of its slot; the caller has made
// the context's device current), or nullptr when pooling is off or the creation failed (the batch then creates
// a stream of its own).
hipStream_t ctx_pipe_stream(mtgpu_ctx *c) {
  if (c->n_pipe_streams <= 0) return nullptr;
  std::lock_guard<std::mutex> lock(c->pipe_mu);
  const unsigned slot = c->pipe_rr++ % (unsigned)c->n_pipe_streams;
  if (!c->pipe_streams[slot] &&
      hipStreamCreateWithFlags(&c->pipe_streams[slot], hipStreamNonBlocking) != hipSuccess) {
    c->pipe_streams[slot] = nullptr;
    (void)hipGetLastError();
  }
  return c->pipe_streams[slot];
}
int physical_device(int logical) {
  int n = 0;
  if (alias_devices() <= 0 || hipGetDeviceCount(&n) != hipSuccess || n < 1) return logical;
  return logical >= 0 ? logical % n : logical;
}
int ctx_launch_scan(mtgpu_ctx *c, const void *d_mv, uint64_t n_records, const uint64_t *d_off,
                    const uint8_t *d_sd, uint32_t n_frames, uint8_t *d_flags, hipStream_t st, int rec_bytes, int flags_in_host_memory,
                    void *plan_ws, size_t plan_ws_bytes) {
  return launch_scan_on(c, d_mv, n_records, d_off, d_sd, n_frames, d_flags, st, rec_bytes, flags_in_host_memory ? 1 : 0, 0,
                        plan_ws, plan_ws_bytes);
}
size_t ctx_plan_ws_bytes(uint32_t n_frames) { return (plan_scratch_bytes(n_frames) + 255u) & ~(size_t)255u; }
}  // namespace mtgpu

#ifdef MTGPU_PHASE_TIMES
namespace mtgpu { hipError_t debug_set_phase_times(unsigned long long *p); }
extern "C" int mtgpu_debug_set_phase_times(void *p) {      // developer build only, not in include/mtgpu.h
  return mtgpu::debug_set_phase_times(static_cast<unsigned long long *>(p)) == hipSuccess ? MT_OK : MT_ERR_DEVICE;
}
#endif

extern "C" {

const char *mtgpu_version(void) {
  return mtgpu::kExperiments ? "mtgpu 0.6 (gfx950 MV scan + segment merge) +experiments" : "mtgpu 0.6 (gfx950 MV scan + segment merge)";
}

const char *mtgpu_last_error(void) { return g_err; }

int mtgpu_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  if (n > 0 && alias_devices() > 0) return alias_devices();
  return n;
}

int mtgpu_device_pci_address(int device, char *buf, uint64_t cap) {
  if (!buf) return fail(MT_ERR_INVALID, "buf is NULL");
  if (cap < 13) return fail(MT_ERR_CAPACITY, "buffer too small for a PCI address (13 bytes)");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n < 1) return fail(MT_ERR_DEVICE, "no HIP device available");
  const int logical = alias_devices() > 0 ? alias_devices() : n;
  if (device < 0 || device >= logical) return fail(MT_ERR_INVALID, "device %d outside [0,%d)", device, logical);
  char tmp[64] = "";
  hipError_t e = hipDeviceGetPCIBusId(tmp, (int)sizeof tmp, device % n);
  if (e != hipSuccess) return hip_fail(e, "hipDeviceGetPCIBusId");
  size_t i = 0;
  for (; tmp[i] && i + 1 < cap; ++i) buf[i] = (char)std::tolower((unsigned char)tmp[i]);
  buf[i] = 0;
  return MT_OK;
}

int mtgpu_params_from_config(mt_scan_params *out, int width, int height, double mv_threshold_sq,
                             int block_size, int block_shift, int vectors_needed,
                             int clusters_needed, float vertical_mask) {
  if (!out) return fail(MT_ERR_INVALID, "out is NULL");
  if (block_shift < 0 || block_shift > 31) return fail(MT_ERR_INVALID, "block_shift %d outside [0,31]", block_shift);
  // src/motion_scanner.cpp:190-193
  const long long gw = ((long long)width + block_size - 1) >> block_shift;
  const long long gh = ((long long)height + block_size - 1) >> block_shift;
  if (gw < 1 || gh < 1 || gw > 32767 || gh > 32767)
    return fail(MT_ERR_INVALID, "grid %lldx%lld outside [1,32767]", gw, gh);
  std::memset(out, 0, sizeof *out);
  out->mv_threshold_sq = mv_threshold_sq;              // :184
  out->block_shift = block_shift;                      // :185
  out->vectors_needed = (uint8_t)vectors_needed;       // :186, config.hpp:75
  out->clusters_needed = clusters_needed;              // :187
  out->grid_w = (int32_t)gw;
  out->grid_h = (int32_t)gh;
  const float margin = (float)(int16_t)gh * vertical_mask;   // :196, float32 product
  out->vertical_margin = (int)margin;
  return MT_OK;
}

int mtgpu_create(const mt_scan_params *params, int device, mtgpu_ctx **out) {
  if (!out) return fail(MT_ERR_INVALID, "out is NULL");
  *out = nullptr;
  int rc = validate_params(params);
  if (rc != MT_OK) return rc;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(MT_ERR_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  const int logical_devices = alias_devices() > 0 ? alias_devices() : ndev;
  if (device < 0 || device >= logical_devices) return fail(MT_ERR_INVALID, "device %d outside [0,%d)", device, logical_devices);
  const int logical = device;
  device = device % ndev;                    // (MTGPU_ALIAS_DEVICES: logical -> physical)
  HIP_TRY(hipSetDevice(device));
  int lds_max = 0, cus = 0;
  HIP_TRY(hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, device));
  HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
  mtgpu_ctx *c = new (std::nothrow) mtgpu_ctx();
  if (!c) return fail(MT_ERR_NOMEM, "out of host memory");
  c->params = *params;
  c->device = device;
  c->logical_device = logical;
  c->lds_max = lds_max;
  c->stream = nullptr;
  rc = make_plan(c, lds_max, cus);
  if (rc != MT_OK) { delete c; return rc; }
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
  *out = c;
  return MT_OK;
}

void mtgpu_destroy(mtgpu_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
  for (hipStream_t &ps : c->pipe_streams)            // (pipes are destroyed before their context: include/mtgpu.h)
    if (ps) { (void)hipStreamSynchronize(ps); (void)hipStreamDestroy(ps); ps = nullptr; }
  for (auto &t : c->prof.ev)
    for (hipEvent_t &e : t)
      if (e) { (void)hipEventDestroy(e); e = nullptr; }
  for (auto &sl : c->scratch.slot) {
    if (sl.ev) { if (sl.recorded) (void)hipEventSynchronize(sl.ev); (void)hipEventDestroy(sl.ev); sl.ev = nullptr; }
    if (sl.p) { (void)hipFree(sl.p); sl.p = nullptr; }
  }
  c->d_mv.release(); c->d_off.release(); c->d_sd.release(); c->d_flags.release(); c->d_misc.release();
  delete c;
}

int mtgpu_get_stats(mtgpu_ctx *c, mtgpu_ctx_stats *out) {
  if (!c || !out) return fail(MT_ERR_INVALID, "NULL argument");
  std::memset(out, 0, sizeof *out);
  std::lock_guard<std::mutex> lock(c->mu);
  out->staging_device_bytes = c->d_mv.cap + c->d_off.cap + c->d_sd.cap + c->d_flags.cap + c->d_misc.cap;
  out->hip_streams = c->stream ? 1u : 0u;
  {
    std::lock_guard<std::mutex> pl(c->pipe_mu);
    for (hipStream_t ps : c->pipe_streams) out->hip_streams += ps ? 1u : 0u;
  }
  out->private_pool = 1u;
  {
    std::lock_guard<std::mutex> sl(c->scratch.mu);
    out->pool_reserved_bytes = c->scratch.reserved;
    out->pool_reserved_high = c->scratch.reserved_high;
  }
  return MT_OK;
}

int mtgpu_trim(mtgpu_ctx *c) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  HIP_TRY(hipSetDevice(c->device));
  // blocks nobody holds and whose last user has finished go back to the device; the others stay
  mtgpu_ctx::Scratch &sc = c->scratch;
  for (int i = 0; i < mtgpu_ctx::Scratch::kSlots; ++i) {
    mtgpu_ctx::Scratch::Slot &s = sc.slot[i];
    {
      std::lock_guard<std::mutex> lock(sc.mu);
      if (s.busy || !s.p) continue;
      if (s.recorded && hipEventQuery(s.ev) != hipSuccess) { (void)hipGetLastError(); continue; }
      s.busy = true;                                            // ours while it is being freed
    }
    const hipError_t e = hipFree(s.p);
    {
      std::lock_guard<std::mutex> lock(sc.mu);
      if (e == hipSuccess) { sc.reserved -= s.cap; s.p = nullptr; s.cap = 0; s.recorded = false; }
      s.busy = false;
    }
    sc.cv.notify_one();
    if (e != hipSuccess) return hip_fail(e, "hipFree(scratch)");
  }
  return MT_OK;
}

int mtgpu_get_params(const mtgpu_ctx *c, mt_scan_params *out) {
  if (!c || !out) return fail(MT_ERR_INVALID, "NULL argument");
  *out = c->params;
  return MT_OK;
}

int mtgpu_plan_preview(const mt_scan_params *params, int lds_bytes_per_workgroup, int cu_count, mtgpu_plan *out) {
  if (!out) return fail(MT_ERR_INVALID, "out is NULL");
  int rc = validate_params(params);
  if (rc != MT_OK) return rc;
  if (lds_bytes_per_workgroup < 1024 || cu_count < 1) return fail(MT_ERR_INVALID, "LDS size / CU count out of range");
  mtgpu_ctx *tmp = new (std::nothrow) mtgpu_ctx();          // host object only: no HIP call is made
  if (!tmp) return fail(MT_ERR_NOMEM, "out of host memory");
  tmp->params = *params;
  tmp->device = -1;
  tmp->logical_device = -1;
  tmp->stream = nullptr;
  tmp->lds_max = lds_bytes_per_workgroup;
  rc = make_plan(tmp, lds_bytes_per_workgroup, cu_count);
  if (rc == MT_OK) *out = tmp->plan;
  delete tmp;
  return rc;
}

int mtgpu_get_plan(const mtgpu_ctx *c, mtgpu_plan *out) {
  if (!c || !out) return fail(MT_ERR_INVALID, "NULL argument");
  *out = c->plan;
  return MT_OK;
}

namespace {
// MTGPU_CHECK_OFFSETS=1: verify the CSR precondition of the device entry points on the device, BEFORE the scan is
// queued — costs one small kernel and one stream synchronisation per call, which is why it is opt-in.
int check_offsets_on(mtgpu_ctx *c, const uint64_t *d_off, uint32_t n_frames, hipStream_t st) {
  void *d_bad = nullptr;
  int slot = -1;
  const int rc0 = scratch_acquire(c, sizeof(unsigned int), st, &slot, &d_bad);
  if (rc0 != MT_OK) return rc0;
  unsigned int first_bad = 0xffffffffu;
  hipError_t e = hipMemsetAsync(d_bad, 0xff, sizeof(unsigned int), st);
  if (e == hipSuccess)
    e = mtgpu::launch_check_offsets(reinterpret_cast<const unsigned long long *>(d_off), n_frames,
                                    static_cast<unsigned int *>(d_bad), st);
  if (e == hipSuccess) e = hipMemcpyAsync(&first_bad, d_bad, sizeof first_bad, hipMemcpyDeviceToHost, st);
  const hipError_t e2 = hipStreamSynchronize(st);
  scratch_release(c, slot, st);
  if (e != hipSuccess) return hip_fail(e, "frame_off check");
  if (e2 != hipSuccess) return hip_fail(e2, "hipStreamSynchronize(frame_off check)");
  if (first_bad != 0xffffffffu)
    return fail(MT_ERR_INVALID, "frame_off[%u] > frame_off[%u]: record offsets must be non-decreasing "
                "(frames are disjoint record ranges; a banded plan keeps each frame's spill queue at its offset)",
                first_bad, first_bad + 1u);
  return MT_OK;
}
}  // namespace

int mtgpu_scan_frames_device(mtgpu_ctx *c, const void *d_mv, uint64_t n_records,
                             const uint64_t *d_frame_off, const uint8_t *d_has_sd,
                             uint32_t n_frames, uint8_t *d_flags, void *stream) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  if (n_frames == 0) return MT_OK;
  if (!d_frame_off || !d_flags) return fail(MT_ERR_INVALID, "frame_off/flags is NULL");
  if (n_records > 0 && !d_mv) return fail(MT_ERR_INVALID, "mv is NULL with n_records > 0");
  if (((uintptr_t)d_mv & 3u) != 0) return fail(MT_ERR_INVALID, "mv must be 4-byte aligned");
  HIP_TRY(hipSetDevice(c->device));
  if (c->check_offsets) {
    const int rc = check_offsets_on(c, d_frame_off, n_frames, static_cast<hipStream_t>(stream));
    if (rc != MT_OK) return rc;
  }
  return launch_scan_on(c, d_mv, n_records, d_frame_off, d_has_sd, n_frames, d_flags,
                        static_cast<hipStream_t>(stream), MT_MV_BYTES, -1);
}

int mtgpu_scan_frames_device_compact(mtgpu_ctx *c, const void *d_rec8, uint64_t n_records,
                                     const uint64_t *d_frame_off, const uint8_t *d_has_sd,
                                     uint32_t n_frames, uint8_t *d_flags, void *stream) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  if (n_frames == 0) return MT_OK;
  if (!d_frame_off || !d_flags) return fail(MT_ERR_INVALID, "frame_off/flags is NULL");
  if (n_records > 0 && !d_rec8) return fail(MT_ERR_INVALID, "records is NULL with n_records > 0");
  if (((uintptr_t)d_rec8 & 7u) != 0) return fail(MT_ERR_INVALID, "compact records must be 8-byte aligned");
  HIP_TRY(hipSetDevice(c->device));
  if (c->check_offsets) {
    const int rc = check_offsets_on(c, d_frame_off, n_frames, static_cast<hipStream_t>(stream));
    if (rc != MT_OK) return rc;
  }
  return launch_scan_on(c, d_rec8, n_records, d_frame_off, d_has_sd, n_frames, d_flags,
                        static_cast<hipStream_t>(stream), MT_COMPACT_BYTES, -1);
}

int mtgpu_pack_records(const void *mv_bytes, uint64_t n_records, void *out8) {
  if (n_records == 0) return MT_OK;
  if (!mv_bytes || !out8) return fail(MT_ERR_INVALID, "NULL argument");
  mtgpu::pack_records(static_cast<const unsigned char *>(mv_bytes), n_records, static_cast<unsigned char *>(out8));
  return MT_OK;
}

int mtgpu_pack_records_with(int impl_flags, const void *mv_bytes, uint64_t n_records, void *out8) {
  const int impl = impl_flags & MT_PACK_IMPL_MASK;
  if ((impl_flags & ~(MT_PACK_IMPL_MASK | MT_PACK_NT | MT_PACK_PREFETCH_LINES(255))) != 0 || impl < MT_PACK_SCALAR ||
      impl > MT_PACK_AVX512)
    return fail(MT_ERR_INVALID, "impl_flags must be MT_PACK_SCALAR, MT_PACK_AVX2 or MT_PACK_AVX512, optionally "
                "| MT_PACK_NT | MT_PACK_PREFETCH_LINES(0..255)");
  if (n_records > 0 && (!mv_bytes || !out8)) return fail(MT_ERR_INVALID, "NULL argument");
  const uint64_t prefetch = 64ull * (uint64_t)((impl_flags >> 8) & 255);
  if (mtgpu::pack_records_with(impl_flags & (MT_PACK_IMPL_MASK | MT_PACK_NT), static_cast<const unsigned char *>(mv_bytes),
                               n_records, static_cast<unsigned char *>(out8), prefetch) != 0)
    return fail(MT_ERR_UNSUPPORTED, "this CPU cannot run the requested copy-out loop");
  return MT_OK;
}

int mtgpu_pack_selected(void) { return mtgpu::pack_selected(); }

int mtgpu_profile_enable(mtgpu_ctx *c, int on) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  mtgpu_ctx::Profile &pf = c->prof;
  std::lock_guard<std::mutex> lock(pf.mu);
  if (on && !pf.created) {
    HIP_TRY(hipSetDevice(c->device));
    for (auto &t : pf.ev)
      for (hipEvent_t &e : t) HIP_TRY(hipEventCreate(&e));       // (a failure leaves what was created to mtgpu_destroy)
    pf.created = true;
  }
  pf.on.store(on ? 1 : 0, std::memory_order_relaxed);
  return MT_OK;
}

int mtgpu_profile_read(mtgpu_ctx *c, double *plan_ms, double *scan_ms, uint32_t *launches) {
  if (!c || !plan_ms || !scan_ms || !launches) return fail(MT_ERR_INVALID, "NULL argument");
  mtgpu_ctx::Profile &pf = c->prof;
  std::lock_guard<std::mutex> lock(pf.mu);
  while (pf.count > 0) {
    const hipError_t e = pf.drain_one();
    if (e != hipSuccess) return hip_fail(e, "profile events");
  }
  *plan_ms = pf.plan_ms; *scan_ms = pf.scan_ms; *launches = pf.launches;
  pf.plan_ms = pf.scan_ms = 0.0;
  pf.launches = 0;
  return MT_OK;
}

int mtgpu_set_slices(mtgpu_ctx *c, int slices) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  if (slices != 0 && slices != 1 && slices != 2 && slices != 4 && slices != 8)
    return fail(MT_ERR_INVALID, "slices must be 0 (auto), 1, 2, 4 or 8");
  c->slices_request.store(slices, std::memory_order_relaxed);
  return MT_OK;
}

int mtgpu_scan_frames(mtgpu_ctx *c, const mt_mv *mv, const uint64_t *frame_off,
                      const uint8_t *has_sd, uint32_t n_frames, uint8_t *flags) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  if (n_frames == 0) return MT_OK;
  if (!frame_off || !flags) return fail(MT_ERR_INVALID, "frame_off/flags is NULL");
  for (uint32_t f = 0; f < n_frames; ++f)
    if (frame_off[f + 1] < frame_off[f])
      return fail(MT_ERR_INVALID, "frame_off not monotonic at frame %u", f);
  const uint64_t r_begin = frame_off[0], r_end = frame_off[n_frames];
  const uint64_t n_records = r_end - r_begin;
  if (n_records > 0 && !mv) return fail(MT_ERR_INVALID, "mv is NULL with records present");

  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  int rc;
  if ((rc = c->d_mv.reserve((size_t)n_records * MT_MV_BYTES + 16)) != MT_OK) return rc;
  if ((rc = c->d_off.reserve(sizeof(uint64_t) * ((size_t)n_frames + 1))) != MT_OK) return rc;
  if ((rc = c->d_flags.reserve(n_frames)) != MT_OK) return rc;
  if (has_sd && (rc = c->d_sd.reserve(n_frames)) != MT_OK) return rc;

  hipStream_t st = c->stream;
  DrainOnExit drain{st};
  if (n_records)
    HIP_TRY(hipMemcpyAsync(c->d_mv.p, mv + r_begin, (size_t)n_records * MT_MV_BYTES, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(c->d_off.p, frame_off, sizeof(uint64_t) * ((size_t)n_frames + 1), hipMemcpyHostToDevice, st));
  if (has_sd) HIP_TRY(hipMemcpyAsync(c->d_sd.p, has_sd, n_frames, hipMemcpyHostToDevice, st));
  // records of frame f live at d_mv + (frame_off[f] - r_begin) * 40: the work list is built with rebased offsets
  rc = launch_scan_on(c, c->d_mv.p, r_end, static_cast<const uint64_t *>(c->d_off.p),
                      has_sd ? static_cast<const uint8_t *>(c->d_sd.p) : nullptr, n_frames,
                      static_cast<uint8_t *>(c->d_flags.p), st, MT_MV_BYTES, 0, r_begin);
  if (rc != MT_OK) return rc;
  HIP_TRY(hipMemcpyAsync(flags, c->d_flags.p, n_frames, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MT_OK;
}

int mtgpu_merge_streams_device(mtgpu_ctx *c, const uint8_t *d_flags, const double *d_pts,
                               const uint64_t *d_stream_off, uint32_t n_streams,
                               const mt_merge_params *d_mp, int job_semantics, double *d_ts,
                               mt_segment *d_seg, uint64_t seg_cap, mt_merge_result *d_res,
                               void *stream) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  if (n_streams == 0) return MT_OK;
  if (!d_pts || !d_stream_off || !d_mp || !d_ts || !d_res || (seg_cap && !d_seg))
    return fail(MT_ERR_INVALID, "NULL device pointer");
  HIP_TRY(hipSetDevice(c->device));
  mtgpu::MergeLaunch L;
  L.flags = d_flags;
  L.pts = d_pts;
  L.stream_off = reinterpret_cast<const unsigned long long *>(d_stream_off);
  L.n_frames_total = ~0ull;   // stream_off is trusted to lie inside the caller's arrays
  L.mp = d_mp;
  L.job_semantics = job_semantics;
  L.ts_ws = d_ts;
  L.seg = d_seg;
  L.seg_cap = seg_cap;
  L.res = d_res;
  L.n_streams = n_streams;
  L.stream = static_cast<hipStream_t>(stream);
  hipError_t e = mtgpu::launch_merge(L);
  if (e != hipSuccess) return hip_fail(e, "merge launch");
  return MT_OK;
}

}  // extern "C" (continued below)

namespace {

// One stream's pooled timestamps, device-resident: small lists go to the one-workgroup stream
// kernel, large ones (>= merge_large_min) to the multi-workgroup path.  `ws` must hold
// merge_ts_ws_bytes(n); d_mp is ONE parameter block on the device.  Asynchronous on st.
size_t merge_ts_ws_bytes(const mtgpu_ctx *c, uint64_t n) {
  if (n >= c->merge_large_min && n > 0) return mtgpu::merge_large_ws_bytes(n);
  return sizeof(double) * 2 * (size_t)n + 64;          // [off[2] | pad] [ws 2n]
}

int merge_ts_on(mtgpu_ctx *c, const double *d_ts, uint64_t n, const mt_merge_params *d_mp, int job_semantics,
                void *ws, mt_segment *d_seg, uint64_t seg_cap, mt_merge_result *d_res, hipStream_t st) {
  if (n >= c->merge_large_min && n > 0) {
    hipError_t e = mtgpu::launch_merge_large(d_ts, n, d_mp, job_semantics, ws, d_seg, seg_cap, d_res, st);
    if (e != hipSuccess) return hip_fail(e, "large merge launch");
    return MT_OK;
  }
  unsigned char *w = static_cast<unsigned char *>(ws);
  mtgpu::MergeLaunch L;
  L.flags = nullptr;
  L.pts = d_ts;
  L.stream_off = nullptr;                               // one stream: [0, n)
  L.n_frames_total = n;
  L.mp = d_mp;
  L.job_semantics = job_semantics;
  L.ts_ws = reinterpret_cast<double *>(w + 64);
  L.seg = d_seg;
  L.seg_cap = seg_cap;
  L.res = d_res;
  L.n_streams = 1;
  L.stream = st;
  hipError_t e = mtgpu::launch_merge(L);
  if (e != hipSuccess) return hip_fail(e, "merge launch");
  return MT_OK;
}

}  // namespace

extern "C" {

int mtgpu_merge_timestamps_device(mtgpu_ctx *c, const double *d_ts, uint64_t n, const mt_merge_params *mp,
                                  int job_semantics, mt_segment *d_seg, uint64_t seg_cap,
                                  mt_merge_result *d_res, void *stream) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  if (!mp || !d_res) return fail(MT_ERR_INVALID, "mp/res is NULL");
  if (n > 0 && !d_ts) return fail(MT_ERR_INVALID, "ts is NULL with n > 0");
  if (seg_cap > 0 && !d_seg) return fail(MT_ERR_INVALID, "seg is NULL with seg_cap > 0");
  HIP_TRY(hipSetDevice(c->device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t wsb = merge_ts_ws_bytes(c, n) + 64;               // + the parameter block
  void *scratch = nullptr;
  int slot = -1;
  {
    const int rc0 = scratch_acquire(c, wsb, st, &slot, &scratch);
    if (rc0 != MT_OK) return rc0;
  }
  unsigned char *w = static_cast<unsigned char *>(scratch);
  int rc = MT_OK;
  hipError_t e = mtgpu::launch_store_params(*mp, reinterpret_cast<mt_merge_params *>(w), st);   // by value: captured now
  if (e != hipSuccess) rc = hip_fail(e, "merge params launch");
  if (rc == MT_OK)
    rc = merge_ts_on(c, d_ts, n, reinterpret_cast<const mt_merge_params *>(w), job_semantics, w + 64, d_seg, seg_cap,
                     d_res, st);
  scratch_release(c, slot, st);
  return rc;
}

int mtgpu_merge_segments(mtgpu_ctx *c, const double *ts, uint64_t n, const mt_merge_params *mp,
                         int job_semantics, mt_segment *out, uint64_t cap, mt_merge_result *res) {
  if (!c) return fail(MT_ERR_INVALID, "ctx is NULL");
  if (!mp || !res) return fail(MT_ERR_INVALID, "mp/res is NULL");
  if (n > 0 && !ts) return fail(MT_ERR_INVALID, "ts is NULL with n > 0");
  if (cap > 0 && !out) return fail(MT_ERR_INVALID, "out is NULL with cap > 0");

  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  // one staging block: mp | res | pts[n] | seg[segs] | workspace
  const uint64_t segs = (cap < n ? cap : n) + 1;        // K <= n; +1 for the full-copy segment
  size_t o_mp = 0;
  size_t o_res = o_mp + 64;
  size_t o_pts = o_res + 64;
  size_t o_seg = o_pts + sizeof(double) * (size_t)n;
  size_t o_ws = (o_seg + sizeof(mt_segment) * (size_t)segs + 63) & ~(size_t)63;
  size_t total = o_ws + merge_ts_ws_bytes(c, n);
  int rc = c->d_misc.reserve(total);
  if (rc != MT_OK) return rc;
  unsigned char *d = static_cast<unsigned char *>(c->d_misc.p);
  hipStream_t st = c->stream;
  DrainOnExit drain{st};
  if (n) HIP_TRY(hipMemcpyAsync(d + o_pts, ts, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, st));
  HIP_TRY(hipMemcpyAsync(d + o_mp, mp, sizeof *mp, hipMemcpyHostToDevice, st));
  rc = merge_ts_on(c, reinterpret_cast<const double *>(d + o_pts), n, reinterpret_cast<const mt_merge_params *>(d + o_mp),
                   job_semantics, d + o_ws, reinterpret_cast<mt_segment *>(d + o_seg), segs,
                   reinterpret_cast<mt_merge_result *>(d + o_res), st);
  if (rc != MT_OK) return rc;
  HIP_TRY(hipMemcpyAsync(res, d + o_res, sizeof *res, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (res->status != MT_OK) return fail(res->status, "timestamps contain NaN");
  const uint64_t ncopy = res->n_segments < cap ? res->n_segments : cap;
  if (ncopy) {
    HIP_TRY(hipMemcpyAsync(out, d + o_seg, sizeof(mt_segment) * (size_t)ncopy, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  if (res->n_segments > cap) return fail(MT_ERR_CAPACITY, "need %llu segments, capacity %llu",
                                         (unsigned long long)res->n_segments, (unsigned long long)cap);
  return MT_OK;
}

}  // extern "C"
