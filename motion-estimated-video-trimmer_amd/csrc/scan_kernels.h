// scan_kernels.h — launch interface between the C ABI (mtgpu_api.hip) and the
// gfx950 scan kernels (scan_kernels.hip).  Internal; not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mtgpu {

// Kernel-side parameter block, derived on the host from mt_scan_params
// (reference cfg: include/motion_trim/motion_scanner.hpp:86-93).
struct ScanK {
  unsigned long long thr;  // keep an MV iff |d|^2 >= thr   (src/motion_scanner.cpp:251)
  int shift;               // block_shift                     (:255-256)
  int gw, gh;              // grid_w, grid_h
  int y_lo, y_hi;          // analysed rows [vertical_margin, gh - vertical_margin)  (:237-238)
  unsigned int vec_need;   // vectors_needed                  (:272)
  unsigned int clust_need; // max(1, clusters_needed)         (:288)
  int W;                   // 64-bit words per activity-mask row = ceil(gw / 64)
  int bands;               // row bands per frame (> 1: one workgroup walks them, spilling votes to a queue)
  int band_rows;           // analysed rows per band
  int fb;                  // bits per LDS vote counter: 32, or 1/2/4/8 packed
  int mode;                // 0 ADD32 (fb 32), 1 UNARY thermometer (vec_need <= fb), 2 CAS (fb 8)
  unsigned int active_min; // cell active <=> field value >= active_min
  int chunk_rows;          // centre rows per phase-2 chunk (mask buffer holds chunk_rows + 2 rows)
  int slices;              // workgroups per frame along the record array (1 = none; needs bands == 1)
  int cnt_words;           // LDS counter words per workgroup ((band_rows + 2) * gw fields, padded to 4)
  int mask_rows;           // chunk_rows + 2
  int group;               // consecutive work items per workgroup (>= 1; > 1 only with slices == 1)
  int stage_word;          // group > 1: LDS word (32-byte aligned offset behind the tile) where the workgroup parks the list
                           // entries of its 2nd .. group-th frame: 8 words each, kStageBytes in all
  int align_lines;         // 40-byte records: peel < 16 head records so that the stream starts on a 128-byte line
  int prefetch;            // compact records, group > 1: issue the next frame's first step before this frame's cluster test
  int resident;            // experiments build only: > 0 = that many resident workgroups per CU pull work items with tickets
                           // (one agent-scope atomic per k.group items) instead of one workgroup per k.group items
  int sys_flags;           // flags do not live in device memory (pinned host memory: the pipe's zero-copy staging, a caller's
                           // hipHostMalloc'ed buffer): result bytes leave with system-scope write-through stores
};

// One frame WITH motion-vector side data, as the scan's workgroups see it.  plan_frames (two small kernels ahead of
// every scan) writes the frames that have side data, in stream order, to the front of the launch's work list and
// answers every other frame on the spot (src/motion_scanner.cpp:219-221: no side data -> false): a frame without
// records never costs a workgroup slot, so no period of key frames can leave an XCD without work (the hardware deals
// workgroups to its 8 XCDs in turn).  32 bytes, read with one scalar load per workgroup.
struct WorkItem {
  unsigned long long r0, r1;   // records [r0, r1) of the frame, clamped to the batch (r0 <= r1 <= n_records)
  unsigned int f;              // frame index; kNoFrame: past the last frame with side data
  unsigned int pad[3];
};
constexpr unsigned int kNoFrame = 0xffffffffu;

// Frames per block of the planning kernels (1024 threads, `per` frames each): at most 1024 blocks; up to kPlanFused
// blocks (per == 1 then) the plan is ONE kernel.
constexpr unsigned int kPlanBlock = 1024u;
constexpr unsigned int kPlanFused = 32u;
inline unsigned int plan_per(unsigned int n_frames) {
  const unsigned long long cap = (unsigned long long)kPlanBlock * 1024ull;
  const unsigned long long per = ((unsigned long long)n_frames + cap - 1ull) / cap;
  return per > 0ull ? (unsigned int)per : 1u;
}
inline unsigned int plan_blocks(unsigned int n_frames) {
  const unsigned long long ch = (unsigned long long)kPlanBlock * plan_per(n_frames);
  return (unsigned int)(((unsigned long long)n_frames + ch - 1ull) / ch);
}
// Launch scratch of the plan: the work list (n_frames + 1 items: the last one is always kNoFrame), one count per
// planning block, the ticket word of the resident form.
inline size_t plan_scratch_bytes(unsigned int n_frames) {
  return sizeof(WorkItem) * ((size_t)n_frames + 1u) + sizeof(unsigned int) * ((size_t)plan_blocks(n_frames) + 4u);
}

constexpr int kStageBytes = 64 * 32;     // LDS behind the tile for the parked entries of a grouped workgroup (group <= 64)

struct ScanLaunch {
  const unsigned char *mv;
  unsigned long long n_records;    // frame_off entries are clamped to this (in the caller's units, before `rebase`)
  const unsigned long long *frame_off;
  const unsigned char *has_sd;
  unsigned int n_frames;
  unsigned char *flags;
  unsigned int *spill_q;        // n_records words (one slot per record), only when k.bands > 1
  unsigned int *slice_ws;       // n_frames * slices * cnt_words words, only when k.slices > 1
  unsigned int *tickets;        // n_frames words (zeroed by launch_scan), only when k.slices > 1
  void *plan_ws;                // plan_scratch_bytes(n_frames), 32-byte aligned: work list + planning counts
  unsigned long long rebase;    // records [frame_off[f], frame_off[f + 1]) live at mv + (frame_off[f] - rebase) * rec_bytes
                                // (a host-pointer call copies only the window its offsets span); <= n_records
  int cu_count;
  ScanK k;
  int block;
  int variant;             // experiment knob (MTGPU_VARIANT), 0 = shipped kernel
  unsigned long long item_chunk;  // WORKGROUPS per launch (0 = 2^30; MTGPU_ITEM_CHUNK shrinks it for tests); a workgroup scans k.group items
  int lds_bytes;
  int lds_max;             // device limit of dynamic LDS per workgroup (set once per kernel instantiation)
  int device;
  int rec_bytes;           // 40 = AVMotionVector records, 8 = compact {src_x,src_y,dst_x,dst_y}
  hipStream_t stream;
  hipEvent_t ev_planned;   // profiling (mtgpu_profile_enable): recorded between the planning kernels and the scan kernel; else nullptr
};

hipError_t launch_scan(const ScanLaunch &L);
// *first_bad (device, pre-set to 0xffffffff) = smallest f with frame_off[f] > frame_off[f + 1]
hipError_t launch_check_offsets(const unsigned long long *frame_off, unsigned int n_frames, unsigned int *first_bad,
                                hipStream_t stream);

}  // namespace mtgpu
