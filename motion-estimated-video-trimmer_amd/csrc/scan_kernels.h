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
  int align_lines;         // 40-byte records: peel < 16 head records so that the stream starts on a 128-byte line
  int prefetch;            // compact records, group > 1: issue the next frame's first step before this frame's cluster test
  int xcd_mix;             // workgroup -> work item: the 8 workgroups of an octet (one per XCD) take the octet's 8 items rotated
                           // by a hash of the octet's index, so that no period of empty frames falls on the same XCDs
  int sys_flags;           // flags do not live in device memory (pinned host memory: the pipe's zero-copy staging, a caller's
                           // hipHostMalloc'ed buffer): result bytes leave with system-scope write-through stores
};

struct ScanLaunch {
  const unsigned char *mv;
  unsigned long long n_records;
  const unsigned long long *frame_off;
  const unsigned char *has_sd;
  unsigned int n_frames;
  unsigned char *flags;
  unsigned int *spill_q;        // n_records words (one slot per record), only when k.bands > 1
  unsigned int *slice_ws;       // n_frames * slices * cnt_words words, only when k.slices > 1
  unsigned int *tickets;        // n_frames words (zeroed by launch_scan), only when k.slices > 1
  ScanK k;
  int block;
  int variant;             // experiment knob (MTGPU_VARIANT), 0 = shipped kernel
  unsigned long long item_chunk;  // WORKGROUPS per launch (0 = 2^30; MTGPU_ITEM_CHUNK shrinks it for tests); a workgroup scans k.group items
  int lds_bytes;
  int lds_max;             // device limit of dynamic LDS per workgroup (set once per kernel instantiation)
  int device;
  int rec_bytes;           // 40 = AVMotionVector records, 8 = compact {src_x,src_y,dst_x,dst_y}
  hipStream_t stream;
};

hipError_t launch_scan(const ScanLaunch &L);
// *first_bad (device, pre-set to 0xffffffff) = smallest f with frame_off[f] > frame_off[f + 1]
hipError_t launch_check_offsets(const unsigned long long *frame_off, unsigned int n_frames, unsigned int *first_bad,
                                hipStream_t stream);
// calibration: shape 0 = 16 contiguous bytes per lane, 1 = the scan's 12-of-40-byte records; chunk = bytes per workgroup (0: 1.25 MiB)
hipError_t launch_read_ceiling(const void *p, unsigned long long bytes, int shape, unsigned long long chunk,
                               unsigned int lds_bytes, unsigned int skip, unsigned int *sink, hipStream_t stream);

}  // namespace mtgpu
