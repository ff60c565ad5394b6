// merge_kernels.h — launch interface for the stream merge kernel
// (merge_kernels.hip).  Internal; not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mt_types.h"

namespace mtgpu {

struct MergeLaunch {
  const unsigned char *flags;             // n_frames_total bytes or NULL (= every frame flagged)
  const double *pts;                      // n_frames_total
  const unsigned long long *stream_off;   // n_streams + 1; NULL (n_streams == 1): the one stream is [0, n_frames_total)
  unsigned long long n_frames_total;
  const mt_merge_params *mp;              // n_streams
  int job_semantics;
  double *ts_ws;                          // 2 * n_frames_total doubles (include/mtgpu.h: d_ts)
  mt_segment *seg;                        // n_streams * seg_cap
  unsigned long long seg_cap;
  mt_merge_result *res;                   // n_streams
  unsigned int n_streams;
  hipStream_t stream;
};

hipError_t launch_merge(const MergeLaunch &L);

// *d_dst = v on `stream` (v is captured at launch time).
hipError_t launch_store_params(const mt_merge_params &v, mt_merge_params *d_dst, hipStream_t stream);

// Multi-workgroup merge of ONE stream's pooled timestamps (any order, duplicates allowed), n >= 1:
// device-wide sort (LDS tile sort + merge-path passes), then unique / gap merge / clamp / savings
// with the same arithmetic as merge_streams_kernel.  ws: merge_large_ws_bytes(n) bytes.
size_t merge_large_ws_bytes(unsigned long long n);
hipError_t launch_merge_large(const double *d_ts, unsigned long long n, const mt_merge_params *d_mp, int job_semantics,
                              void *ws, mt_segment *d_seg, unsigned long long seg_cap, mt_merge_result *d_res,
                              hipStream_t st);

}  // namespace mtgpu
