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
  const unsigned long long *stream_off;   // n_streams + 1
  unsigned long long n_frames_total;
  const mt_merge_params *mp;              // n_streams
  int job_semantics;
  double *ts_ws;                          // 2 * n_frames_total doubles
  mt_segment *seg;                        // n_streams * seg_cap
  unsigned long long seg_cap;
  mt_merge_result *res;                   // n_streams
  unsigned int n_streams;
  hipStream_t stream;
};

hipError_t launch_merge(const MergeLaunch &L);

}  // namespace mtgpu
