// api_internal.h — helpers shared by the translation units that implement the C ABI
// (mtgpu_api.hip, pipe.hip).  Internal.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mtgpu.h"

namespace mtgpu {

int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int hip_fail(hipError_t e, const char *what);

int ctx_device(const mtgpu_ctx *c);
// a pooled stream of the context for one staging batch (nullptr: create your own)
hipStream_t ctx_pipe_stream(mtgpu_ctx *c);
// logical -> physical HIP device (identity unless MTGPU_ALIAS_DEVICES presents more logical devices than exist)
int physical_device(int logical);
// Launch the scan for a device-resident batch on `st`.
// rec_bytes: MT_MV_BYTES (AVMotionVector records) or MT_COMPACT_BYTES (packed src/dst fields).
// flags_in_host_memory: d_flags is pinned host memory (zero-copy staging): result bytes are stored at system scope.
int ctx_launch_scan(mtgpu_ctx *c, const void *d_mv, uint64_t n_records, const uint64_t *d_off,
                    const uint8_t *d_sd, uint32_t n_frames, uint8_t *d_flags, hipStream_t st, int rec_bytes,
                    int flags_in_host_memory);

}  // namespace mtgpu
