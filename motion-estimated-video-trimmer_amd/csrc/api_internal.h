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
// plan_ws / plan_ws_bytes: device memory OWNED BY THE CALLER for the launch's work list (ctx_plan_ws_bytes(frames) bytes,
// 256-byte aligned; not shared with any launch that may be in flight at the same time), or nullptr / 0: stream-ordered
// scratch from the context's pool.  A pipe gives every staging batch its own: batches are submitted from many threads
// on a handful of shared streams, and their launches then allocate nothing.
int ctx_launch_scan(mtgpu_ctx *c, const void *d_mv, uint64_t n_records, const uint64_t *d_off,
                    const uint8_t *d_sd, uint32_t n_frames, uint8_t *d_flags, hipStream_t st, int rec_bytes,
                    int flags_in_host_memory, void *plan_ws, size_t plan_ws_bytes);
size_t ctx_plan_ws_bytes(uint32_t n_frames);

}  // namespace mtgpu
