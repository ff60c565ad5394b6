/* mtgpu_calib.h — calibration kernels of bench.py (libmtgpu_calib.so).  NOT part of the product or of its C ABI
 * (include/mtgpu.h): a kernel that only reads, in the scan's own load shapes, to state the read rate this box and
 * buffer allow beside the 8 TB/s spec peak. */
#ifndef MTGPU_CALIB_H
#define MTGPU_CALIB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* Stream `bytes` of a device buffer and discard them.  shape 0: 16 contiguous bytes per lane; 1: the scan's own access
 * (bytes 4..15 of every 40-byte record); 2: shape 1 plus the scan's arithmetic on a record that does not vote; 3: shape 2
 * inside the scan's LDS phases (a tile of `lds_bytes` zeroed first, walked once at the end).  chunk: contiguous bytes per
 * 512-thread workgroup (0 = 1.25 MiB).  idle_every > 1: every idle_every-th workgroup leaves at once.  Asynchronous on
 * `stream`.  0 = ok, -1 = error (mtcalib_last_error). */
int mtcalib_read_ceiling(int device, const void *d_buf, uint64_t bytes, int shape, uint64_t chunk, uint32_t lds_bytes,
                         uint32_t idle_every, void *stream);
const char *mtcalib_last_error(void);
#ifdef __cplusplus
}
#endif
#endif
