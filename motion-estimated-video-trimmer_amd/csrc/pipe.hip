// pipe.hip — host dispatcher of the C ABI (include/mtgpu.h, "Host dispatcher"):
// pinned multi-buffered staging, asynchronous H2D + scan + D2H per batch.
// Replaces the synchronous check_frame call in the decode loop
// (reference src/motion_scanner.cpp:375-383) and the copy-out that the MV side
// data's lifetime (:347) forces on any batched backend.
//
// Staging layout: MT_LAYOUT_COMPACT8 (default, with MT_LAYOUT_ZERO_COPY) copies only bytes 6..13 of every 40-byte
// AVMotionVector (src_x, src_y, dst_x, dst_y — all check_frame reads) into pinned memory, so
// 8 instead of 40 bytes per record cross PCIe and the scan reads the 8-byte records
// (scan_kernels.hip, REC 8); MT_LAYOUT_AOS40 stages the records unchanged.
//
// MT_LAYOUT_ZERO_COPY (either record layout): no device mirror and no copy commands at all — the
// scan kernel streams the pinned block over PCIe itself (every record is read exactly once by
// every plan, banded ones included, since bands replay a device-side queue) and writes the flag
// bytes straight into pinned memory; a submit is then one kernel launch + one event record.
//
// Batch states: 0 free -> (acquire) 1 filling -> (submit) 2 in flight -> (collect) 3 collected
// -> (release) 0.  A failed submit drains the batch's stream and leaves it in state 1 with its
// contents intact (retry or release).  A failed collect (the wait on the batch's event failed)
// first drains the batch's STREAM — with zero-copy staging the kernel reads h_stage and writes
// h_flags over PCIe, so the pinned block must not be refilled or freed while it may still run —
// and then hands the batch out (state 3) so that it can be released.  If even that drain fails
// the batch is POISONED (state 4): release accepts it, acquire never hands it out again, and
// only mtgpu_pipe_destroy touches its memory.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <vector>

#include "api_internal.h"
#include "pack_simd.h"

struct mtgpu_batch {
  // pinned host staging.  Offsets, has_sd bytes and records share ONE block
  // ([off (cap_frames+1) x 8 | sd cap_frames | pad to 64 | records]) mirrored by one device
  // block, so a batch goes up with a single H2D copy: with many decoder threads submitting
  // concurrently the runtime's per-call cost, not PCIe, is what a submit pays for.
  unsigned char *h_stage = nullptr;
  unsigned char *h_mv = nullptr;     // = h_stage + hdr_bytes
  uint64_t *h_off = nullptr;         // = h_stage
  uint8_t *h_sd = nullptr;           // = h_stage + (cap_frames + 1) * 8
  double *h_pts = nullptr;
  uint64_t *h_tag = nullptr;
  uint8_t *h_flags = nullptr;
  // device mirrors
  unsigned char *d_stage = nullptr;
  unsigned char *d_mv = nullptr;
  uint64_t *d_off = nullptr;
  uint8_t *d_sd = nullptr;
  uint8_t *d_flags = nullptr;
  size_t hdr_bytes = 0;
  size_t stage_bytes = 0;             // size of h_stage (and of d_stage when it exists)
  size_t aux_bytes = 0;               // pts + tag + flag arrays (pinned; flags mirrored on the device without zero-copy)
  bool own_stage = false;             // h_stage / d_stage are this batch's own blocks (grown for an oversize frame),
                                      // not slices of the pipe's slabs
  uint64_t cap_records = 0, n_records = 0;
  uint32_t cap_frames = 0, n_frames = 0;
  int rec_bytes = MT_COMPACT_BYTES;   // bytes per staged record: 8 (compact) or 40 (AoS)
  bool zero_copy = false;             // the scan reads the pinned staging (and writes the flags) over PCIe itself
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;
  int state = 0;   // 0 free, 1 filling, 2 in flight, 3 collected, 4 poisoned (never reused)
  mtgpu_pipe *owner = nullptr;
};

struct mtgpu_pipe {
  mtgpu_ctx *ctx = nullptr;
  // ONE pinned slab (and, without zero-copy, one device slab) holds the staging of every batch: a pipe is set
  // up with two allocations instead of six per batch — with 64 x T worker threads creating their pipes at
  // once the driver serialises those calls, and 768 of them took ~0.7 s per worker (profiles/r03_host_batch64.json)
  unsigned char *h_slab = nullptr, *d_slab = nullptr;
  size_t slab_bytes = 0, d_slab_bytes = 0;
  int rec_bytes = MT_COMPACT_BYTES;
  bool zero_copy = false;
  bool blocking_events = false;  // MTGPU_EVENT_BLOCKING=1: collect sleeps on the batch's event instead of polling it
  long inject_submit_fail = 0;   // MTGPU_INJECT_SUBMIT_FAIL=k (tests): the k-th submit fails after its copies were queued
  long inject_collect_fail = 0;  // MTGPU_INJECT_COLLECT_FAIL=k (tests): the k-th collect's event wait "fails";
                                 // negative: its stream drain "fails" as well (the batch is poisoned)
  long collects = 0;
  long submits = 0;
  bool inject_grow_fail = false; // MTGPU_INJECT_GROW_FAIL=1 (tests): growing a batch for an oversize frame fails
  std::vector<mtgpu_batch *> bufs;
  std::deque<mtgpu_batch *> inflight;
  std::mutex mu;
};

namespace {

using mtgpu::fail;
using mtgpu::hip_fail;

void free_batch(mtgpu_batch *b) {
  if (!b) return;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->own_stage) {
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    if (b->d_stage) (void)hipFree(b->d_stage);
  }
  if (b->done) (void)hipEventDestroy(b->done);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

size_t stage_bytes_for(uint32_t cap_frames, uint64_t records, int rec_bytes, size_t *hdr_out) {
  const size_t nf = (size_t)cap_frames;
  const size_t hdr = (sizeof(uint64_t) * (nf + 1) + (nf + 1) + 63u) & ~(size_t)63u;
  if (hdr_out) *hdr_out = hdr;
  return (hdr + (size_t)records * (size_t)rec_bytes + 64 + 63u) & ~(size_t)63u;
}

size_t aux_bytes_for(uint32_t cap_frames) {          // [pts (nf+1) x 8 | tags (nf+1) x 8 | flags nf+1], 64-byte aligned
  const size_t nf = (size_t)cap_frames + 1;
  return (nf * (sizeof(double) + sizeof(uint64_t)) + nf + 63u) & ~(size_t)63u;
}

// Point a batch's staging at `h` (pinned) / `dv` (what the kernel dereferences: the device view of the pinned
// block with zero-copy, device memory otherwise) / `d_own` (the device block to free later, or nullptr).
void set_stage(mtgpu_batch *b, unsigned char *h, unsigned char *dv, unsigned char *d_own, size_t hdr, size_t bytes,
               uint64_t records) {
  const size_t nf = (size_t)b->cap_frames;
  b->h_stage = h;
  b->d_stage = d_own;
  b->hdr_bytes = hdr;
  b->stage_bytes = bytes;
  b->h_off = reinterpret_cast<uint64_t *>(h);
  b->h_sd = h + sizeof(uint64_t) * (nf + 1);
  b->h_mv = h + hdr;
  b->d_off = reinterpret_cast<uint64_t *>(dv);
  b->d_sd = dv + sizeof(uint64_t) * (nf + 1);
  b->d_mv = dv + hdr;
  b->h_off[0] = 0;
  b->cap_records = records;
}

#define PIPE_TRY(expr)                                               \
  do {                                                               \
    hipError_t _e = (expr);                                          \
    if (_e != hipSuccess) { rc = hip_fail(_e, #expr); goto bad; }    \
  } while (0)

// Give a batch its OWN, larger staging block for `records` records (a frame larger than a whole batch); the
// batch must be idle and empty.  The new blocks are allocated BEFORE the old ones are let go: when that fails
// the batch keeps its previous staging and capacity, so it stays usable.  Its slice of the pipe's slab (if it
// still used one) simply stays unused from then on.
int alloc_records(mtgpu_batch *b, uint64_t records, bool inject_failure = false) {
  int rc = MT_OK;
  unsigned char *h_new = nullptr, *d_new = nullptr, *dev_view = nullptr;
  size_t hdr = 0;
  const size_t bytes = stage_bytes_for(b->cap_frames, records, b->rec_bytes, &hdr);
  PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_new), bytes, hipHostMallocDefault));
  if (inject_failure) { rc = fail(MT_ERR_NOMEM, "injected allocation failure (MTGPU_INJECT_GROW_FAIL)"); goto bad; }
  if (b->zero_copy) {
    // no device mirror: the kernel reads the pinned block through its device-visible address
    PIPE_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev_view), h_new, 0));
  } else {
    PIPE_TRY(hipMalloc(reinterpret_cast<void **>(&d_new), bytes));
    dev_view = d_new;
  }
  if (b->own_stage) {
    if (b->h_stage) (void)hipHostFree(b->h_stage);
    if (b->d_stage) (void)hipFree(b->d_stage);
  }
  b->own_stage = true;
  set_stage(b, h_new, dev_view, d_new, hdr, bytes, records);
  return MT_OK;
bad:
  if (h_new) (void)hipHostFree(h_new);
  if (d_new) (void)hipFree(d_new);
  return rc;
}

// Batch i of a pipe: staging and result arrays are slices of the pipe's slabs at `off` (64-byte aligned).
int alloc_batch(mtgpu_batch **out, mtgpu_pipe *p, size_t off, unsigned char *slab_dev_view, uint64_t max_records,
                uint32_t max_frames) {
  int rc = MT_OK;
  mtgpu_batch *b = new (std::nothrow) mtgpu_batch();
  if (!b) return fail(MT_ERR_NOMEM, "out of host memory");
  b->cap_frames = max_frames;
  b->rec_bytes = p->rec_bytes;
  b->zero_copy = p->zero_copy;
  {
    size_t hdr = 0;
    const size_t sbytes = stage_bytes_for(max_frames, max_records, p->rec_bytes, &hdr);
    const size_t nf = (size_t)max_frames + 1;
    set_stage(b, p->h_slab + off, slab_dev_view + off, nullptr, hdr, sbytes, max_records);
    unsigned char *aux = p->h_slab + off + sbytes;
    b->aux_bytes = aux_bytes_for(max_frames);
    b->h_pts = reinterpret_cast<double *>(aux);
    b->h_tag = reinterpret_cast<uint64_t *>(aux + nf * sizeof(double));
    b->h_flags = aux + nf * (sizeof(double) + sizeof(uint64_t));
    // zero-copy: the kernel writes the flag bytes straight into the pinned slab; otherwise into the device slab
    b->d_flags = slab_dev_view + off + sbytes + nf * (sizeof(double) + sizeof(uint64_t));
    PIPE_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    // system-scope release at the event: the flag bytes a zero-copy scan wrote into pinned memory
    // are visible to the host thread that waits on it
    PIPE_TRY(hipEventCreateWithFlags(&b->done, hipEventDisableTiming | hipEventReleaseToSystem |
                                                   (p->blocking_events ? hipEventBlockingSync : 0u)));
  }
  *out = b;
  return MT_OK;
bad:
  free_batch(b);
  return rc;
}

// hipSetDevice only when this thread's current device differs (a submit per few hundred
// microseconds from each of many threads: every runtime call counts).
hipError_t use_device(int dev) {
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur == dev) return hipSuccess;
  return hipSetDevice(dev);
}

}  // namespace

extern "C" {

int mtgpu_pipe_create(mtgpu_ctx *ctx, uint64_t max_records_per_batch, uint32_t max_frames_per_batch,
                      int n_buffers, mtgpu_pipe **out) {
  return mtgpu_pipe_create_layout(ctx, max_records_per_batch, max_frames_per_batch, n_buffers,
                                  MT_LAYOUT_COMPACT8 | MT_LAYOUT_ZERO_COPY, out);
}

int mtgpu_pipe_create_layout(mtgpu_ctx *ctx, uint64_t max_records_per_batch, uint32_t max_frames_per_batch,
                             int n_buffers, int layout, mtgpu_pipe **out) {
  if (!ctx || !out) return fail(MT_ERR_INVALID, "NULL argument");
  *out = nullptr;
  if (layout < 0 || layout > (MT_LAYOUT_AOS40 | MT_LAYOUT_ZERO_COPY))
    return fail(MT_ERR_INVALID, "layout must be MT_LAYOUT_COMPACT8 or MT_LAYOUT_AOS40, optionally | MT_LAYOUT_ZERO_COPY");
  if (max_records_per_batch == 0 || max_frames_per_batch == 0 || n_buffers < 1 || n_buffers > 64)
    return fail(MT_ERR_INVALID, "pipe needs max_records > 0, max_frames > 0, 1 <= n_buffers <= 64");
  hipError_t e = hipSetDevice(mtgpu::ctx_device(ctx));
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  mtgpu_pipe *p = new (std::nothrow) mtgpu_pipe();
  if (!p) return fail(MT_ERR_NOMEM, "out of host memory");
  p->ctx = ctx;
  p->rec_bytes = (layout & MT_LAYOUT_AOS40) ? MT_MV_BYTES : MT_COMPACT_BYTES;
  p->zero_copy = (layout & MT_LAYOUT_ZERO_COPY) != 0;
  if (const char *v = std::getenv("MTGPU_EVENT_BLOCKING")) p->blocking_events = std::atol(v) != 0;
  if (const char *v = std::getenv("MTGPU_INJECT_SUBMIT_FAIL")) p->inject_submit_fail = std::atol(v);
  if (const char *v = std::getenv("MTGPU_INJECT_GROW_FAIL")) p->inject_grow_fail = std::atol(v) != 0;
  if (const char *v = std::getenv("MTGPU_INJECT_COLLECT_FAIL")) p->inject_collect_fail = std::atol(v);
  {
    const size_t per = stage_bytes_for(max_frames_per_batch, max_records_per_batch, p->rec_bytes, nullptr) +
                       aux_bytes_for(max_frames_per_batch);
    p->slab_bytes = per * (size_t)n_buffers;
    unsigned char *dev_view = nullptr;
    e = hipHostMalloc(reinterpret_cast<void **>(&p->h_slab), p->slab_bytes, hipHostMallocDefault);
    if (e != hipSuccess) { p->h_slab = nullptr; mtgpu_pipe_destroy(p); return hip_fail(e, "hipHostMalloc(pipe staging)"); }
    if (p->zero_copy) {
      e = hipHostGetDevicePointer(reinterpret_cast<void **>(&dev_view), p->h_slab, 0);
      if (e != hipSuccess) { mtgpu_pipe_destroy(p); return hip_fail(e, "hipHostGetDevicePointer"); }
    } else {
      e = hipMalloc(reinterpret_cast<void **>(&p->d_slab), p->slab_bytes);
      if (e != hipSuccess) { p->d_slab = nullptr; mtgpu_pipe_destroy(p); return hip_fail(e, "hipMalloc(pipe staging mirror)"); }
      p->d_slab_bytes = p->slab_bytes;
      dev_view = p->d_slab;
    }
    for (int i = 0; i < n_buffers; ++i) {
      mtgpu_batch *b = nullptr;
      int rc = alloc_batch(&b, p, per * (size_t)i, dev_view, max_records_per_batch, max_frames_per_batch);
      if (rc != MT_OK) {
        char keep[512];
        std::snprintf(keep, sizeof keep, "%s", mtgpu_last_error());
        mtgpu_pipe_destroy(p);
        return fail(rc, "%s", keep);
      }
      b->owner = p;
      p->bufs.push_back(b);
    }
  }
  *out = p;
  return MT_OK;
}

void mtgpu_pipe_destroy(mtgpu_pipe *p) {
  if (!p) return;
  (void)hipSetDevice(mtgpu::ctx_device(p->ctx));
  for (mtgpu_batch *b : p->bufs) free_batch(b);      // drains every batch's stream first
  if (p->h_slab) (void)hipHostFree(p->h_slab);
  if (p->d_slab) (void)hipFree(p->d_slab);
  delete p;
}

int mtgpu_pipe_acquire(mtgpu_pipe *p, mtgpu_batch **out) {
  if (!p || !out) return fail(MT_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> lock(p->mu);
  for (mtgpu_batch *b : p->bufs)
    if (b->state == 0) {
      b->state = 1;
      b->n_frames = 0;
      b->n_records = 0;
      b->h_off[0] = 0;
      *out = b;
      return MT_OK;
    }
  *out = nullptr;
  size_t retired = 0;
  for (const mtgpu_batch *b : p->bufs) retired += b->state == 4;
  if (retired == p->bufs.size())
    return fail(MT_ERR_DEVICE, "every staging batch of this pipe was retired after a failed collect: destroy the pipe");
  return fail(MT_ERR_BUSY, "all %zu staging batches are in flight or held: collect and release one", p->bufs.size());
}

int mtgpu_batch_add_frame(mtgpu_batch *b, const void *mv_bytes, uint64_t n_bytes, int has_side_data,
                          double pts, uint64_t tag) {
  if (!b) return fail(MT_ERR_INVALID, "batch is NULL");
  if (b->state != 1) return fail(MT_ERR_INVALID, "batch is not being filled (acquire it first)");
  const uint64_t n = (mv_bytes && has_side_data) ? n_bytes / MT_MV_BYTES : 0;   // :226 integer division
  if (b->n_frames >= b->cap_frames || b->n_records + n > b->cap_records) {
    if (b->n_frames != 0 || n <= b->cap_records) return fail(MT_ERR_CAPACITY, "batch full");
    // One frame larger than a whole batch (check_frame accepts any count, :226): grow this
    // batch's staging.  The batch is empty and idle (its previous work was collected), so its
    // buffers can be replaced.
    hipError_t e = hipSetDevice(mtgpu::ctx_device(b->owner->ctx));
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    int rc = alloc_records(b, n + n / 4, b->owner->inject_grow_fail);
    if (rc != MT_OK) return rc;          // the batch keeps its previous staging and stays usable
  }
  if (n) {
    unsigned char *dst = b->h_mv + (size_t)b->n_records * (size_t)b->rec_bytes;
    if (b->rec_bytes == MT_MV_BYTES) std::memcpy(dst, mv_bytes, (size_t)n * MT_MV_BYTES);
    else mtgpu::pack_records(static_cast<const unsigned char *>(mv_bytes), n, dst);   // bytes 6..13 only
  }
  const uint32_t f = b->n_frames;
  b->n_records += n;
  b->h_off[f + 1] = b->n_records;
  b->h_sd[f] = has_side_data ? 1 : 0;
  b->h_pts[f] = pts;
  b->h_tag[f] = tag;
  b->n_frames = f + 1;
  return MT_OK;
}

uint32_t mtgpu_batch_frames(const mtgpu_batch *b) { return b ? b->n_frames : 0; }

int mtgpu_pipe_submit(mtgpu_pipe *p, mtgpu_batch *b) {
  if (!p || !b || b->owner != p) return fail(MT_ERR_INVALID, "batch does not belong to this pipe");
  if (b->state != 1) return fail(MT_ERR_INVALID, "batch is not being filled");
  hipError_t e = use_device(mtgpu::ctx_device(p->ctx));
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t st = b->stream;
  int rc = MT_OK;
  long nth;
  { std::lock_guard<std::mutex> lock(p->mu); nth = ++p->submits; }
  if (b->n_frames) {
    // one copy: offsets + has_sd header and the records that follow it (zero-copy: none at all)
    if (!b->zero_copy)
      PIPE_TRY(hipMemcpyAsync(b->d_off, b->h_stage, b->hdr_bytes + (size_t)b->n_records * (size_t)b->rec_bytes,
                              hipMemcpyHostToDevice, st));
    if (p->inject_submit_fail > 0 && nth == p->inject_submit_fail) {
      rc = fail(MT_ERR_DEVICE, "injected submit failure (MTGPU_INJECT_SUBMIT_FAIL)");
      goto bad;
    }
    rc = mtgpu::ctx_launch_scan(p->ctx, b->d_mv, b->n_records, b->d_off, b->d_sd, b->n_frames, b->d_flags, st,
                                b->rec_bytes);
    if (rc != MT_OK) goto bad;
    if (!b->zero_copy)
      PIPE_TRY(hipMemcpyAsync(b->h_flags, b->d_flags, b->n_frames, hipMemcpyDeviceToHost, st));
  }
  PIPE_TRY(hipEventRecord(b->done, st));
  {
    std::lock_guard<std::mutex> lock(p->mu);
    b->state = 2;
    p->inflight.push_back(b);
  }
  return MT_OK;
bad:
  // Copies / kernels of this batch may already be queued: let them finish before the caller can
  // touch the pinned staging again.  The batch stays in state 1 (contents intact).
  {
    char keep[512];
    std::snprintf(keep, sizeof keep, "%s", mtgpu_last_error());
    (void)hipStreamSynchronize(st);
    (void)fail(rc, "%s", keep);
  }
  return rc;
}

int mtgpu_pipe_collect(mtgpu_pipe *p, mtgpu_batch **out, const uint8_t **flags, const double **pts,
                       const uint64_t **tags, uint32_t *n_frames) {
  if (!p || !out) return fail(MT_ERR_INVALID, "NULL argument");
  mtgpu_batch *b = nullptr;
  {
    std::lock_guard<std::mutex> lock(p->mu);
    if (p->inflight.empty()) return fail(MT_ERR_INVALID, "no batch in flight");
    b = p->inflight.front();
    p->inflight.pop_front();
    b->state = 3;
  }
  long nth;
  { std::lock_guard<std::mutex> lock(p->mu); nth = ++p->collects; }
  *out = b;                       // handed out even on failure, so that it can be released
  hipError_t e = hipEventSynchronize(b->done);
  const bool inject = p->inject_collect_fail != 0 && nth == std::labs(p->inject_collect_fail);
  if (inject && e == hipSuccess) e = hipErrorUnknown;
  if (e != hipSuccess) {
    // The kernel of this batch may still be running: nobody may refill (or free) its pinned staging
    // until the batch's stream has drained.  If the drain fails too, the batch is never reused.
    hipError_t e2 = hipStreamSynchronize(b->stream);
    if (inject) e2 = p->inject_collect_fail < 0 ? hipErrorUnknown : e2;
    if (e2 != hipSuccess) {
      std::lock_guard<std::mutex> lock(p->mu);
      b->state = 4;
      return fail(MT_ERR_DEVICE, "hipEventSynchronize: %s; draining the batch's stream failed too (%s): "
                  "the batch is retired", hipGetErrorString(e), hipGetErrorString(e2));
    }
    return hip_fail(e, inject ? "hipEventSynchronize (injected, MTGPU_INJECT_COLLECT_FAIL)" : "hipEventSynchronize");
  }
  if (flags) *flags = b->h_flags;
  if (pts) *pts = b->h_pts;
  if (tags) *tags = b->h_tag;
  if (n_frames) *n_frames = b->n_frames;
  return MT_OK;
}

int mtgpu_pipe_get_stats(mtgpu_pipe *p, mtgpu_pipe_stats *out) {
  if (!p || !out) return fail(MT_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> lock(p->mu);
  std::memset(out, 0, sizeof *out);
  out->pinned_bytes = p->slab_bytes;
  out->device_bytes = p->d_slab_bytes;
  for (const mtgpu_batch *b : p->bufs)
    if (b->own_stage) {                              // grown for an oversize frame: its own blocks on top of the slabs
      out->pinned_bytes += b->stage_bytes;
      if (!b->zero_copy) out->device_bytes += b->stage_bytes;
    }
  out->submits = (uint64_t)p->submits;
  out->n_buffers = (uint32_t)p->bufs.size();
  out->layout = (p->rec_bytes == MT_MV_BYTES ? MT_LAYOUT_AOS40 : MT_LAYOUT_COMPACT8) | (p->zero_copy ? MT_LAYOUT_ZERO_COPY : 0);
  return MT_OK;
}

int mtgpu_pipe_release(mtgpu_pipe *p, mtgpu_batch *b) {
  if (!p || !b || b->owner != p) return fail(MT_ERR_INVALID, "batch does not belong to this pipe");
  std::lock_guard<std::mutex> lock(p->mu);
  if (b->state == 4) return MT_OK;        // poisoned by a failed collect: stays out of circulation
  if (b->state != 3 && b->state != 1) return fail(MT_ERR_INVALID, "batch is in flight");
  b->state = 0;
  return MT_OK;
}

}  // extern "C"
