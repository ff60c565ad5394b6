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

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <vector>

#include "api_internal.h"
#include "knobs.h"
#include "pack_simd.h"

struct mtgpu_batch {
  // pinned host staging: ONE block per batch,
  //   [off (cap_frames+1) x 8 | sd cap_frames | pad to 64 | records | pad to 128 | pts | tags | pad to 128 | flags | pad to 128]
  // The flag bytes — the only part of the block the DEVICE writes — own their 128-byte lines: no line holds both
  // bytes the host stored (pts, tags, records) and bytes the device stores.  x86 keeps such a shared line coherent
  // for snooped PCIe writes, so this is not a fix of a known fault; it removes the one place where the block's
  // correctness leaned on how a partial device write merges into a line the host holds modified (round 4's
  // unexplained wrong flag with hipHostRegister'ed staging).
  // so a batch goes up with a single H2D copy of its head (without zero-copy): with many decoder threads
  // submitting concurrently the runtime's per-call cost, not PCIe, is what a submit pays for.  The block is
  // pinned on the batch's FIRST use (mtgpu_pipe_acquire), not at pipe creation: page-locking costs ~0.2 ms per
  // MiB whatever the block size and is serialised across threads by the driver (profiles/r04_pin_probe.json:
  // 16 MiB 3.6 ms, 3 GiB 691 ms, 64 threads x 48 MiB at once 622 ms), so a worker that pins only what it is
  // about to fill starts three times sooner and the rest of the pinning overlaps the scan.
  unsigned char *h_stage = nullptr;  // the block (nullptr: not pinned yet)
  unsigned char *h_mv = nullptr;     // = h_stage + hdr_bytes
  uint64_t *h_off = nullptr;         // = h_stage
  uint8_t *h_sd = nullptr;           // = h_stage + (cap_frames + 1) * 8
  double *h_pts = nullptr;
  uint64_t *h_tag = nullptr;
  uint8_t *h_flags = nullptr;
  // what the kernel dereferences: the device view of the pinned block (zero-copy) or a device mirror
  unsigned char *d_stage = nullptr;  // the mirror to free later (nullptr with zero-copy)
  void *d_plan = nullptr;            // device memory for this batch's work list (ctx_plan_ws_bytes(cap_frames)): its scans
  size_t plan_bytes = 0;             // allocate nothing — a batch has at most one launch in flight
  unsigned char *d_mv = nullptr;
  uint64_t *d_off = nullptr;
  uint8_t *d_sd = nullptr;
  uint8_t *d_flags = nullptr;
  size_t hdr_bytes = 0;
  size_t stage_bytes = 0;             // [off | sd | records] part of the block
  size_t block_bytes = 0;             // the whole block: stage + pts / tag / flag arrays
  uint64_t cap_records = 0, n_records = 0;
  uint64_t want_records = 0;          // capacity the block gets when it is pinned
  uint32_t cap_frames = 0, n_frames = 0;
  int rec_bytes = MT_COMPACT_BYTES;   // bytes per staged record: 8 (compact) or 40 (AoS)
  bool zero_copy = false;             // the scan reads the pinned staging (and writes the flags) over PCIe itself
  hipStream_t stream = nullptr;       // from the context's pool (shared with other batches / pipes) or this batch's own
  bool own_stream = false;
  hipEvent_t done = nullptr;
  int state = 0;   // 0 free, 1 filling, 2 in flight, 3 collected, 4 poisoned (never reused)
  mtgpu_pipe *owner = nullptr;
};

struct mtgpu_pipe {
  mtgpu_ctx *ctx = nullptr;
  int rec_bytes = MT_COMPACT_BYTES;
  bool zero_copy = false;
  bool blocking_events = false;  // MTGPU_EVENT_BLOCKING=1: collect sleeps on the batch's event instead of polling it
  bool eager_pin = false;        // MTGPU_PIPE_EAGER=1: pin every batch at creation (round 3 behaviour)
  long inject_submit_fail = 0;   // MTGPU_INJECT_SUBMIT_FAIL=k (tests): the k-th submit fails after its copies were queued
  long inject_collect_fail = 0;  // MTGPU_INJECT_COLLECT_FAIL=k (tests): the k-th collect's event wait "fails";
                                 // negative: its stream drain "fails" as well (the batch is poisoned)
  bool inject_once = false;      // MTGPU_INJECT_ONCE=1 (tests): an injected collect failure fires once per PROCESS, not per pipe
  long collects = 0;
  long submits = 0;
  bool inject_grow_fail = false; // MTGPU_INJECT_GROW_FAIL=1 (tests): growing a batch for an oversize frame fails
  std::atomic<uint64_t> pin_us{0};   // time spent page-locking staging blocks (creation + first uses + growth); add_frame
                                     // grows a batch without the pipe lock, get_stats reads under it: atomic
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
  if (b->h_stage) (void)hipHostFree(b->h_stage);
  if (b->d_stage) (void)hipFree(b->d_stage);
  if (b->d_plan) (void)hipFree(b->d_plan);
  if (b->done) (void)hipEventDestroy(b->done);
  if (b->stream && b->own_stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

size_t stage_bytes_for(uint32_t cap_frames, uint64_t records, int rec_bytes, size_t *hdr_out) {
  const size_t nf = (size_t)cap_frames;
  const size_t hdr = (sizeof(uint64_t) * (nf + 1) + (nf + 1) + 63u) & ~(size_t)63u;
  if (hdr_out) *hdr_out = hdr;
  return (hdr + (size_t)records * (size_t)rec_bytes + 64 + 127u) & ~(size_t)127u;
}

// [pts (nf+1) x 8 | tags (nf+1) x 8 | pad to 128 | flags nf+1 | pad to 128]; *flags_off = offset of the flags
size_t aux_bytes_for(uint32_t cap_frames, size_t *flags_off = nullptr) {
  const size_t nf = (size_t)cap_frames + 1;
  const size_t fo = (nf * (sizeof(double) + sizeof(uint64_t)) + 127u) & ~(size_t)127u;
  if (flags_off) *flags_off = fo;
  return fo + ((nf + 127u) & ~(size_t)127u);
}

#define PIPE_TRY(expr)                                               \
  do {                                                               \
    hipError_t _e = (expr);                                          \
    if (_e != hipSuccess) { rc = hip_fail(_e, #expr); goto bad; }    \
  } while (0)

// Pin (or re-pin, larger) a batch's staging block for `records` records.  The batch must be idle and empty.
// The new blocks are allocated BEFORE the old ones are let go: when that fails the batch keeps what it had
// (possibly nothing) and stays usable.  The caller has made the pipe's device current.
int pin_block(mtgpu_batch *b, uint64_t records, bool inject_failure = false) {
  int rc = MT_OK;
  unsigned char *h_new = nullptr, *d_new = nullptr, *dev_view = nullptr;
  void *plan_new = nullptr;
  size_t hdr = 0;
  const size_t sbytes = stage_bytes_for(b->cap_frames, records, b->rec_bytes, &hdr);
  size_t flags_off = 0;
  const size_t bytes = sbytes + aux_bytes_for(b->cap_frames, &flags_off);
  const size_t nf = (size_t)b->cap_frames + 1;
  // Driver-allocated pinned memory (hipHostMallocDefault: coherent, mapped into the device), on purpose — see
  // include/mtgpu.h "Memory the device entry points accept" and DESIGN.md §5a for why hipHostRegister'ed user
  // memory is not used although it page-locks five times faster.
  PIPE_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_new), bytes, hipHostMallocDefault));
  if (inject_failure) { rc = fail(MT_ERR_NOMEM, "injected allocation failure (MTGPU_INJECT_GROW_FAIL)"); goto bad; }
  if (b->zero_copy) {
    // no device mirror: the kernel reads the pinned block through its device-visible address and writes the
    // flag bytes straight into it
    PIPE_TRY(hipHostGetDevicePointer(reinterpret_cast<void **>(&dev_view), h_new, 0));
  } else {
    PIPE_TRY(hipMalloc(reinterpret_cast<void **>(&d_new), bytes));
    dev_view = d_new;
  }
  if (!b->d_plan) {                                  // (frames per batch never grow: allocated once)
    PIPE_TRY(hipMalloc(&plan_new, mtgpu::ctx_plan_ws_bytes(b->cap_frames)));
    b->d_plan = plan_new;
    b->plan_bytes = mtgpu::ctx_plan_ws_bytes(b->cap_frames);
    plan_new = nullptr;
  }
  if (b->h_stage) (void)hipHostFree(b->h_stage);
  if (b->d_stage) (void)hipFree(b->d_stage);
  b->h_stage = h_new;
  b->d_stage = d_new;
  b->hdr_bytes = hdr;
  b->stage_bytes = sbytes;
  b->block_bytes = bytes;
  b->h_off = reinterpret_cast<uint64_t *>(h_new);
  b->h_sd = h_new + sizeof(uint64_t) * nf;
  b->h_mv = h_new + hdr;
  b->h_pts = reinterpret_cast<double *>(h_new + sbytes);
  b->h_tag = reinterpret_cast<uint64_t *>(h_new + sbytes + nf * sizeof(double));
  b->h_flags = h_new + sbytes + flags_off;
  b->d_off = reinterpret_cast<uint64_t *>(dev_view);
  b->d_sd = dev_view + sizeof(uint64_t) * nf;
  b->d_mv = dev_view + hdr;
  b->d_flags = dev_view + sbytes + flags_off;
  b->h_off[0] = 0;
  b->cap_records = records;
  return MT_OK;
bad:
  if (h_new) (void)hipHostFree(h_new);
  if (d_new) (void)hipFree(d_new);
  return rc;
}

int pin_block_timed(mtgpu_pipe *p, mtgpu_batch *b, uint64_t records, bool inject_failure = false) {
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = pin_block(b, records, inject_failure);
  p->pin_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}

// Batch i of a pipe: its stream and event now, its staging when `pin_now` (else on first acquire).
int alloc_batch(mtgpu_batch **out, mtgpu_pipe *p, uint64_t max_records, uint32_t max_frames, bool pin_now) {
  int rc = MT_OK;
  mtgpu_batch *b = new (std::nothrow) mtgpu_batch();
  if (!b) return fail(MT_ERR_NOMEM, "out of host memory");
  b->cap_frames = max_frames;
  b->want_records = max_records;
  b->rec_bytes = p->rec_bytes;
  b->zero_copy = p->zero_copy;
  b->owner = p;
  b->stream = mtgpu::ctx_pipe_stream(p->ctx);
  if (!b->stream) {
    PIPE_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    b->own_stream = true;
  }
  // system-scope release at the event: the flag bytes a zero-copy scan wrote into pinned memory
  // are visible to the host thread that waits on it
  PIPE_TRY(hipEventCreateWithFlags(&b->done, hipEventDisableTiming | hipEventReleaseToSystem |
                                                 (p->blocking_events ? hipEventBlockingSync : 0u)));
  if (pin_now) {
    rc = pin_block_timed(p, b, max_records);
    if (rc != MT_OK) goto bad;
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
  p->blocking_events = mtgpu::exp_int("MTGPU_EVENT_BLOCKING", 0) != 0;     // experiments build only
  p->eager_pin = mtgpu::exp_int("MTGPU_PIPE_EAGER", 0) != 0;                // experiments build only
  if (const char *v = std::getenv("MTGPU_INJECT_SUBMIT_FAIL")) p->inject_submit_fail = std::atol(v);
  if (const char *v = std::getenv("MTGPU_INJECT_GROW_FAIL")) p->inject_grow_fail = std::atol(v) != 0;
  if (const char *v = std::getenv("MTGPU_INJECT_COLLECT_FAIL")) p->inject_collect_fail = std::atol(v);
  if (const char *v = std::getenv("MTGPU_INJECT_ONCE")) p->inject_once = std::atol(v) != 0;
  for (int i = 0; i < n_buffers; ++i) {
    mtgpu_batch *b = nullptr;
    // the first batch is pinned here (the caller is about to fill it), the others when they are first acquired
    int rc = alloc_batch(&b, p, max_records_per_batch, max_frames_per_batch, i == 0 || p->eager_pin);
    if (rc != MT_OK) {
      char keep[512];
      std::snprintf(keep, sizeof keep, "%s", mtgpu_last_error());
      mtgpu_pipe_destroy(p);
      return fail(rc, "%s", keep);
    }
    p->bufs.push_back(b);
  }
  *out = p;
  return MT_OK;
}

void mtgpu_pipe_destroy(mtgpu_pipe *p) {
  if (!p) return;
  (void)hipSetDevice(mtgpu::ctx_device(p->ctx));
  for (mtgpu_batch *b : p->bufs) free_batch(b);      // drains every batch's stream first
  delete p;
}

int mtgpu_pipe_acquire(mtgpu_pipe *p, mtgpu_batch **out) {
  if (!p || !out) return fail(MT_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> lock(p->mu);
  mtgpu_batch *pick = nullptr;
  for (mtgpu_batch *b : p->bufs)                    // a free batch that is already pinned, else one that is not yet
    if (b->state == 0 && (b->h_stage || !pick)) {
      pick = b;
      if (b->h_stage) break;
    }
  if (pick) {
    if (!pick->h_stage) {                           // first use: page-lock its staging now
      hipError_t e = use_device(mtgpu::ctx_device(p->ctx));
      if (e != hipSuccess) { *out = nullptr; return hip_fail(e, "hipSetDevice"); }
      const int rc = pin_block_timed(p, pick, pick->want_records);
      if (rc != MT_OK) { *out = nullptr; return rc; }   // the batch stays free and unpinned
    }
    pick->state = 1;
    pick->n_frames = 0;
    pick->n_records = 0;
    pick->h_off[0] = 0;
    *out = pick;
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
    int rc = pin_block_timed(b->owner, b, n + n / 4, b->owner->inject_grow_fail);
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
                                b->rec_bytes, b->zero_copy ? 1 : 0, b->d_plan, b->plan_bytes);
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
  static std::atomic<bool> injected_once{false};
  bool inject = p->inject_collect_fail != 0 && nth == std::labs(p->inject_collect_fail);
  if (inject && p->inject_once && injected_once.exchange(true)) inject = false;
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
  for (const mtgpu_batch *b : p->bufs)
    if (b->h_stage) {                                  // pinned so far (the others pin on first use)
      out->pinned_bytes += b->block_bytes;
      if (!b->zero_copy) out->device_bytes += b->block_bytes;
      out->list_bytes += b->plan_bytes;
      out->pinned_batches += 1;
    }
  out->pin_us = p->pin_us.load();
  for (const mtgpu_batch *b : p->bufs) out->hip_streams += b->own_stream ? 1u : 0u;
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
