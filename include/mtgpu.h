/*
 * mtgpu.h — C ABI of the MI355X (gfx950) motion-vector scanner.
 *
 * Drop-in boundary for ONE path of Vaibhav-20022002/Motion-Estimated-Video-Trimmer:
 * the per-frame MV scan (MotionScanner::check_frame) and the gap-bounded segment
 * merge.  The reference has no FFI for this path (it is plain C++ inside one
 * static library); every entry point below names the reference
 * code it replaces (file:line in the reference tree).  Plain pointers and sizes
 * only; no C++ or torch types cross this boundary.  INTEGRATION.md shows the
 * adapter a maintainer adds on the reference side.
 *
 * Conventions
 *  - every function returns an MT_* status (0 = ok), never throws, and may be
 *    called from many host threads (the reference enters the scanner from
 *    N workers x S streams: src/pipeline.cpp:186-197, src/batch_processor.cpp:152-157);
 *  - `*_device` entry points take DEVICE pointers and a hipStream_t passed as
 *    `void *stream` (NULL = the default stream) and are asynchronous on it
 *    unless stated; the others take HOST pointers and are synchronous;
 *  - a batch of frames is CSR: frame f owns records [frame_off[f], frame_off[f+1])
 *    of one packed array of 40-byte AVMotionVector records (mt_mv).  The bytes
 *    are copied / consumed at call time, as the MV side data of an AVFrame dies
 *    when the frame is reused (src/motion_scanner.cpp:347);
 *  - there is NO CPU fallback: if no gfx950 device is usable every compute entry
 *    point fails with MT_ERR_DEVICE.
 *
 * Memory the `*_device` entry points accept ("device pointer")
 *  - memory of the context's device from hipMalloc / hipMallocAsync / hipMallocFromPoolAsync (what the bench
 *    and the Python host use), or
 *  - pinned host memory ALLOCATED BY THE DRIVER — hipHostMalloc, any flags — through its device address
 *    (hipHostGetDevicePointer).  The pipe's zero-copy staging is exactly this: the kernel streams such memory over
 *    PCIe and writes its one result byte per frame into it.
 *  Every result byte is written once, by one lane; when `d_flags` is not device memory (the entry points ask the
 *  runtime: hipPointerGetAttributes) with a system-scope write-through store, so that no cache keeps the line.
 *  `d_flags` in host memory should share its lines with nothing the HOST writes while the call is in flight — give
 *  it lines of its own (128-byte aligned), as the pipe does.  Results in host memory are complete when
 *  an event recorded behind the call on `stream` (hipEventReleaseToSystem) or a stream synchronisation has passed.
 *  NOT supported, and not checked: host memory page-locked with hipHostRegister (a user-pointer mapping of pages the
 *  process owns, which the kernel driver re-validates whenever the OS changes the page tables underneath — the one
 *  wrong result this library ever returned came from such staging), hipMallocManaged memory that is
 *  not resident on the device, and memory of another device.
 *
 * Environment (read once, at mtgpu_create / at the first use; csrc/knobs.h)
 *    production    MTGPU_CHECK_OFFSETS=1   device entry points verify frame_off first (one sync per call)
 *                  MTGPU_PACK=scalar|avx2|avx512   copy-out loop of mtgpu_pack_records / the pipe (default: best the CPU has)
 *                  MTGPU_ALIAS_DEVICES=N   present N logical devices on the GPUs that exist (rehearsals on small boxes)
 *                  C++ host layer only: MTGPU_STAGING, MTGPU_BATCH_MB, MTGPU_CPU_TOKENS, MTGPU_CPU_WINDOW (mtgpu_host.hpp)
 *    tests only    MTGPU_FORCE_FB=32|1|2|4|8|108, MTGPU_FORCE_BLOCK=512|1024, MTGPU_FORCE_SLICES, MTGPU_GROUP,
 *                  MTGPU_ITEM_CHUNK, MTGPU_MERGE_LARGE_MIN (reach every kernel form on small inputs),
 *                  MTGPU_INJECT_SUBMIT_FAIL / _GROW_FAIL / _COLLECT_FAIL / MTGPU_INJECT_ONCE (pipe error paths)
 *  (A/B switches of measurements exist only in the experiments build, `make -C csrc experiments`; this library
 *  ignores them.  Their table: csrc/knobs.h.)
 */
#ifndef MTGPU_H
#define MTGPU_H

#include "mt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mtgpu_ctx mtgpu_ctx;

/* Library identification string. */
const char *mtgpu_version(void);

/* Last error message of the calling thread ("" if none). Replaces the reference's
 * LOG_ERROR lines (src/motion_scanner.cpp:66-177). */
const char *mtgpu_last_error(void);

/* Number of usable HIP devices (0 if none / no driver). Host-side analogue of
 * get_available_cpus() for the stream->GPU assignment (src/system.cpp:166-184). */
int mtgpu_device_count(void);

/* PCI address of a device as "dddd:bb:dd.f" (lower case) into buf — what /sys/bus/pci/devices/<address>/ is named
 * after, where `local_cpulist` and `numa_node` say which CPUs sit next to the GPU.  The host layer places its
 * worker threads there (the reference pins its stream threads to CPU sets: src/batch_processor.cpp:102-110,
 * src/system.cpp:199-225).  MT_ERR_CAPACITY if cap < 13. */
int mtgpu_device_pci_address(int device, char *buf, uint64_t cap);

/*
 * Derive the parameter block exactly as MotionScanner::initialize() does
 * (src/motion_scanner.cpp:184-199; defaults include/motion_trim/config.hpp:56-89):
 *   grid_w/h = int16((dim + block_size - 1) >> block_shift),
 *   vertical_margin = int(float(grid_h) * vertical_mask)   [float32 product],
 *   vectors_needed = uint8(vectors_needed).
 * Pure host arithmetic; no device needed.
 */
int mtgpu_params_from_config(mt_scan_params *out, int width, int height,
                             double mv_threshold_sq, int block_size, int block_shift,
                             int vectors_needed, int clusters_needed, float vertical_mask);

/*
 * Create / destroy a scanner context bound to one device.  Replaces the
 * per-worker `MotionScanner` instance's scan state (cfg + grid_votes,
 * include/motion_trim/motion_scanner.hpp:75-93): the vote grid now lives in LDS,
 * the context only holds the validated parameters, the launch plan and a
 * private stream + staging buffers for the host-pointer entry points.
 */
int mtgpu_create(const mt_scan_params *params, int device, mtgpu_ctx **out);
void mtgpu_destroy(mtgpu_ctx *ctx);
int mtgpu_get_params(const mtgpu_ctx *ctx, mt_scan_params *out);

/*
 * What a context holds on the device, and how to give it back.  No reference counterpart: the
 * reference's per-scanner memory is the grid_votes vector (src/motion_scanner.cpp:199), which lives
 * in LDS here.  Launch scratch (the work list of a device-resident scan: 32 bytes per frame; the spill queue of
 * banded plans: 4 bytes per record of the largest batch scanned so far; slice tiles; the merge workspace: 24 bytes
 * per timestamp) comes from a ring of up to 16 device blocks owned by the context, each guarded by the event of its
 * last user (a block never has two users, on whatever streams and threads the launches run), and KEPT until
 * mtgpu_trim or mtgpu_destroy, so that a steady stream of batches never allocates.  A host that scans
 * one huge batch and then idles should call mtgpu_trim; the C++ host layer does so between videos when a context has
 * one user.
 */
typedef struct mtgpu_ctx_stats {
  uint64_t staging_device_bytes;  /* grow-only device buffers of the HOST-pointer entry points   */
  uint64_t pool_reserved_bytes;   /* device memory the scratch ring holds right now             */
  uint64_t pool_reserved_high;    /* its high-water mark                                        */
  uint32_t hip_streams;           /* streams owned by the context: its own + the pipe-stream pool */
  uint32_t private_pool;          /* 1: scratch from the context's own ring of blocks (always, since round 6) */
} mtgpu_ctx_stats;
int mtgpu_get_stats(mtgpu_ctx *ctx, mtgpu_ctx_stats *out);
/* Return the scratch blocks nobody uses to the device (blocks of launches still in flight stay);
 * safe at any time, from any thread; scans that follow simply allocate again. */
int mtgpu_trim(mtgpu_ctx *ctx);

/* Launch timing — a profiling aid, off by default.  While on, every scan launched through the context records three
 * HIP events on its stream: before the planning kernels (which answer the frames without side data and list the
 * others), between them and the scan kernel, after the scan kernel — from a ring of 64 triples (a launch that finds
 * the ring full first waits for the oldest one).  mtgpu_profile_read waits for the launches recorded so far, adds up
 * their times (*plan_ms, *scan_ms: totals over *launches launches) and starts over.  bench.py uses it to time the
 * scan kernel by itself, as `rocprofv3 --kernel-trace --stats` does.  For one user at a time: launches of a context
 * that is being profiled are serialised on the host. */
int mtgpu_profile_enable(mtgpu_ctx *ctx, int on);
int mtgpu_profile_read(mtgpu_ctx *ctx, double *plan_ms, double *scan_ms, uint32_t *launches);

/* Launch plan chosen for the context's grid (for reports and tests). */
typedef struct mtgpu_plan {
  int32_t block_threads;   /* workgroup size                                      */
  int32_t bands;           /* row bands per frame (1 = whole grid in one LDS tile) */
  int32_t band_rows;       /* analysed rows per band                               */
  int32_t lds_bytes;       /* dynamic LDS per workgroup                            */
  int32_t counter_bits;    /* bits per LDS vote counter: 32, or 1/2/4/8 packed           */
  int32_t device;
  int32_t cu_count;
  int32_t chunk_rows;      /* centre rows per cluster-test chunk (mask buffer rows - 2) */
  int32_t counter_mode;    /* 0 = 32-bit add, 1 = thermometer (OR), 2 = 8-bit binary (CAS) */
  int32_t _pad;
} mtgpu_plan;
int mtgpu_get_plan(const mtgpu_ctx *ctx, mtgpu_plan *out);
/* The plan mtgpu_create would choose on a device with `lds_bytes_per_workgroup` of LDS (gfx950:
 * 163840) and `cu_count` CUs — pure host arithmetic, no device needed (capacity planning, tests). */
int mtgpu_plan_preview(const mt_scan_params *params, int lds_bytes_per_workgroup, int cu_count,
                       mtgpu_plan *out);

/* Workgroups per frame along the record array: 0 = automatic (the default: split frames only
 * when a batch has too few frames to fill the chip and the frames are large), or 1, 2, 4, 8.
 * Results never depend on it. */
int mtgpu_set_slices(mtgpu_ctx *ctx, int slices);

/*
 * check_frame() over a device-resident batch — replaces the per-frame call at
 * src/motion_scanner.cpp:376 (body :217-295).
 *   d_mv         packed mt_mv records, n_records of them (device)
 *   d_frame_off  n_frames+1 record offsets (device), NON-DECREASING: frames are disjoint record ranges
 *                (entries beyond n_records are clamped to it).  PRECONDITION, not checked by default — the
 *                offsets live in device memory and the call is asynchronous.  With a decreasing entry two
 *                frames overlap: a single-tile plan merely scans the shared records for both, but a banded
 *                plan (mtgpu_get_plan: bands > 1) keeps frame f's vote queue at the frame's own offset, one
 *                slot per record, so overlapping frames race on it and their flags are undefined.
 *                MTGPU_CHECK_OFFSETS=1 in the environment of mtgpu_create makes both device entry points
 *                verify the offsets on the device first (one small kernel + one stream synchronisation per
 *                call) and return MT_ERR_INVALID, naming the first offending frame, before anything is scanned.
 *                (mtgpu_scan_frames validates its HOST offsets always; the pipe builds its own.)
 *   d_has_sd     optional (device, may be NULL): has_sd[f]==0 <=> the frame had no
 *                AV_FRAME_DATA_MOTION_VECTORS side data (-> false, :219-221).
 *                NULL: a frame has side data iff it owns >= 1 record.
 *   d_flags      n_frames bytes (device): 1 = significant motion, 0 = none
 * Asynchronous on `stream`: two or three launches — the planning kernels (frames without side data are answered
 * there and never get a workgroup: 4-14 us per call, measured as `plan_ms` by mtgpu_profile_read), then the scan.
 * What to expect: the rate does not depend on where the frames without records fall (key frame every 2 / 8 / 16 / 30
 * frames: within 0.3 % of none for large frames, 1.6 % for 10^5 small ones, profiles/r06_gop_sweep_*.log,
 * r06_staged_entries.log) nor on which HIP streams the process has used
 * before — through round 5 a launch on the default stream read 2 % faster once another stream of the process had run
 * a kernel, and only with a key frame every 30 frames (bench.py's state and workload): a single-stream caller now
 * gets the bench's rate (2952 us per call of the headline batch, single stream, against 2946 us in bench.py).
 */
int mtgpu_scan_frames_device(mtgpu_ctx *ctx, const void *d_mv, uint64_t n_records,
                             const uint64_t *d_frame_off, const uint8_t *d_has_sd,
                             uint32_t n_frames, uint8_t *d_flags, void *stream);

/*
 * The same scan over COMPACT records (mt_mv_compact: bytes 6..13 of each AVMotionVector, the
 * only bytes check_frame reads, src/motion_scanner.cpp:246-256), 8 bytes each, device-resident
 * and 8-byte aligned.  This is what the host dispatcher below stages and ships over PCIe; a
 * host that packs its own batches (mtgpu_pack_records) can call it directly.  Same results as
 * mtgpu_scan_frames_device on the records the compact form was packed from.
 */
int mtgpu_scan_frames_device_compact(mtgpu_ctx *ctx, const void *d_rec8, uint64_t n_records,
                                     const uint64_t *d_frame_off, const uint8_t *d_has_sd,
                                     uint32_t n_frames, uint8_t *d_flags, void *stream);

/* Host helper (data movement only, no result is computed): copy bytes 6..13 of each of
 * n_records 40-byte AVMotionVector records at `mv_bytes` into 8-byte compact records at `out8`. */
int mtgpu_pack_records(const void *mv_bytes, uint64_t n_records, void *out8);
/* The copy-out loop behind mtgpu_pack_records and mtgpu_batch_add_frame is chosen once per process from
 * what the host CPU supports (AVX-512BW: five 64-byte loads -> one full 64-byte line per 8 records; AVX2;
 * scalar), with non-temporal full-line stores into the staging the GPU reads next (csrc/pack_simd.cpp;
 * MTGPU_PACK=scalar|avx2|avx512 overrides).  mtgpu_pack_selected() reports the
 * choice; mtgpu_pack_records_with runs one given loop (tests, microbenchmarks): byte-identical output,
 * MT_ERR_UNSUPPORTED if this CPU cannot run it. */
#define MT_PACK_SCALAR 1
#define MT_PACK_AVX2 2
#define MT_PACK_AVX512 3
#define MT_PACK_IMPL_MASK 15
#define MT_PACK_NT 16        /* OR-ed in: non-temporal stores (vector loops, 8-byte aligned destination) */
#define MT_PACK_PREFETCH_LINES(n) (((n) & 255) << 8)   /* OR-ed in: software-prefetch the source n 64-byte lines ahead */
int mtgpu_pack_records_with(int impl_flags, const void *mv_bytes, uint64_t n_records, void *out8);
int mtgpu_pack_selected(void);

/* Same for HOST pointers: validates frame_off, copies the batch to the device,
 * scans, copies the flags back, synchronous.  This is the call an adapter makes
 * after copying each AVFrame's side data into a batch. */
int mtgpu_scan_frames(mtgpu_ctx *ctx, const mt_mv *mv, const uint64_t *frame_off,
                      const uint8_t *has_sd, uint32_t n_frames, uint8_t *flags);

/*
 * Segment merge — replaces src/pipeline.cpp:302-358, 387-388 (std::sort +
 * std::unique of the pooled motion timestamps, the gap-bounded merge, the clamp,
 * the savings and the cut decision).  Runs on the device.
 *   ts            n motion timestamps in ANY order, duplicates allowed (host)
 *   job_semantics 0: `out` = merged, clamped segments; 1: what FFmpegJob::segments
 *                 would carry (full-copy segment {0,duration} when the savings are
 *                 too low; nothing when there was no motion)
 *   out/cap       host array; MT_ERR_CAPACITY if too small (res->n_segments = need)
 */
int mtgpu_merge_segments(mtgpu_ctx *ctx, const double *ts, uint64_t n,
                         const mt_merge_params *mp, int job_semantics,
                         mt_segment *out, uint64_t cap, mt_merge_result *res);

/*
 * The same merge for ONE stream's pooled motion timestamps that already sit on the device (any
 * order, duplicates allowed) — e.g. the timestamps all-gathered from the ranks that scanned
 * time ranges of one video.  d_ts / d_seg / d_res are device pointers, `mp` is a host pointer
 * (copied at call time).  Lists of a few thousand entries and more are sorted and merged by many
 * workgroups (device-wide merge sort + scan), so a day of footage (~10^6 timestamps) merges in
 * well under a millisecond.  Asynchronous on `stream`; d_res->status reports NaN input.
 */
int mtgpu_merge_timestamps_device(mtgpu_ctx *ctx, const double *d_ts, uint64_t n,
                                  const mt_merge_params *mp, int job_semantics,
                                  mt_segment *d_seg, uint64_t seg_cap, mt_merge_result *d_res,
                                  void *stream);

/*
 * Streams on the device, end to end: for S independent streams whose frames sit
 * contiguously in one batch (stream s owns frames [stream_off[s], stream_off[s+1])),
 * turn per-frame flags + per-frame pts into per-stream segment lists without
 * leaving the device: compaction (src/motion_scanner.cpp:382-383), sort+unique
 * (src/pipeline.cpp:302-304), merge/clamp/savings/decision (:323-358, 387-388).
 *   d_flags       n_frames bytes from mtgpu_scan_frames_device
 *   d_pts         n_frames doubles: pts of each frame in seconds (:361)
 *   d_stream_off  S+1 frame offsets (device)
 *   d_mp          S merge-parameter blocks (device)
 *   d_ts          workspace, 2 * n_frames doubles (device), n_frames = d_stream_off[S]: stream s keeps its
 *                 compacted timestamps in [2a, 2a + (b-a)) and its per-segment durations in [2a + (b-a), 2b),
 *                 a = d_stream_off[s], b = d_stream_off[s+1] (every flagged frame may be a segment of its own);
 *                 contents unspecified afterwards; must stay allocated until the call has finished on `stream`
 *   d_seg         S * seg_cap segments (device): stream s writes at d_seg[s*seg_cap], never beyond seg_cap
 *   d_res         S results (device); n_segments > seg_cap signals truncation
 * Asynchronous on `stream`.  (tests/c/abi_buffer_canaries.c allocates every buffer of this header at exactly
 * its stated size with a canary behind it.)
 */
int mtgpu_merge_streams_device(mtgpu_ctx *ctx, const uint8_t *d_flags, const double *d_pts,
                               const uint64_t *d_stream_off, uint32_t n_streams,
                               const mt_merge_params *d_mp, int job_semantics,
                               double *d_ts, mt_segment *d_seg, uint64_t seg_cap,
                               mt_merge_result *d_res, void *stream);

/* ---------------------------------------------------------------------------
 * Host dispatcher: pinned, multi-buffered H2D + scan pipeline on one device.
 *
 * Replaces the synchronous per-frame call inside the decode loop
 * (src/motion_scanner.cpp:375-383) for a host decoder thread: the thread copies each
 * AVFrame's MV side data into a pinned staging batch (the side data dies when the frame
 * is reused, :347), submits the batch asynchronously (H2D copy, scan kernel, flags D2H on
 * the batch's own stream) and keeps decoding into the next staging batch; results are
 * collected in submission order.  `n_buffers` batches bound the memory in flight
 * (back-pressure: acquire fails with MT_ERR_BUSY until a batch is collected and released).
 * One pipe per decoder thread — the reference's one-MotionScanner-per-worker model
 * (src/pipeline.cpp:186-197); pipes of one context may be used concurrently, so N worker threads need
 * N pipes but only ONE context per device.  Each batch's staging is one pinned block (plus a device mirror
 * without MT_LAYOUT_ZERO_COPY), page-locked when the batch is first used: creating a pipe pins one batch, the
 * others follow on their first mtgpu_pipe_acquire (a few ms each), so S x T workers that start together are
 * not queued behind S x T x n_buffers page-locking calls, and a worker that never has more than one batch in
 * flight never pins the rest.
 */
typedef struct mtgpu_pipe mtgpu_pipe;
typedef struct mtgpu_batch mtgpu_batch;

int mtgpu_pipe_create(mtgpu_ctx *ctx, uint64_t max_records_per_batch, uint32_t max_frames_per_batch,
                      int n_buffers, mtgpu_pipe **out);
/* Staging layout of a pipe.  COMPACT8: add_frame copies only the 8 bytes per record that the scan
 * reads into pinned memory, so 5x fewer bytes cross PCIe; AOS40: the 40-byte records are staged
 * unchanged.  mtgpu_pipe_create picks COMPACT8 | ZERO_COPY.  Results are identical. */
#define MT_LAYOUT_COMPACT8 0
#define MT_LAYOUT_AOS40 1
/* OR-ed into either layout: no H2D / D2H copy commands — the scan kernel reads the pinned staging
 * over PCIe itself and writes the flags into pinned memory (one launch + one event per batch). */
#define MT_LAYOUT_ZERO_COPY 2
int mtgpu_pipe_create_layout(mtgpu_ctx *ctx, uint64_t max_records_per_batch, uint32_t max_frames_per_batch,
                             int n_buffers, int layout, mtgpu_pipe **out);
void mtgpu_pipe_destroy(mtgpu_pipe *pipe);

/* A free staging batch to fill, or MT_ERR_BUSY if all are in flight / held (MT_ERR_DEVICE if every batch
 * of the pipe was retired after failed collects: destroy the pipe).  Pins the batch's staging on its first
 * use; if that fails (MT_ERR_DEVICE / MT_ERR_NOMEM) the batch stays free and the pipe stays usable. */
int mtgpu_pipe_acquire(mtgpu_pipe *pipe, mtgpu_batch **out);

/* Append one decoded frame: copies n_bytes / 40 records (trailing bytes ignored, :226).
 * mv_bytes == NULL with has_side_data == 0 records a frame without MV side data (:219-221).
 * `tag` is carried through untouched (e.g. a frame index).  MT_ERR_CAPACITY if the frame
 * does not fit the batch: submit this batch and add the frame to the next one.  A single frame
 * with more records than a whole batch is accepted (check_frame takes any count): an EMPTY
 * batch grows its staging to hold it; if that allocation fails (MT_ERR_NOMEM / MT_ERR_DEVICE) the
 * batch keeps its previous staging and stays usable for frames that fit. */
int mtgpu_batch_add_frame(mtgpu_batch *batch, const void *mv_bytes, uint64_t n_bytes,
                          int has_side_data, double pts, uint64_t tag);
uint32_t mtgpu_batch_frames(const mtgpu_batch *batch);

/* Asynchronous: H2D copy + scan + flags D2H on the batch's stream.  Empty batches are legal.
 * On failure nothing of the batch is left in flight and it is still being filled (retry the
 * submit, or release it). */
int mtgpu_pipe_submit(mtgpu_pipe *pipe, mtgpu_batch *batch);

/* Block until the OLDEST submitted batch is done and expose its results (host pointers valid
 * until mtgpu_pipe_release).  MT_ERR_INVALID if nothing is in flight.  If waiting fails, *out
 * is still set so that the batch can be released — its results are lost, but its stream has been
 * drained first, so the staging may be refilled at once (with zero-copy staging the kernel reads
 * that pinned memory itself).  If even the drain fails the batch is retired: release accepts it,
 * acquire never returns it again, and only mtgpu_pipe_destroy frees it. */
int mtgpu_pipe_collect(mtgpu_pipe *pipe, mtgpu_batch **out, const uint8_t **flags,
                       const double **pts, const uint64_t **tags, uint32_t *n_frames);
int mtgpu_pipe_release(mtgpu_pipe *pipe, mtgpu_batch *batch);

/* What a pipe costs: every worker thread of the reference's N x S model (src/pipeline.cpp:186-197,
 * src/batch_processor.cpp:152-157) owns one pipe, so 64 streams x T workers multiply these. */
typedef struct mtgpu_pipe_stats {
  uint64_t pinned_bytes;   /* page-locked host memory SO FAR: staging blocks + pts / tag / flag arrays */
  uint64_t device_bytes;   /* device mirrors of the staging (0 with MT_LAYOUT_ZERO_COPY)           */
  uint64_t submits;        /* batches submitted so far                                             */
  uint32_t n_buffers;      /* staging batches = HIP events owned by the pipe                        */
  int32_t layout;          /* MT_LAYOUT_* flags                                                    */
  uint64_t pin_us;         /* time spent page-locking staging (creation, first uses, growth)       */
  uint32_t pinned_batches; /* batches whose staging is pinned (the first at creation, the others   *
                            * when mtgpu_pipe_acquire first hands them out)                         */
  uint32_t hip_streams;    /* HIP streams OWNED by the pipe: 0 — batches run on the context's pool of 8 streams      *
                            * (creating a stream costs ~3.5 ms, serialised by the runtime)                    */
  uint64_t list_bytes;     /* device memory for the batches' work lists (32 bytes per frame a batch can hold): each *
                            * pinned batch owns one, so that its scans allocate nothing                       */
} mtgpu_pipe_stats;
int mtgpu_pipe_get_stats(mtgpu_pipe *pipe, mtgpu_pipe_stats *out);

/* ---------------------------------------------------------------------------
 * Multi-GPU exchange for hosts that run one process per GPU without torch.distributed:
 * thin wrappers over RCCL (loaded lazily; librccl.so.1 must be installed).  The path has
 * exactly one exchange step — the per-GPU segment lists (or motion timestamps) after the
 * scan — so one all-gather of fixed-size records is all that is exported.  The reference has
 * no counterpart (single process, std::mutex pooling: src/task_queue.cpp:43-57).
 *
 *   rank 0: mtgpu_comm_unique_id(id)  -> ship the 128 bytes to the other ranks by any means
 *   all   : mtgpu_comm_create(rank, n_ranks, id, device, &comm)
 *   all   : mtgpu_gather_segments(comm, d_send, bytes, d_recv, stream)   [d_recv: n_ranks*bytes]
 */
#define MTGPU_UNIQUE_ID_BYTES 128
typedef struct mtgpu_comm mtgpu_comm;
int mtgpu_comm_unique_id(void *id_out /* MTGPU_UNIQUE_ID_BYTES */);
int mtgpu_comm_create(int rank, int n_ranks, const void *id, int device, mtgpu_comm **out);
void mtgpu_comm_destroy(mtgpu_comm *comm);
/* All-gather `bytes_per_rank` bytes from every rank's d_send into d_recv (rank-major),
 * asynchronous on `stream`.  Typically d_send is the packed {segments, mt_merge_result} block
 * of this rank's streams. */
int mtgpu_gather_segments(mtgpu_comm *comm, const void *d_send, uint64_t bytes_per_rank,
                          void *d_recv, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MTGPU_H */
