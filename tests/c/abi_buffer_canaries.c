/* abi_buffer_canaries.c — every caller-sized buffer of include/mtgpu.h, allocated EXACTLY as the header states,
 * with a canary region behind it that must be untouched after the call.  A header that under-states a buffer
 * (round 4: "d_ts: n_frames doubles" where the kernel used 2 * n_frames) makes this program fail.
 * Plain C, HIP runtime API only for device memory; built and run by tests/test_gpu_abi_contracts.py (-m gpu).
 *
 *   gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/abi_buffer_canaries.c \
 *       -Lmotion-estimated-video-trimmer_amd -lmtgpu -L/opt/rocm/lib -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mtgpu.h"

#define CANARY 4096u           /* bytes behind every buffer */
#define PATTERN 0xA5

static int failures = 0;

#define MT(call)                                                                   \
  do {                                                                             \
    int rc_ = (call);                                                              \
    if (rc_ != MT_OK) {                                                            \
      fprintf(stderr, "%s:%d %s -> %d: %s\n", __FILE__, __LINE__, #call, rc_, mtgpu_last_error()); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)
#define HIP(call)                                                                  \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

/* device buffer of `bytes` USABLE bytes, the canary right behind them */
typedef struct dbuf { unsigned char *p; size_t bytes; const char *what; } dbuf;

static dbuf dalloc(size_t bytes, const char *what) {
  dbuf b = {NULL, bytes, what};
  HIP(hipMalloc((void **)&b.p, bytes + CANARY));
  if (bytes) HIP(hipMemset(b.p, 0, bytes));
  HIP(hipMemset(b.p + bytes, PATTERN, CANARY));
  return b;
}

static void dcheck(dbuf *b, const char *call) {
  static unsigned char host[CANARY];
  HIP(hipMemcpy(host, b->p + b->bytes, CANARY, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < CANARY; ++i)
    if (host[i] != PATTERN) {
      fprintf(stderr, "FAIL %s: %s (%zu bytes as the header sizes it) was overrun: canary byte %zu = 0x%02x\n", call,
              b->what, b->bytes, i, host[i]);
      ++failures;
      break;
    }
  HIP(hipFree(b->p));
  b->p = NULL;
}

/* host buffer with a canary behind it */
static void *halloc(size_t bytes) {
  unsigned char *p = malloc(bytes + CANARY);
  if (!p) exit(2);
  memset(p, 0, bytes);
  memset(p + bytes, PATTERN, CANARY);
  return p;
}
static void hcheck(void *p, size_t bytes, const char *call, const char *what) {
  const unsigned char *c = (const unsigned char *)p + bytes;
  for (size_t i = 0; i < CANARY; ++i)
    if (c[i] != PATTERN) {
      fprintf(stderr, "FAIL %s: host buffer %s (%zu bytes) was overrun at +%zu\n", call, what, bytes, i);
      ++failures;
      break;
    }
  free(p);
}

/* a frame of `n` records that all vote into two neighbouring cells of row 30 (1080p, code defaults: flag 1) */
static void fill_frame(mt_mv *mv, size_t n) {
  for (size_t k = 0; k < n; ++k) {
    mv[k].dst_x = (int16_t)(16 * (40 + (int)(k & 1)) + 8);
    mv[k].dst_y = (int16_t)(16 * 30 + 8);
    mv[k].src_x = (int16_t)(mv[k].dst_x - 6);
    mv[k].src_y = mv[k].dst_y;
    mv[k].w = mv[k].h = 8;
    mv[k].source = -1;
  }
}

int main(int argc, char **argv) {
  /* self-check of the canaries: "round4-header" sizes d_ts as round 4's header stated it (n_frames doubles);
   * the program must then FAIL on d_ts — run by the test to show that an under-stated size is caught */
  const size_t ts_per_frame = (argc > 1 && !strcmp(argv[1], "round4-header")) ? 1 : 2;
  mt_scan_params p;
  MT(mtgpu_params_from_config(&p, 1920, 1080, 16.0, 16, 4, 2, 2, 0.05f));
  mtgpu_ctx *ctx = NULL;
  MT(mtgpu_create(&p, 0, &ctx));
  hipStream_t st;
  HIP(hipStreamCreate(&st));

  /* ---- batch: F frames, 4 records each (every frame says "motion"), S streams of equal length */
  enum { S = 5, PER_STREAM = 37, F = S * PER_STREAM, PER = 4 };
  const size_t n_rec = (size_t)F * PER;
  mt_mv *mv = calloc(n_rec, sizeof *mv);
  uint64_t *off = malloc(sizeof(uint64_t) * (F + 1));
  double *pts = malloc(sizeof(double) * F);
  fill_frame(mv, n_rec);
  for (int f = 0; f <= F; ++f) off[f] = (uint64_t)f * PER;
  /* 10 s between frames and MAX_GAP 5 s: EVERY flagged frame is its own segment (K = M = frames of the stream),
   * the case in which the merge workspace is used to its last double */
  for (int f = 0; f < F; ++f) pts[f] = 10.0 * (f % PER_STREAM);

  /* 1. mtgpu_scan_frames_device: d_flags = n_frames bytes */
  dbuf d_mv = dalloc(n_rec * sizeof(mt_mv), "d_mv");
  dbuf d_off = dalloc(sizeof(uint64_t) * (F + 1), "d_frame_off");
  dbuf d_flags = dalloc(F, "d_flags (n_frames bytes)");
  HIP(hipMemcpy(d_mv.p, mv, n_rec * sizeof(mt_mv), hipMemcpyHostToDevice));
  HIP(hipMemcpy(d_off.p, off, sizeof(uint64_t) * (F + 1), hipMemcpyHostToDevice));
  MT(mtgpu_scan_frames_device(ctx, d_mv.p, n_rec, (const uint64_t *)d_off.p, NULL, F, d_flags.p, st));
  HIP(hipStreamSynchronize(st));
  {
    unsigned char fl[F];
    HIP(hipMemcpy(fl, d_flags.p, F, hipMemcpyDeviceToHost));
    for (int f = 0; f < F; ++f)
      if (fl[f] != 1) { fprintf(stderr, "FAIL scan: frame %d flag %d, expected 1\n", f, fl[f]); ++failures; break; }
  }

  /* 2. mtgpu_scan_frames_device_compact: n_records * 8 bytes in, n_frames bytes out */
  dbuf d_rec8 = dalloc(n_rec * MT_COMPACT_BYTES, "d_rec8");
  dbuf d_flags8 = dalloc(F, "d_flags (compact scan)");
  {
    void *rec8 = halloc(n_rec * MT_COMPACT_BYTES);
    MT(mtgpu_pack_records(mv, n_rec, rec8));                    /* out8 = n_records * 8 bytes (host) */
    HIP(hipMemcpy(d_rec8.p, rec8, n_rec * MT_COMPACT_BYTES, hipMemcpyHostToDevice));
    hcheck(rec8, n_rec * MT_COMPACT_BYTES, "mtgpu_pack_records", "out8");
  }
  MT(mtgpu_scan_frames_device_compact(ctx, d_rec8.p, n_rec, (const uint64_t *)d_off.p, NULL, F, d_flags8.p, st));
  HIP(hipStreamSynchronize(st));
  dcheck(&d_rec8, "mtgpu_scan_frames_device_compact");
  dcheck(&d_flags8, "mtgpu_scan_frames_device_compact");

  /* 3. mtgpu_merge_streams_device: d_ts = 2 * n_frames doubles, d_seg = S * seg_cap, d_res = S
   *    once with room for every segment, once truncated (seg_cap < K: the list is cut, never overrun) */
  for (int pass = 0; pass < 2; ++pass) {
    const uint64_t seg_cap = pass == 0 ? PER_STREAM : 3;
    uint64_t soff[S + 1];
    mt_merge_params mp[S];
    for (int s = 0; s <= S; ++s) soff[s] = (uint64_t)s * PER_STREAM;
    for (int s = 0; s < S; ++s) { mp[s].max_gap_sec = 5.0; mp[s].padding_sec = 0.5; mp[s].duration = 10.0 * PER_STREAM; mp[s].min_savings_pct = 5.0; }
    dbuf d_pts = dalloc(sizeof(double) * F, "d_pts");
    dbuf d_soff = dalloc(sizeof soff, "d_stream_off");
    dbuf d_mp = dalloc(sizeof mp, "d_mp");
    dbuf d_ts = dalloc(sizeof(double) * ts_per_frame * F, ts_per_frame == 2 ? "d_ts (2 * n_frames doubles)" : "d_ts (n_frames doubles)");
    dbuf d_seg = dalloc(sizeof(mt_segment) * S * seg_cap, "d_seg (S * seg_cap)");
    dbuf d_res = dalloc(sizeof(mt_merge_result) * S, "d_res (S)");
    HIP(hipMemcpy(d_pts.p, pts, sizeof(double) * F, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_soff.p, soff, sizeof soff, hipMemcpyHostToDevice));
    HIP(hipMemcpy(d_mp.p, mp, sizeof mp, hipMemcpyHostToDevice));
    MT(mtgpu_merge_streams_device(ctx, d_flags.p, (const double *)d_pts.p, (const uint64_t *)d_soff.p, S,
                                  (const mt_merge_params *)d_mp.p, 0, (double *)d_ts.p, (mt_segment *)d_seg.p, seg_cap,
                                  (mt_merge_result *)d_res.p, st));
    HIP(hipStreamSynchronize(st));
    mt_merge_result res[S];
    HIP(hipMemcpy(res, d_res.p, sizeof res, hipMemcpyDeviceToHost));
    for (int s = 0; s < S; ++s)
      if (res[s].n_segments != PER_STREAM || res[s].n_timestamps != PER_STREAM || res[s].status != MT_OK) {
        fprintf(stderr, "FAIL merge_streams pass %d: stream %d n_segments %llu n_timestamps %llu status %d\n", pass, s,
                (unsigned long long)res[s].n_segments, (unsigned long long)res[s].n_timestamps, res[s].status);
        ++failures;
      }
    dcheck(&d_pts, "mtgpu_merge_streams_device");
    dcheck(&d_soff, "mtgpu_merge_streams_device");
    dcheck(&d_mp, "mtgpu_merge_streams_device");
    dcheck(&d_ts, "mtgpu_merge_streams_device");
    dcheck(&d_seg, "mtgpu_merge_streams_device");
    dcheck(&d_res, "mtgpu_merge_streams_device");
  }
  dcheck(&d_mv, "mtgpu_scan_frames_device");
  dcheck(&d_off, "mtgpu_scan_frames_device");
  dcheck(&d_flags, "mtgpu_scan_frames_device");

  /* 4. mtgpu_merge_timestamps_device: d_ts n doubles (input), d_seg seg_cap, d_res 1 — the one-workgroup path
   *    (n < 4096) and the multi-workgroup path, each with every timestamp its own segment and seg_cap < K */
  {
    const uint64_t sizes[2] = {1000, 20000};
    for (int q = 0; q < 2; ++q) {
      const uint64_t n = sizes[q], seg_cap = 7;
      double *t = malloc(sizeof(double) * n);
      for (uint64_t i = 0; i < n; ++i) t[i] = 10.0 * (double)((i * 7919u) % n);     /* shuffled, distinct */
      mt_merge_params mp1 = {5.0, 0.5, 10.0 * (double)n, 5.0};
      dbuf d_t = dalloc(sizeof(double) * n, "d_ts (n doubles, input)");
      dbuf d_sg = dalloc(sizeof(mt_segment) * seg_cap, "d_seg (seg_cap)");
      dbuf d_r = dalloc(sizeof(mt_merge_result), "d_res (1)");
      HIP(hipMemcpy(d_t.p, t, sizeof(double) * n, hipMemcpyHostToDevice));
      MT(mtgpu_merge_timestamps_device(ctx, (const double *)d_t.p, n, &mp1, 0, (mt_segment *)d_sg.p, seg_cap,
                                       (mt_merge_result *)d_r.p, st));
      HIP(hipStreamSynchronize(st));
      mt_merge_result r;
      HIP(hipMemcpy(&r, d_r.p, sizeof r, hipMemcpyDeviceToHost));
      if (r.n_segments != n || r.status != MT_OK) {
        fprintf(stderr, "FAIL merge_timestamps n=%llu: n_segments %llu status %d\n", (unsigned long long)n,
                (unsigned long long)r.n_segments, r.status);
        ++failures;
      }
      dcheck(&d_t, "mtgpu_merge_timestamps_device");
      dcheck(&d_sg, "mtgpu_merge_timestamps_device");
      dcheck(&d_r, "mtgpu_merge_timestamps_device");

      /* 5. mtgpu_merge_segments (host pointers): out = cap segments; MT_ERR_CAPACITY when cap < need */
      mt_segment *out = halloc(sizeof(mt_segment) * seg_cap);
      mt_merge_result hr;
      const int rc = mtgpu_merge_segments(ctx, t, n, &mp1, 0, out, seg_cap, &hr);
      if (rc != MT_ERR_CAPACITY || hr.n_segments != n) {
        fprintf(stderr, "FAIL merge_segments n=%llu: rc %d n_segments %llu\n", (unsigned long long)n, rc,
                (unsigned long long)hr.n_segments);
        ++failures;
      }
      hcheck(out, sizeof(mt_segment) * seg_cap, "mtgpu_merge_segments", "out (cap segments)");
      free(t);
    }
  }

  /* 6. mtgpu_scan_frames (host pointers): flags = n_frames bytes */
  {
    unsigned char *fl = halloc(F);
    MT(mtgpu_scan_frames(ctx, mv, off, NULL, F, fl));
    hcheck(fl, F, "mtgpu_scan_frames", "flags (n_frames bytes)");
  }

  /* 7. mtgpu_device_pci_address: buf = cap bytes, 13 are enough */
  {
    char *buf = halloc(13);
    MT(mtgpu_device_pci_address(0, buf, 13));
    if (strlen(buf) != 12) { fprintf(stderr, "FAIL pci address '%s'\n", buf); ++failures; }
    hcheck(buf, 13, "mtgpu_device_pci_address", "buf (13 bytes)");
  }

  /* 8. mtgpu_gather_segments: d_recv = n_ranks * bytes_per_rank (one rank on a one-GPU box) */
  {
    unsigned char id[MTGPU_UNIQUE_ID_BYTES];
    mtgpu_comm *comm = NULL;
    const uint64_t bytes = 1064;
    if (mtgpu_comm_unique_id(id) == MT_OK && mtgpu_comm_create(0, 1, id, 0, &comm) == MT_OK) {
      dbuf d_send = dalloc(bytes, "d_send");
      dbuf d_recv = dalloc(bytes * 1, "d_recv (n_ranks * bytes_per_rank)");
      HIP(hipMemset(d_send.p, 0x3c, bytes));
      MT(mtgpu_gather_segments(comm, d_send.p, bytes, d_recv.p, st));
      HIP(hipStreamSynchronize(st));
      unsigned char back[1064];
      HIP(hipMemcpy(back, d_recv.p, bytes, hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < bytes; ++i)
        if (back[i] != 0x3c) { fprintf(stderr, "FAIL gather: byte %llu\n", (unsigned long long)i); ++failures; break; }
      dcheck(&d_send, "mtgpu_gather_segments");
      dcheck(&d_recv, "mtgpu_gather_segments");
      mtgpu_comm_destroy(comm);
      printf("gather: checked\n");
    } else {
      printf("gather: skipped (%s)\n", mtgpu_last_error());
    }
  }

  HIP(hipStreamDestroy(st));
  mtgpu_destroy(ctx);
  free(mv); free(off); free(pts);
  if (failures) { fprintf(stderr, "%d buffer contract(s) violated\n", failures); return 1; }
  printf("all caller-sized buffers respected\n");
  return 0;
}
