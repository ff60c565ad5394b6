// merge_kernels.hip — gfx950 kernel for the timestamp pooling + gap-bounded
// segment merge (reference: src/motion_scanner.cpp:382-383 push_back(pts);
// src/pipeline.cpp:302-304 sort+unique; :323-346 merge; :349-358 clamp, savings,
// cut decision; :387-388 full-copy segment).
//
// One 1024-thread workgroup per stream.  All double arithmetic is plain IEEE
// add/sub/compare/div (built with -ffp-contract=off); the only order-dependent
// sum (out_dur, :353) is accumulated by one lane in segment order, so results are
// bit-identical to the sequential reference.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "merge_kernels.h"

namespace mtgpu {

namespace {

constexpr int MB = 1024;  // workgroup size
constexpr int MW = MB / 64;

// std::max(a,b) / std::min(a,b) exactly as libstdc++ defines them (they differ
// from fmax/fmin on signed zeros): max(a,b) = (a<b)?b:a, min(a,b) = (b<a)?b:a.
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double std_min(double a, double b) { return (b < a) ? b : a; }

struct Shared {
  unsigned long long running;     // elements emitted so far by a chunked pass
  long long carry_start;          // index of the last segment start seen in earlier chunks
  unsigned int wave_cnt[MW];
  long long wave_last[MW];
  unsigned int cond;              // bit0 NaN, bit1 out of order, bit2 duplicates
  double stage[MB];
};

// Exclusive prefix of `flag` over the workgroup; *total = number of set flags.
__device__ __forceinline__ unsigned int block_excl(bool flag, Shared &sh, unsigned int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long b = __ballot(flag);
  if (lane == 0) sh.wave_cnt[wave] = (unsigned int)__popcll(b);
  __syncthreads();
  unsigned int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < MW; ++w) {
    const unsigned int c = sh.wave_cnt[w];
    base += (w < wave) ? c : 0u;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + (unsigned int)__popcll(b & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ void cmpswap(double *ts, unsigned long long i, unsigned long long j) {
  const double a = ts[i], b = ts[j];
  if (b < a) { ts[i] = b; ts[j] = a; }
}

// In-place ascending sort of ts[0..M) by one workgroup: bitonic network with the
// "flip" first step (all comparators ascending), indices >= M act as +inf.
__device__ void block_bitonic_sort(double *ts, unsigned long long M) {
  int lgP = 0;
  while ((1ull << lgP) < M) ++lgP;
  const unsigned long long halfP = (lgP > 0) ? (1ull << (lgP - 1)) : 0ull;
  for (int lgk = 1; lgk <= lgP; ++lgk) {
    const int lgh = lgk - 1;
    const unsigned long long k = 1ull << lgk, half = 1ull << lgh;
    for (unsigned long long p = threadIdx.x; p < halfP; p += MB) {
      const unsigned long long blk = p >> lgh, off = p & (half - 1);
      const unsigned long long i = (blk << lgk) + off, j = (blk << lgk) + (k - 1 - off);
      if (j < M) cmpswap(ts, i, j);
    }
    __syncthreads();
    for (int lgj = lgk - 2; lgj >= 0; --lgj) {
      const unsigned long long jj = 1ull << lgj;
      for (unsigned long long p = threadIdx.x; p < halfP; p += MB) {
        const unsigned long long i = ((p >> lgj) << (lgj + 1)) + (p & (jj - 1));
        const unsigned long long j = i + jj;
        if (j < M) cmpswap(ts, i, j);
      }
      __syncthreads();
    }
  }
}

}  // namespace

__global__ __launch_bounds__(MB) void merge_streams_kernel(
    const unsigned char *__restrict__ flags, const double *__restrict__ pts,
    const unsigned long long *__restrict__ stream_off, unsigned long long n_frames_total,
    const mt_merge_params *__restrict__ mp_arr, int job_semantics, double *ts_ws,
    mt_segment *seg_all, unsigned long long seg_cap, mt_merge_result *res_all) {
  __shared__ Shared sh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned int s = blockIdx.x;
  unsigned long long a = stream_off[s], b = stream_off[s + 1];
  b = b < n_frames_total ? b : n_frames_total;
  a = a < b ? a : b;
  const mt_merge_params mp = mp_arr[s];
  double *ts = ts_ws + 2ull * a;          // [b-a] compacted timestamps
  double *durs = ts + (b - a);            // [b-a] per-segment (end - start)
  mt_segment *seg = seg_all + (unsigned long long)s * seg_cap;
  mt_merge_result *res = res_all + s;

  if (tid == 0) { sh.running = 0; sh.cond = 0; }
  __syncthreads();

  // ---- A: pool the motion timestamps (scan_range: if (has_motion) ts.push_back(pts))
  for (unsigned long long base = a; base < b; base += MB) {
    const unsigned long long i = base + tid;
    const bool f = (i < b) && (flags ? (flags[i] != 0) : true);
    const double v = f ? pts[i] : 0.0;
    unsigned int tot;
    const unsigned int ex = block_excl(f, sh, &tot);
    if (f) {
      ts[sh.running + ex] = v;
      if (v != v) atomicOr(&sh.cond, 1u);
    }
    __syncthreads();
    if (tid == 0) sh.running += tot;
    __syncthreads();
  }
  unsigned long long M = sh.running;

  // ---- B: already sorted / unique?  (frames normally arrive in pts order)
  for (unsigned long long i = tid + 1ull; i < M; i += MB) {
    const double p = ts[i - 1], c = ts[i];
    if (c < p) atomicOr(&sh.cond, 2u);
    else if (c == p) atomicOr(&sh.cond, 4u);
  }
  __syncthreads();
  const unsigned int cond = sh.cond;
  if (cond & 1u) {  // NaN timestamp: outside the defined domain
    if (tid == 0) {
      res->n_timestamps = M; res->n_segments = 0; res->time_removed = 0.0; res->saved_pct = 0.0;
      res->do_cut = -1; res->status = MT_ERR_INVALID;
    }
    return;
  }

  // ---- C: std::sort (pipeline.cpp:302)
  if (cond & 2u) block_bitonic_sort(ts, M);

  // ---- D: std::unique (pipeline.cpp:303-304), in place, order preserving
  if (cond & 6u) {
    __syncthreads();
    if (tid == 0) sh.running = 0;
    __syncthreads();
    for (unsigned long long base = 0; base < M; base += MB) {
      const unsigned long long i = base + tid;
      const bool in = i < M;
      const double c = in ? ts[i] : 0.0;
      const double p = (in && i > 0) ? ts[i - 1] : 0.0;   // original predecessor: reads precede
      const bool keep = in && (i == 0 || !(c == p));       // this chunk's writes (barrier below)
      unsigned int tot;
      const unsigned int ex = block_excl(keep, sh, &tot);  // contains __syncthreads
      // writes land at positions <= i and never on ts[base+MB-1 ..], whose old value the next
      // chunk reads as predecessor, unless nothing was dropped (then the value is unchanged)
      if (keep) ts[sh.running + ex] = c;
      __syncthreads();
      if (tid == 0) sh.running += tot;
      __syncthreads();
    }
    M = sh.running;
  }

  if (M == 0) {  // pipeline.cpp:308-319 — "No motion found": no segments, no job
    if (tid == 0) {
      res->n_timestamps = 0; res->n_segments = 0; res->time_removed = 0.0; res->saved_pct = 0.0;
      res->do_cut = -1; res->status = MT_OK;
    }
    return;
  }

  // ---- E: gap-bounded merge (pipeline.cpp:328-344) + clamp (:351-352)
  __syncthreads();
  if (tid == 0) { sh.running = 0; sh.carry_start = 0; }
  __syncthreads();
  const double gap = mp.max_gap_sec, pad = mp.padding_sec, dur = mp.duration;
  for (unsigned long long base = 0; base < M; base += MB) {
    const unsigned long long i = base + tid;
    const bool in = i < M;
    const double c = in ? ts[i] : 0.0;
    // a new segment starts at i iff i == 0 or ts[i] - ts[i-1] > MAX_GAP (:332-333; last_act == ts[i-1])
    const bool is_start = in && (i == 0 || (c - ts[i - 1] > gap));
    const bool is_end = in && (i + 1 == M || (ts[i + 1] - c > gap));
    const unsigned long long bs = __ballot(is_start);
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    if (lane == 0) {
      sh.wave_cnt[wave] = (unsigned int)__popcll(bs);
      sh.wave_last[wave] = bs ? (long long)(base + wave * 64 + (63 - __clzll(bs))) : -1ll;
    }
    __syncthreads();
    unsigned int kbase = 0, tot = 0;
    long long prev_last = sh.carry_start;
#pragma unroll
    for (int w = 0; w < MW; ++w) {
      const unsigned int cw = sh.wave_cnt[w];
      if (w < wave) { kbase += cw; if (sh.wave_last[w] >= 0) prev_last = sh.wave_last[w]; }
      tot += cw;
    }
    if (is_end) {
      const unsigned long long mine = bs & le;
      const long long sidx = mine ? (long long)(base + wave * 64 + (63 - __clzll(mine))) : prev_last;
      const unsigned long long k = sh.running + kbase + (unsigned int)__popcll(mine) - 1ull;
      double st = std_max(0.0, ts[sidx] - pad);   // :337 / :343
      double en = c + pad;                        // :338 / :344
      en = std_min(en, dur);                      // :351
      st = std_min(st, en);                       // :352
      if (k < seg_cap) { seg[k].start = st; seg[k].end = en; }
      durs[k] = en - st;                          // :353 summand
    }
    __syncthreads();
    if (tid == 0) {
      sh.running += tot;
      for (int w = MW - 1; w >= 0; --w)
        if (sh.wave_last[w] >= 0) { sh.carry_start = sh.wave_last[w]; break; }
    }
    __syncthreads();
  }
  const unsigned long long K = sh.running;

  // ---- F: out_dur += (end - start) in segment order (:349-354), one lane adds
  double out_dur = 0.0;
  for (unsigned long long base = 0; base < K; base += MB) {
    __syncthreads();
    if (base + tid < K) sh.stage[tid] = durs[base + tid];
    __syncthreads();
    if (tid == 0) {
      const unsigned int n = (unsigned int)((K - base < MB) ? (K - base) : MB);
      for (unsigned int q = 0; q < n; ++q) out_dur += sh.stage[q];
    }
  }

  // ---- G: savings + cut decision (:355-358, 387-388)
  if (tid == 0) {
    const double removed = dur - out_dur;
    const double pct = (dur > 0) ? removed / dur * 100.0 : 0.0;
    const int cut = (pct > mp.min_savings_pct) ? 1 : 0;
    unsigned long long nseg = K;
    if (job_semantics && !cut) {
      if (seg_cap >= 1) { seg[0].start = 0.0; seg[0].end = dur; }
      nseg = 1;
    }
    res->n_timestamps = M; res->n_segments = nseg; res->time_removed = removed;
    res->saved_pct = pct; res->do_cut = cut; res->status = MT_OK;
  }
}

hipError_t launch_merge(const MergeLaunch &L) {
  if (L.n_streams == 0) return hipSuccess;
  hipLaunchKernelGGL(merge_streams_kernel, dim3(L.n_streams), dim3(MB), 0, L.stream, L.flags,
                     L.pts, L.stream_off, L.n_frames_total, L.mp, L.job_semantics, L.ts_ws, L.seg,
                     L.seg_cap, L.res);
  return hipGetLastError();
}

}  // namespace mtgpu
