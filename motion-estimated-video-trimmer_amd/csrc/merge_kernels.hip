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
  // stream_off == NULL: ONE stream that spans [0, n_frames_total)
  unsigned long long a = stream_off ? stream_off[s] : 0ull, b = stream_off ? stream_off[s + 1] : n_frames_total;
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
      // same answer as the multi-workgroup path (ml_result): nothing of a NaN list is reported
      res->n_timestamps = 0; res->n_segments = 0; res->time_removed = 0.0; res->saved_pct = 0.0;
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

// ---------------------------------------------------------------------------------------
// Large single stream (mtgpu_merge_segments / mtgpu_merge_timestamps_device with n above
// MERGE_LARGE_MIN): the same a8 + a9 arithmetic spread over many workgroups, for pooled
// timestamp lists of 10^5 .. 10^7 entries (a day of footage analysed at 10 fps is ~10^6).
//   sort    ml_tile_sort (bitonic in LDS, 2048-element tiles) + log2(n / 2048) ml_merge_pass
//           launches (merge path: every thread finds its diagonal by binary search and merges
//           4 outputs)                                                    (pipeline.cpp:302)
//   unique  never materialised: on the sorted list, element i survives std::unique iff
//           ts[i] != ts[i-1]; its predecessor ts[i-1] is then the previous DISTINCT value, so
//           segment starts / ends are decided from neighbours in place          (:303-304)
//   merge   ml_mark (per-tile counts of survivors and segment starts) -> ml_scan (one
//           workgroup, exclusive scan of the tile counts) -> ml_emit (the element that starts
//           segment k writes its raw start, the element that ends it its raw end) ->
//           ml_finalize (padding, clamp, per-segment duration)            (:328-344, 351-353)
//   sum     ml_result: ONE lane adds the durations in segment order (bit-exact with the
//           sequential reference, :353), then savings / cut decision      (:355-358, 387-388)
// All double arithmetic is the same plain IEEE add / sub / compare as in merge_streams_kernel.

namespace {

constexpr int LT = 2048;   // sort tile (elements)
constexpr int LB = 512;    // threads of the sort / merge kernels
constexpr int LV = LT / LB;

struct LargeCtl {           // device-side control block of one large merge
  unsigned int status;      // bit0: a NaN timestamp was seen
  unsigned int pad;
  unsigned long long n_unique;   // M
  unsigned long long n_seg;      // K
};

__global__ __launch_bounds__(LB) void ml_tile_sort(const double *__restrict__ in, double *__restrict__ out,
                                                    unsigned long long n, LargeCtl *ctl) {
  __shared__ double s[LT];
  const int tid = threadIdx.x;
  const unsigned long long base = (unsigned long long)blockIdx.x * LT;
  bool nan = false;
  for (int i = tid; i < LT; i += LB) {
    const unsigned long long g = base + i;
    const double v = g < n ? in[g] : __builtin_inf();     // padding sorts to the end
    nan |= (v != v);
    s[i] = v;
  }
  if (nan) atomicOr(&ctl->status, 1u);
  __syncthreads();
  for (int k = 2; k <= LT; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < LT / 2; t += LB) {
        const int i = 2 * t - (t & (j - 1));              // index with a 0 bit at position log2(j)
        const int p = i + j;
        const bool up = (i & k) == 0;
        const double a = s[i], b = s[p];
        if (up ? (b < a) : (a < b)) { s[i] = b; s[p] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < LT; i += LB) {
    const unsigned long long g = base + i;
    if (g < n) out[g] = s[i];
  }
}

// One pass: runs of `run` sorted elements (run is a multiple of LT) are merged pairwise.
__global__ __launch_bounds__(LB) void ml_merge_pass(const double *__restrict__ src, double *__restrict__ dst,
                                                     unsigned long long n, unsigned long long run) {
  const unsigned long long out0 = (unsigned long long)blockIdx.x * LT;
  const unsigned long long a0 = (out0 / (2ull * run)) * 2ull * run;
  const unsigned long long a1 = min(n, a0 + run), b1 = min(n, a0 + 2ull * run);
  const unsigned long long na = a1 - a0, nb = b1 - a1;
  const double *A = src + a0, *B = src + a1;
  const unsigned long long d0 = (out0 - a0) + (unsigned long long)threadIdx.x * LV;
  if (d0 >= na + nb) return;
  // merge path: i = how many of the first d0 outputs come from A (ties: A first)
  unsigned long long lo = d0 > nb ? d0 - nb : 0ull, hi = min(d0, na);
  while (lo < hi) {
    const unsigned long long mid = (lo + hi) >> 1;
    if (A[mid] <= B[d0 - 1ull - mid]) lo = mid + 1ull; else hi = mid;
  }
  unsigned long long i = lo, j = d0 - lo;
  double *o = dst + a0 + d0;
#pragma unroll
  for (int v = 0; v < LV; ++v) {
    if (d0 + (unsigned long long)v >= na + nb) break;
    const bool take_a = (j >= nb) || (i < na && A[i] <= B[j]);
    o[v] = take_a ? A[i++] : B[j++];
  }
}

// survivors of std::unique and segment starts, decided from the sorted neighbours
__device__ __forceinline__ void classify(const double *__restrict__ ts, unsigned long long n, unsigned long long i,
                                         double gap, bool &keep, bool &start, bool &end) {
  keep = start = end = false;
  if (i >= n) return;
  const double c = ts[i];
  const double p = i > 0 ? ts[i - 1] : 0.0;
  keep = (i == 0) || !(c == p);                                   // :303-304 (operator==)
  start = keep && (i == 0 || (c - p > gap));                      // :332-333, last_act == previous distinct value
  const bool last_of_run = (i + 1 == n) || !(ts[i + 1] == c);
  end = last_of_run && (i + 1 == n || (ts[i + 1] - c > gap));
}

__global__ __launch_bounds__(MB) void ml_mark(const double *__restrict__ ts, unsigned long long n,
                                               const mt_merge_params *__restrict__ mp,
                                               unsigned int *__restrict__ tile_keep, unsigned int *__restrict__ tile_start) {
  __shared__ Shared sh;
  const unsigned long long i = (unsigned long long)blockIdx.x * MB + threadIdx.x;
  bool keep, start, end;
  classify(ts, n, i, mp->max_gap_sec, keep, start, end);
  unsigned int nk, ns;
  (void)block_excl(keep, sh, &nk);
  (void)block_excl(start, sh, &ns);
  if (threadIdx.x == 0) { tile_keep[blockIdx.x] = nk; tile_start[blockIdx.x] = ns; }
}

// exclusive scan of the per-tile counts (one workgroup), totals into ctl
__global__ __launch_bounds__(MB) void ml_scan(unsigned int *tile_keep, unsigned int *tile_start, unsigned int n_tiles,
                                               unsigned long long *start_base, LargeCtl *ctl) {
  __shared__ unsigned long long wsum[MW];
  __shared__ unsigned long long carry_s, carry_k;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { carry_s = 0ull; carry_k = 0ull; }
  __syncthreads();
  for (unsigned int base = 0; base < n_tiles; base += MB) {
    const unsigned int t = base + tid;
    const unsigned long long vs = t < n_tiles ? tile_start[t] : 0u;
    const unsigned long long vk = t < n_tiles ? tile_keep[t] : 0u;
    // inclusive wave scan of vs
    unsigned long long x = vs;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    unsigned long long wbase = 0ull, tot = 0ull;
#pragma unroll
    for (int w = 0; w < MW; ++w) { if (w < wave) wbase += wsum[w]; tot += wsum[w]; }
    if (t < n_tiles) start_base[t] = carry_s + wbase + x - vs;
    __syncthreads();
    // total of vk over the chunk
    unsigned long long k = vk;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) k += __shfl_xor(k, d);
    if (lane == 0) wsum[wave] = k;
    __syncthreads();
    if (tid == 0) {
      unsigned long long kk = 0ull;
      for (int w = 0; w < MW; ++w) kk += wsum[w];
      carry_k += kk;
      carry_s += tot;
    }
    __syncthreads();
  }
  if (tid == 0) { ctl->n_unique = carry_k; ctl->n_seg = carry_s; }
}

__global__ __launch_bounds__(MB) void ml_emit(const double *__restrict__ ts, unsigned long long n,
                                               const mt_merge_params *__restrict__ mp,
                                               const unsigned long long *__restrict__ start_base,
                                               double *__restrict__ raw_start, double *__restrict__ raw_end) {
  __shared__ Shared sh;
  const unsigned long long i = (unsigned long long)blockIdx.x * MB + threadIdx.x;
  bool keep, start, end;
  classify(ts, n, i, mp->max_gap_sec, keep, start, end);
  unsigned int tot;
  const unsigned int ex = block_excl(start, sh, &tot);
  const unsigned long long kb = start_base[blockIdx.x] + ex;        // segments started before element i
  if (start) raw_start[kb] = ts[i];
  if (end) raw_end[kb + (start ? 1ull : 0ull) - 1ull] = ts[i];      // the segment whose start is the latest one <= i
}

__global__ __launch_bounds__(256) void ml_finalize(const mt_merge_params *__restrict__ mp, const LargeCtl *ctl,
                                                    const double *__restrict__ raw_start, double *raw_end_durs,
                                                    mt_segment *seg, unsigned long long seg_cap) {
  const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
  if (k >= ctl->n_seg) return;
  const double pad = mp->padding_sec, dur = mp->duration;
  double st = std_max(0.0, raw_start[k] - pad);   // :337 / :343
  double en = raw_end_durs[k] + pad;              // :338 / :344
  en = std_min(en, dur);                          // :351
  st = std_min(st, en);                           // :352
  if (k < seg_cap) { seg[k].start = st; seg[k].end = en; }
  raw_end_durs[k] = en - st;                      // :353 summand (in place of the raw end)
}

__global__ __launch_bounds__(MB) void ml_result(const mt_merge_params *__restrict__ mp, const LargeCtl *ctl,
                                                 const double *__restrict__ durs, int job_semantics,
                                                 mt_segment *seg, unsigned long long seg_cap, mt_merge_result *res) {
  __shared__ double stage[MB];
  const int tid = threadIdx.x;
  if (ctl->status & 1u) {                          // NaN timestamp: outside the defined domain
    if (tid == 0) {
      res->n_timestamps = 0; res->n_segments = 0; res->time_removed = 0.0; res->saved_pct = 0.0;
      res->do_cut = -1; res->status = MT_ERR_INVALID;
    }
    return;
  }
  const unsigned long long K = ctl->n_seg;
  double out_dur = 0.0;
  for (unsigned long long base = 0; base < K; base += MB) {       // :349-354, one lane adds in segment order
    __syncthreads();
    if (base + tid < K) stage[tid] = durs[base + tid];
    __syncthreads();
    if (tid == 0) {
      const unsigned int m = (unsigned int)((K - base < MB) ? (K - base) : MB);
      for (unsigned int q = 0; q < m; ++q) out_dur += stage[q];
    }
  }
  if (tid == 0) {
    const double dur = mp->duration;
    const double removed = dur - out_dur;
    const double pct = (dur > 0) ? removed / dur * 100.0 : 0.0;
    const int cut = (pct > mp->min_savings_pct) ? 1 : 0;
    unsigned long long nseg = K;
    if (job_semantics && !cut) {
      if (seg_cap >= 1) { seg[0].start = 0.0; seg[0].end = dur; }
      nseg = 1;
    }
    res->n_timestamps = ctl->n_unique; res->n_segments = nseg; res->time_removed = removed;
    res->saved_pct = pct; res->do_cut = cut; res->status = MT_OK;
  }
}

}  // namespace

size_t merge_large_ws_bytes(unsigned long long n) {
  const unsigned long long tiles = (n + MB - 1) / MB;
  return sizeof(double) * 3ull * n + sizeof(unsigned int) * 2ull * tiles + sizeof(unsigned long long) * tiles + 64ull + 64ull;
}

// n >= 1.  ws: merge_large_ws_bytes(n) bytes, 16-byte aligned.  d_mp: ONE mt_merge_params on the device.
hipError_t launch_merge_large(const double *d_ts, unsigned long long n, const mt_merge_params *d_mp, int job_semantics,
                              void *ws, mt_segment *d_seg, unsigned long long seg_cap, mt_merge_result *d_res,
                              hipStream_t st) {
  if (n == 0 || n > (1ull << 40)) return hipErrorInvalidValue;
  unsigned char *w = static_cast<unsigned char *>(ws);
  LargeCtl *ctl = reinterpret_cast<LargeCtl *>(w);                      w += 64;
  double *A = reinterpret_cast<double *>(w);                           w += sizeof(double) * n;
  double *B = reinterpret_cast<double *>(w);                           w += sizeof(double) * n;
  double *C = reinterpret_cast<double *>(w);                           w += sizeof(double) * n;
  const unsigned long long tiles = (n + MB - 1) / MB;
  unsigned long long *start_base = reinterpret_cast<unsigned long long *>(w);   w += sizeof(unsigned long long) * tiles;
  unsigned int *tile_keep = reinterpret_cast<unsigned int *>(w);       w += sizeof(unsigned int) * tiles;
  unsigned int *tile_start = reinterpret_cast<unsigned int *>(w);
  hipError_t e = hipMemsetAsync(ctl, 0, 64, st);
  if (e != hipSuccess) return e;
  const unsigned long long sort_tiles = (n + LT - 1) / LT;
  if (sort_tiles > 0x7fffffffull || tiles > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ml_tile_sort, dim3((unsigned int)sort_tiles), dim3(LB), 0, st, d_ts, A, n, ctl);
  double *src = A, *dst = B;
  for (unsigned long long run = LT; run < n; run <<= 1) {
    hipLaunchKernelGGL(ml_merge_pass, dim3((unsigned int)sort_tiles), dim3(LB), 0, st, src, dst, n, run);
    double *t = src; src = dst; dst = t;
  }
  // src = sorted list; dst (the other ping-pong buffer) and C are free: raw starts / raw ends + durations
  hipLaunchKernelGGL(ml_mark, dim3((unsigned int)tiles), dim3(MB), 0, st, src, n, d_mp, tile_keep, tile_start);
  hipLaunchKernelGGL(ml_scan, dim3(1), dim3(MB), 0, st, tile_keep, tile_start, (unsigned int)tiles, start_base, ctl);
  hipLaunchKernelGGL(ml_emit, dim3((unsigned int)tiles), dim3(MB), 0, st, src, n, d_mp, start_base, dst, C);
  const unsigned long long fin_blocks = (n + 255) / 256;               // K <= n (K itself lives on the device)
  hipLaunchKernelGGL(ml_finalize, dim3((unsigned int)fin_blocks), dim3(256), 0, st, d_mp, ctl, dst, C, d_seg, seg_cap);
  hipLaunchKernelGGL(ml_result, dim3(1), dim3(MB), 0, st, d_mp, ctl, C, job_semantics, d_seg, seg_cap, d_res);
  return hipGetLastError();
}

// One parameter block by value -> device memory (kernel arguments are captured at launch, so the
// caller's block may live on its stack; an async copy from pageable memory gives no such promise).
__global__ void store_merge_params_kernel(mt_merge_params v, mt_merge_params *dst) { *dst = v; }

hipError_t launch_store_params(const mt_merge_params &v, mt_merge_params *d_dst, hipStream_t st) {
  hipLaunchKernelGGL(store_merge_params_kernel, dim3(1), dim3(1), 0, st, v, d_dst);
  return hipGetLastError();
}

hipError_t launch_merge(const MergeLaunch &L) {
  if (L.n_streams == 0) return hipSuccess;
  if (!L.stream_off && L.n_streams != 1) return hipErrorInvalidValue;
  hipLaunchKernelGGL(merge_streams_kernel, dim3(L.n_streams), dim3(MB), 0, L.stream, L.flags,
                     L.pts, L.stream_off, L.n_frames_total, L.mp, L.job_semantics, L.ts_ws, L.seg,
                     L.seg_cap, L.res);
  return hipGetLastError();
}

}  // namespace mtgpu
