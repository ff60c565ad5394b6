// scan_kernels.hip — gfx950 kernels for MotionScanner::check_frame
// (reference: src/motion_scanner.cpp:217-295).  Integer threshold / scatter /
// stencil work: HBM-read bound, no MFMA.
//
// Work item = (frame, row band).  One workgroup per item:
//   phase 0  zero the band's vote counters in LDS                  (:229 memset)
//   phase 1  stream the frame's packed 40-byte records, one record per lane and
//            load (bytes 4..15 of each record = w,h,src_x,src_y,dst_x,dst_y),
//            threshold on |d|^2, map dst to a cell, vote in LDS     (:242-268)
//   phase 2  per chunk of rows:
//     2a     one wave per (row, 64-cell word): `count >= vectors_needed`
//            -> __ballot -> 64-bit row masks in LDS                  (:282)
//     2b     one lane per (row, word): shifted-mask 4-neighbour test,
//            __popcll, LDS reduction                                 (:277-293)
//   compare the centre count with max(1, clusters_needed)           (:288)
//
// Vote counters come in three forms (template FB = bits per cell, MODE):
//   ADD32    FB = 32, plain fire-and-forget `ds_add_u32`.  The reference's u8 saturation at
//            255 (:265-266) is unobservable (only `>= vectors_needed`, vectors_needed <= 255,
//            is tested), so 32-bit counts are bit-exact.  Used whenever the tile fits LDS.
//   UNARY    FB = 1/2/4/8 >= vectors_needed: each field is a thermometer code of
//            min(votes, vectors_needed).  A vote reads the field, then sets the first clear
//            bit with a returning `ds_or_rtn_b32`; if another lane set that bit first it
//            moves to the next one.  No retry loop on contention (same-word ORs are
//            serialised by the LDS unit), at most vectors_needed ORs per vote, and a
//            saturated field costs one plain read.  cell active <=> bit vectors_needed-1 set.
//   CAS8     FB = 8 binary field saturating at vectors_needed (9..255) via compare-and-swap.
//            Only for unusually large VECTORS_NEEDED on grids too big for 32-bit counters.
//   Packed forms need 32x..4x less LDS, so big grids (960x540) stay in one LDS tile and
//   every record is read from HBM once.
// The early `return true` (:288-289) does not change the value:
// result = (#centre cells >= max(1, clusters_needed)).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_kernels.h"

namespace mtgpu {

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Bytes 4..15 of a record: d.x = w | h<<8 | src_x<<16, d.y = src_y | dst_x<<16,
// d.z = dst_y | pad<<16   (layout: include/mt_types.h, mt_mv).
typedef u32x4 u32x4_a8 __attribute__((aligned(8)));

// VAR bit1: 16-byte load of bytes 0..15 instead of 12 bytes at +4; bit2: default cache policy
// instead of the streaming (nt) hint.  Experiment knobs (MTGPU_VARIANT), results in DESIGN.md.
template <int VAR>
__device__ __forceinline__ u32x3 load_fields(const unsigned char *rec) {
  if constexpr ((VAR & 2) != 0) {
    u32x4 q;
    if constexpr ((VAR & 4) != 0) q = *reinterpret_cast<const u32x4_a8 *>(rec);
    else q = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a8 *>(rec));
    return (u32x3){q.y, q.z, q.w};
  } else {
    if constexpr ((VAR & 4) != 0) return *reinterpret_cast<const u32x3_a4 *>(rec + 4);
    else return __builtin_nontemporal_load(reinterpret_cast<const u32x3_a4 *>(rec + 4));
  }
}

enum { MODE_ADD32 = 0, MODE_UNARY = 1, MODE_CAS = 2 };

template <int FB, int MODE>
__device__ __forceinline__ void bump(unsigned int *cnt, unsigned int cell, unsigned int cap) {
  if constexpr (MODE == MODE_ADD32) {
    atomicAdd(&cnt[cell], 1u);
  } else {
    constexpr unsigned int FM = (FB >= 32) ? 0xffffffffu : ((1u << FB) - 1u);
    const unsigned int bit = cell * FB;
    unsigned int *w = &cnt[bit >> 5];
    const unsigned int sh = bit & 31u;
    unsigned int f = (*w >> sh) & FM;                        // once saturated: a plain read
    if constexpr (MODE == MODE_UNARY) {
      unsigned int j = (unsigned int)__popc(f);              // thermometer: bits 0..j-1 are set
      while (j < cap) {
        const unsigned int old = atomicOr(w, 1u << (sh + j));
        f = (old >> sh) & FM;
        if (((f >> j) & 1u) == 0u) break;                    // this lane set bit j: vote counted
        j = (unsigned int)__popc(f);                         // somebody else did: next clear bit
      }
    } else {
      unsigned int old = *w;
      while (((old >> sh) & FM) < cap) {
        const unsigned int seen = atomicCAS(w, old, old + (1u << sh));
        if (seen == old) break;
        old = seen;
      }
    }
  }
}

template <int FB>
__device__ __forceinline__ unsigned int count_of(const unsigned int *cnt, unsigned int cell) {
  if constexpr (FB == 32) {
    return cnt[cell];
  } else {
    const unsigned int bit = cell * FB;
    return (cnt[bit >> 5] >> (bit & 31u)) & ((1u << FB) - 1u);
  }
}

// Field-wise saturating sum of two packed counter words (slice combine).
template <int FB, int MODE>
__device__ __forceinline__ unsigned int combine_words(unsigned int a, unsigned int b, unsigned int cap) {
  if constexpr (MODE == MODE_ADD32) {
    return a + b;
  } else {
    if (b == 0u) return a;
    if (a == 0u) return b;
    constexpr unsigned int FM = (1u << FB) - 1u;
    unsigned int r = 0u;
#pragma unroll
    for (int sh = 0; sh < 32; sh += FB) {
      const unsigned int fa = (a >> sh) & FM, fb = (b >> sh) & FM;
      if constexpr (MODE == MODE_UNARY) {
        const unsigned int c = min(cap, (unsigned int)__popc(fa) + (unsigned int)__popc(fb));
        r |= ((1u << c) - 1u) << sh;
      } else {
        r |= min(cap, fa + fb) << sh;
      }
    }
    return r;
  }
}

// Threshold + cell mapping + vote for one record (src/motion_scanner.cpp:246-267).
template <int FB, int MODE>
__device__ __forceinline__ void vote(const u32x3 d, const ScanK &k, int t0, int t1,
                                     unsigned int *cnt) {
  const int src_x = (int)d.x >> 16;
  const int src_y = (int)(short)(d.y & 0xffffu);
  const int dst_x = (int)d.y >> 16;
  const int dst_y = (int)(short)(d.z & 0xffffu);
  const unsigned int dx = (unsigned int)(dst_x - src_x);   // |dx| <= 65535
  const unsigned int dy = (unsigned int)(dst_y - src_y);
  // dx*dx < 2^32 exactly; the sum needs 34 bits.
  const unsigned long long mag =
      (unsigned long long)(dx * dx) + (unsigned long long)(dy * dy);
  const int gx = dst_x >> k.shift;
  const int gy = dst_y >> k.shift;
  const bool in = (mag >= k.thr) & (gx >= 0) & (gx < k.gw) & (gy >= k.y_lo) & (gy < k.y_hi) &
                  (gy >= t0) & (gy < t1);
  if (in) bump<FB, MODE>(cnt, (unsigned int)((gy - t0) * k.gw + gx), k.vec_need);
}

template <int BLOCK, int UNROLL, int FB, int MODE, int VAR>
__global__ __launch_bounds__(BLOCK) void scan_frames_kernel(
    const unsigned char *__restrict__ mv, unsigned long long n_records,
    const unsigned long long *__restrict__ frame_off, const unsigned char *__restrict__ has_sd,
    unsigned int item0, unsigned int n_frames, ScanK k, unsigned char *__restrict__ flags,
    unsigned int *__restrict__ frame_centres, unsigned int *slice_ws, unsigned int *tickets) {
  extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
  const int tid = threadIdx.x;
  const unsigned int item = item0 + blockIdx.x;
  // item -> (frame, band) or (frame, slice): bands and slices are never both > 1
  unsigned int f;
  int band = 0, slice = 0;
  if (k.bands > 1) {
    // Bands of one frame re-read the same records: put them on ONE XCD so that the later reader
    // hits that XCD's L2.  Workgroups b and b+8 share an XCD (round-robin dispatch; speed only,
    // never correctness): within a group of 8 frames, frame = item % 8, band = (item / 8) % bands.
    const unsigned int span = 8u * (unsigned int)k.bands;
    const unsigned int grp = item / span, l = item - grp * span;
    f = grp * 8u + (l & 7u);
    band = (int)(l >> 3);
    if (f >= n_frames) return;
  } else {
    f = item / (unsigned int)k.slices;
    slice = (int)(item - f * (unsigned int)k.slices);
  }

  unsigned long long r0 = frame_off[f], r1 = frame_off[f + 1];
  r1 = r1 < n_records ? r1 : n_records;
  r0 = r0 < r1 ? r0 : r1;
  const bool sd = has_sd ? (has_sd[f] != 0) : (r1 > r0);
  if (!sd) {                                   // :219-221 — no side data: false
    if (k.bands == 1 && slice == 0 && tid == 0) flags[f] = 0;
    return;                                    // (bands > 1: finalize kernel writes 0)
  }
  if (k.slices > 1) {                          // this workgroup's share of the frame's records
    const unsigned long long n = r1 - r0, per = (n + (unsigned long long)k.slices - 1ull) / (unsigned long long)k.slices;
    const unsigned long long a = r0 + min(n, per * (unsigned long long)slice);
    const unsigned long long b = r0 + min(n, per * (unsigned long long)(slice + 1));
    r0 = a;
    r1 = b;
  }

  // Band geometry: centres [c0,c1), tracked counter rows [t0,t1).
  const int c0 = k.y_lo + band * k.band_rows;
  const int c1 = min(k.y_hi, c0 + k.band_rows);
  const int t0 = max(c0 - 1, 0);
  const int t1 = min(c1 + 1, k.gh);
  const int trows = t1 - t0;                   // may be <= 0 for an empty analysed range
  const int W = k.W;

  unsigned int *cnt = lds;                                         // packed [trows][gw] fields
  unsigned long long *mask =
      reinterpret_cast<unsigned long long *>(lds + k.cnt_words);   // [chunk_rows+2][W]
  unsigned int *total = reinterpret_cast<unsigned int *>(mask + (size_t)k.mask_rows * W);
  unsigned int *ticket = total + 1;

  // ---- phase 0: zero counters and the centre total
  {
    u32x4 *c4 = reinterpret_cast<u32x4 *>(cnt);
    const int n4 = k.cnt_words >> 2;
    for (int i = tid; i < n4; i += BLOCK) c4[i] = (u32x4){0u, 0u, 0u, 0u};
    if (tid == 0) *total = 0u;
  }
  __syncthreads();

  // ---- phase 1: stream the records
  if (trows > 0 && k.vec_need != 0u) {         // vec_need == 0: every cell is active anyway
    const unsigned char *base = mv + r0 * 40ull;
    const unsigned long long n = r1 - r0;
    unsigned long long i = tid;
    constexpr unsigned long long STEP = (unsigned long long)UNROLL * BLOCK;
    constexpr unsigned long long LAST = (unsigned long long)(UNROLL - 1) * BLOCK;
    if constexpr ((VAR & 8) != 0) {
      // software-pipelined: the next batch of loads is issued before this batch is consumed
      u32x3 cur[UNROLL], nxt[UNROLL];
      bool have = i + LAST < n;
      if (have) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) cur[u] = load_fields<VAR>(base + (i + (unsigned long long)u * BLOCK) * 40ull);
      }
      while (have) {
        const unsigned long long j = i + STEP;
        const bool more = j + LAST < n;
        if (more) {
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) nxt[u] = load_fields<VAR>(base + (j + (unsigned long long)u * BLOCK) * 40ull);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vote<FB, MODE>(cur[u], k, t0, t1, cnt);
        if (more) {
#pragma unroll
          for (int u = 0; u < UNROLL; ++u) cur[u] = nxt[u];
        }
        i = j;
        have = more;
      }
    } else {
      // main body: UNROLL independent loads in flight per lane
      for (; i + LAST < n; i += STEP) {
        u32x3 d[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) d[u] = load_fields<VAR>(base + (i + (unsigned long long)u * BLOCK) * 40ull);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vote<FB, MODE>(d[u], k, t0, t1, cnt);
      }
    }
    for (; i < n; i += BLOCK) vote<FB, MODE>(load_fields<VAR>(base + i * 40ull), k, t0, t1, cnt);
  }
  __syncthreads();

  // ---- slices: publish this partial grid; the LAST workgroup of the frame to arrive sums them
  // (no workgroup ever waits for another: nothing to deadlock on).  Hand-off (Guideline 16, R1
  // in its counter form): 16-byte WRITE-THROUGH (sc1) stores, so no release fence and no L2
  // write-back -> every storing wave drains -> barrier -> one lane's agent-scope ticket add;
  // the last arriver does ONE agent-scope acquire (drops this CU's stale L1 lines: the
  // stream-ordered workspace is reused by later launches), then plain loads.
  if (k.slices > 1) {
    const size_t words = (size_t)k.cnt_words;
    unsigned int *mine = slice_ws + ((size_t)f * (size_t)k.slices + (size_t)slice) * words;
    {
      const u32x4 *c4 = reinterpret_cast<const u32x4 *>(cnt);
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(mine, 0, k.cnt_words * 4, 0x00020000);
      for (int i = tid; i < (k.cnt_words >> 2); i += BLOCK)
        __builtin_amdgcn_raw_buffer_store_b128(c4[i], rsrc, i * 16, 0, /*aux: sc1*/ 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      *ticket = __hip_atomic_fetch_add(&tickets[f], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*ticket != (unsigned int)(k.slices - 1)) return;       // not the last: done
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int s2 = 0; s2 < k.slices; ++s2) {
      if (s2 == slice) continue;
      const u32x4 *g4 = reinterpret_cast<const u32x4 *>(slice_ws + ((size_t)f * (size_t)k.slices + (size_t)s2) * words);
      u32x4 *c4 = reinterpret_cast<u32x4 *>(cnt);
      for (int i = tid; i < (k.cnt_words >> 2); i += BLOCK) {
        const u32x4 o = g4[i];
        u32x4 m = c4[i];
        m.x = combine_words<FB, MODE>(m.x, o.x, k.vec_need);
        m.y = combine_words<FB, MODE>(m.y, o.y, k.vec_need);
        m.z = combine_words<FB, MODE>(m.z, o.z, k.vec_need);
        m.w = combine_words<FB, MODE>(m.w, o.w, k.vec_need);
        c4[i] = m;
      }
    }
    __syncthreads();
  }

  // ---- phase 2: chunks of centre rows [c0+q0, c0+q0+qn); mask row j <-> grid row c0+q0-1+j
  const int crows = c1 - c0;
  unsigned int local = 0;
  for (int q0 = 0; q0 < crows; q0 += k.chunk_rows) {
    const int qn = min(k.chunk_rows, crows - q0);
    {  // 2a: activity masks, one wave per (mask row, word)
      const int lane = tid & 63, wave = tid >> 6;
      const int ntask = (qn + 2) * W;
      for (int t = wave; t < ntask; t += BLOCK / 64) {
        const int j = t / W, w = t - j * W;
        const int g = c0 + q0 - 1 + j;                 // grid row of this mask row
        const int x = w * 64 + lane;
        bool on = false;
        if (g >= t0 && g < t1 && x < k.gw)             // outside the grid = inactive
          on = count_of<FB>(cnt, (unsigned int)((g - t0) * k.gw + x)) >= k.active_min;
        const unsigned long long m = __ballot(on);
        if (lane == 0) mask[(size_t)j * W + w] = m;
      }
    }
    __syncthreads();
    {  // 2b: centre cells with an active 4-neighbour
      const int ntask = qn * W;
      for (int t = tid; t < ntask; t += BLOCK) {
        const int r = t / W, w = t - r * W;            // centre row c0+q0+r -> mask row r+1
        const unsigned long long *mr = mask + (size_t)(r + 1) * W;
        const unsigned long long m = mr[w];
        if (m == 0ull) continue;
        const unsigned long long up = mr[w - W], dn = mr[w + W];
        const unsigned long long lcarry = (w > 0) ? (mr[w - 1] >> 63) : 0ull;
        const unsigned long long rcarry = (w + 1 < W) ? (mr[w + 1] << 63) : 0ull;
        const unsigned long long nb = (m << 1) | lcarry | (m >> 1) | rcarry | up | dn;
        // centres are x in [1, gw-2]  (:280)
        const int lo = max(1 - w * 64, 0), hi = min(k.gw - 1 - w * 64, 64);   // bits [lo,hi)
        unsigned long long valid = 0ull;
        if (hi > lo) {
          valid = (hi >= 64) ? ~0ull : ((1ull << hi) - 1ull);
          valid &= ~((1ull << lo) - 1ull);
        }
        local += (unsigned int)__popcll(m & nb & valid);
      }
    }
    __syncthreads();                                   // masks are rewritten by the next chunk
  }
  if (local) atomicAdd(total, local);
  __syncthreads();

  if (tid == 0) {
    const unsigned int c = *total;
    if (k.bands == 1) flags[f] = (c >= k.clust_need) ? 1 : 0;
    else if (c) atomicAdd(&frame_centres[f], c);
  }
}

// bands > 1: combine the per-band centre counts (one thread per frame).
__global__ void finalize_flags_kernel(const unsigned long long *__restrict__ frame_off,
                                      unsigned long long n_records,
                                      const unsigned char *__restrict__ has_sd,
                                      const unsigned int *__restrict__ frame_centres,
                                      unsigned int n_frames, unsigned int clust_need,
                                      unsigned char *__restrict__ flags) {
  const unsigned int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  unsigned long long r0 = frame_off[f], r1 = frame_off[f + 1];
  r1 = r1 < n_records ? r1 : n_records;
  r0 = r0 < r1 ? r0 : r1;
  const bool sd = has_sd ? (has_sd[f] != 0) : (r1 > r0);
  flags[f] = (sd && frame_centres[f] >= clust_need) ? 1 : 0;
}

// Calibration only: a pure streaming read shaped like the scan (one workgroup per contiguous
// 1.25 MiB chunk, 512 threads, nt loads, 4 x 16 B in flight per lane, nothing else), folded into
// a value that is (almost) never stored.  bench.py reports its rate on the scan's own record
// buffer beside the 8 TB/s spec peak: what a kernel that ONLY reads can reach on this chip.
constexpr unsigned long long READ_CHUNK16 = (1280ull * 1024ull) / 16ull;

__global__ __launch_bounds__(512) void read_ceiling_kernel(const u32x4 *__restrict__ p, unsigned long long n16,
                                                           unsigned int *__restrict__ sink) {
  const unsigned long long b0 = (unsigned long long)blockIdx.x * READ_CHUNK16;
  const unsigned long long b1 = min(n16, b0 + READ_CHUNK16);
  unsigned long long i = b0 + threadIdx.x;
  u32x4 acc = (u32x4){0u, 0u, 0u, 0u};
  for (; i + 3ull * 512ull < b1; i += 4ull * 512ull) {
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(p + i + (unsigned long long)u * 512ull);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= v[u];
  }
  for (; i < b1; i += 512ull) acc ^= __builtin_nontemporal_load(p + i);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) *sink = acc.x;   // keeps the loads alive
}

hipError_t launch_read_ceiling(const void *p, unsigned long long bytes, unsigned int *sink, int cu_count,
                               hipStream_t stream) {
  (void)cu_count;
  const unsigned long long n16 = bytes / 16ull;
  if (n16 == 0) return hipSuccess;
  const unsigned long long blocks = (n16 + READ_CHUNK16 - 1) / READ_CHUNK16;
  if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
  hipLaunchKernelGGL(read_ceiling_kernel, dim3((unsigned int)blocks), dim3(512), 0, stream,
                     static_cast<const u32x4 *>(p), n16, sink);
  return hipGetLastError();
}

// ------------------------------------------------------------------ launchers

template <int BLOCK, int FB, int MODE, int UNROLL = 4, int VAR = 0>
static hipError_t launch_one(const ScanLaunch &L) {
  auto kern = scan_frames_kernel<BLOCK, UNROLL, FB, MODE, VAR>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_bytes);
  if (e != hipSuccess) return e;
  // bands > 1: frames are dealt in groups of 8 (same-XCD band placement), the last group is padded
  const unsigned long long items =
      L.k.bands > 1 ? (((unsigned long long)L.n_frames + 7ull) / 8ull) * 8ull * (unsigned long long)L.k.bands
                    : (unsigned long long)L.n_frames * (unsigned long long)L.k.slices;
  const unsigned long long chunk = L.item_chunk ? L.item_chunk : (1ull << 30);   // grid.x stays < 2^31
  for (unsigned long long i0 = 0; i0 < items; i0 += chunk) {
    const unsigned int n = (unsigned int)((items - i0 < chunk) ? (items - i0) : chunk);
    hipLaunchKernelGGL(kern, dim3(n), dim3(BLOCK), L.lds_bytes, L.stream, L.mv, L.n_records,
                       L.frame_off, L.has_sd, (unsigned int)i0, L.n_frames, L.k, L.flags, L.frame_centres,
                       L.slice_ws, L.tickets);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Experiment variants of the ADD32 kernel (MTGPU_VARIANT): bit0 UNROLL 8, bit1 dwordx4 loads,
// bit2 no nt hint, bit3 software-pipelined loop.
template <int BLOCK>
static hipError_t launch_variant(const ScanLaunch &L) {
  switch (L.variant & 15) {
#define MT_VARIANT_CASE(v) case v: return launch_one<BLOCK, 32, MODE_ADD32, ((v) & 1) ? 8 : 4, (v)>(L);
    MT_VARIANT_CASE(1) MT_VARIANT_CASE(2) MT_VARIANT_CASE(3) MT_VARIANT_CASE(4) MT_VARIANT_CASE(5)
    MT_VARIANT_CASE(6) MT_VARIANT_CASE(7) MT_VARIANT_CASE(8) MT_VARIANT_CASE(9) MT_VARIANT_CASE(10)
    MT_VARIANT_CASE(11) MT_VARIANT_CASE(12) MT_VARIANT_CASE(13) MT_VARIANT_CASE(14) MT_VARIANT_CASE(15)
#undef MT_VARIANT_CASE
    default: return launch_one<BLOCK, 32, MODE_ADD32>(L);
  }
}

// Row bands re-read a frame's records (once per band): their loads use the default cache policy
// so that the XCD's L2 keeps the lines for the sibling band (NT = false); everything else
// streams with the nt hint (-12 % without it on single-pass reads).
template <int BLOCK, bool NT>
static hipError_t launch_policy(const ScanLaunch &L) {
  constexpr int V = NT ? 0 : 4;
  const int key = L.k.mode * 100 + L.k.fb;
  switch (key) {
    case MODE_ADD32 * 100 + 32: return launch_one<BLOCK, 32, MODE_ADD32, 4, V>(L);
    case MODE_UNARY * 100 + 1: return launch_one<BLOCK, 1, MODE_UNARY, 4, V>(L);
    case MODE_UNARY * 100 + 2: return launch_one<BLOCK, 2, MODE_UNARY, 4, V>(L);
    case MODE_UNARY * 100 + 4: return launch_one<BLOCK, 4, MODE_UNARY, 4, V>(L);
    case MODE_UNARY * 100 + 8: return launch_one<BLOCK, 8, MODE_UNARY, 4, V>(L);
    case MODE_CAS * 100 + 8: return launch_one<BLOCK, 8, MODE_CAS, 4, V>(L);
    default: return hipErrorInvalidValue;
  }
}

template <int BLOCK>
static hipError_t launch_block(const ScanLaunch &L) {
  const int key = L.k.mode * 100 + L.k.fb;
  if (key == MODE_ADD32 * 100 + 32 && (L.variant & 15) != 0 && BLOCK != 1024) return launch_variant<BLOCK>(L);
  if (L.k.bands > 1 && (L.variant & 16) == 0) return launch_policy<BLOCK, false>(L);
  return launch_policy<BLOCK, true>(L);
}

hipError_t launch_scan(const ScanLaunch &L) {
  if (L.n_frames == 0) return hipSuccess;
  hipError_t e;
  if (L.k.bands > 1) {
    e = hipMemsetAsync(L.frame_centres, 0, sizeof(unsigned int) * (size_t)L.n_frames, L.stream);
    if (e != hipSuccess) return e;
  }
  if (L.k.slices > 1) {
    if (L.k.bands != 1 || !L.slice_ws || !L.tickets) return hipErrorInvalidValue;
    e = hipMemsetAsync(L.tickets, 0, sizeof(unsigned int) * (size_t)L.n_frames, L.stream);
    if (e != hipSuccess) return e;
  }
  switch (L.block) {
    case 256: e = launch_block<256>(L); break;
    case 512: e = launch_block<512>(L); break;
    case 1024: e = launch_block<1024>(L); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  if (L.k.bands > 1) {
    const unsigned int nb = (L.n_frames + 255u) / 256u;
    hipLaunchKernelGGL(finalize_flags_kernel, dim3(nb), dim3(256), 0, L.stream, L.frame_off,
                       L.n_records, L.has_sd, L.frame_centres, L.n_frames, L.k.clust_need, L.flags);
    e = hipGetLastError();
  }
  return e;
}

}  // namespace mtgpu
