// scan_kernels.hip — gfx950 kernels for MotionScanner::check_frame
// (reference: src/motion_scanner.cpp:217-295).  Integer threshold / scatter /
// stencil work: HBM-read bound, no MFMA.
//
// Work item = frame (or frame slice).  One workgroup per item:
//   phase 0  zero the vote counters in LDS                         (:229 memset)
//   phase 1  stream the frame's packed records, one record per lane and load
//            (REC 40: bytes 4..15 of each AVMotionVector = w,h,src_x,src_y,dst_x,dst_y;
//             REC 8: the compact src_x,src_y,dst_x,dst_y form the host dispatcher stages),
//            threshold on |d|^2, map dst to a cell, vote in LDS     (:242-268)
//   phase 2  per chunk of rows:
//     2a     64-bit row masks of active cells (`count >= vectors_needed`, :282) in LDS: four
//            lanes per mask word on 32-bit counters, one lane per word (bit-squeeze of the
//            fields) on packed counters
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
//            FB = 1 is a plain fire-and-forget `ds_or_b32`.
//   CAS8     FB = 8 binary field saturating at vectors_needed (9..255) via compare-and-swap.
//            Only for unusually large VECTORS_NEEDED on grids too big for 32-bit counters.
//   Packed forms need 32x..4x less LDS, so big grids (960x540) stay in one LDS tile and
//   every record is read from HBM once.
// The early `return true` (:288-289) does not change the value:
// result = (#centre cells >= max(1, clusters_needed)).
//
// Grids whose counters do not fit one LDS tile even packed are cut into ROW BANDS that the
// SAME workgroup handles one after the other (template SPILL): the records are streamed from
// HBM exactly once, during band 0; every surviving vote that a later band needs is appended to
// a per-frame queue in global memory (wave-aggregated append, the tails live in LDS) and bands 1.. replay that queue
// instead of re-reading 40-byte records.  Queue entries (struct SpillQ): 4 bytes for a vote or a run of up to 4 same-cell
// votes, (run - 1) << 30 | gy << 15 | gx — or 8 bytes for a SPAN, three or more queued votes of one wave instruction in
// consecutive cells of a row, which is what dense motion in raster-ordered records produces.  Typical
// footage queues almost nothing (only votes above the threshold); when every record votes, spans keep the queue's
// write traffic at 0.3-0.6 % of the records read (round 4, one entry per run: 2.4-5 %, which cost 8-11 % of the rate).
#if !defined(__HIP_DEVICE_COMPILE__) || defined(__gfx950__)
#else
#error "scan_kernels.hip is written for gfx950 only (sc1 write-through stores, aux bits, 160 KB LDS)"
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>

#include "knobs.h"
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

// Compact record (REC 8): src_x | src_y << 16, dst_x | dst_y << 16 — bytes 6..13 of an
// AVMotionVector, packed by the host dispatcher (pipe.hip) or mtgpu_pack_records.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef u32x2 u32x2_a8 __attribute__((aligned(8)));

template <int VAR>
__device__ __forceinline__ u32x2 load_compact(const unsigned char *rec) {
  if constexpr ((VAR & 4) != 0) return *reinterpret_cast<const u32x2_a8 *>(rec);
  else return __builtin_nontemporal_load(reinterpret_cast<const u32x2_a8 *>(rec));
}

typedef u32x4 u32x4_a16 __attribute__((aligned(16)));

template <int VAR>
__device__ __forceinline__ u32x4 load_pair(const unsigned char *two_records) {   // 16-byte aligned
  if constexpr ((VAR & 4) != 0) return *reinterpret_cast<const u32x4_a16 *>(two_records);
  else return __builtin_nontemporal_load(reinterpret_cast<const u32x4_a16 *>(two_records));
}

template <int REC> struct RawOf { typedef u32x3 type; };
template <> struct RawOf<8> { typedef u32x2 type; };

template <int VAR, int REC>
__device__ __forceinline__ typename RawOf<REC>::type load_rec(const unsigned char *rec) {
  if constexpr (REC == 8) return load_compact<VAR>(rec);
  else return load_fields<VAR>(rec);
}

struct MvFields { int src_x, src_y, dst_x, dst_y; };

__device__ __forceinline__ MvFields decode(const u32x3 d) {
  return {(int)d.x >> 16, (int)(short)(d.y & 0xffffu), (int)d.y >> 16, (int)(short)(d.z & 0xffffu)};
}
__device__ __forceinline__ MvFields decode(const u32x2 d) {
  return {(int)(short)(d.x & 0xffffu), (int)d.x >> 16, (int)(short)(d.y & 0xffffu), (int)d.y >> 16};
}

enum { MODE_ADD32 = 0, MODE_UNARY = 1, MODE_CAS = 2 };

// Developer build only (make EXTRA=-DMTGPU_PHASE_TIMES; scripts/phase_times.py): per-workgroup
// timestamps (100 MHz wall clock) — [0] start, [1] end, then time spent in [2] zeroing, [3] streaming
// records, [4] replaying the spill queue, [5] slice hand-off, [6] cluster test.  Compiled out otherwise.
#ifdef MTGPU_PHASE_TIMES
__device__ unsigned long long *g_phase_times = nullptr;
#define PT_DECL unsigned long long pt_last = wall_clock64(), pt_acc[5] = {0, 0, 0, 0, 0}; \
  if (g_phase_times && threadIdx.x == 0) g_phase_times[(size_t)item * 8] = pt_last
#define PT_ADD(slot) do { const unsigned long long pt_now = wall_clock64(); pt_acc[slot] += pt_now - pt_last; pt_last = pt_now; } while (0)
#define PT_FLUSH() do { if (g_phase_times && threadIdx.x == 0) { g_phase_times[(size_t)item * 8 + 1] = wall_clock64(); \
  for (int q = 0; q < 5; ++q) g_phase_times[(size_t)item * 8 + 2 + q] = pt_acc[q]; } } while (0)
#else
#define PT_DECL do { } while (0)
#define PT_ADD(slot) do { } while (0)
#define PT_FLUSH() do { } while (0)
#endif

template <int FB, int MODE>
__device__ __forceinline__ void bump(unsigned int *cnt, unsigned int cell, unsigned int cap) {
  if constexpr (MODE == MODE_ADD32) {
    atomicAdd(&cnt[cell], 1u);
  } else {
    constexpr unsigned int FM = (FB >= 32) ? 0xffffffffu : ((1u << FB) - 1u);
    const unsigned int bit = cell * FB;
    unsigned int *w = &cnt[bit >> 5];
    const unsigned int sh = bit & 31u;
    if constexpr (MODE == MODE_UNARY && FB == 1) {
      // one bit per cell (vectors_needed <= 1): "voted at least once" — a fire-and-forget ds_or,
      // nothing to read back, nothing to wait for
      if (cap != 0u) (void)atomicOr(w, 1u << sh);
      return;
    }
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

// n votes for one cell at once (a run of records of one wave instruction that landed in the same cell, or a
// queue entry that stands for such a run).  Same final counter state as n single bumps: counters saturate at
// `cap` and only `>= vectors_needed` is ever observed.
template <int FB, int MODE>
__device__ __forceinline__ void bump_n(unsigned int *cnt, unsigned int cell, unsigned int n, unsigned int cap) {
  if constexpr (MODE == MODE_ADD32) {
    atomicAdd(&cnt[cell], n);
  } else if constexpr (MODE == MODE_UNARY && FB == 1) {
    bump<FB, MODE>(cnt, cell, cap);
  } else {
    constexpr unsigned int FM = (FB >= 32) ? 0xffffffffu : ((1u << FB) - 1u);
    const unsigned int bit = cell * FB;
    unsigned int *w = &cnt[bit >> 5];
    const unsigned int sh = bit & 31u;
    if constexpr (MODE == MODE_UNARY) {
      // thermometer field: set the next min(left, cap - j) clear bits with ONE returning OR; bits that another
      // lane set in the meantime do not count for this one, which then continues above them.  Every updater
      // only ever sets bits from the current fill level upwards, so the code stays contiguous.
      // (measured and not kept, profiles/r04_ab_pan2.log: a fire-and-forget OR of all `cap` bits when n >= cap, and
      //  starting at bit 0 without looking first — both within +-0.5 % of this form on every input)
      unsigned int left = n < cap ? n : cap;
      unsigned int j = (unsigned int)__popc((*w >> sh) & FM);
      while (left != 0u && j < cap) {
        const unsigned int take = min(left, cap - j);
        const unsigned int m = ((1u << take) - 1u) << j;
        const unsigned int fo = (atomicOr(w, m << sh) >> sh) & FM;
        left -= (unsigned int)__popc(m & ~fo);
        j = (unsigned int)__popc(fo | m);
      }
    } else {
      unsigned int old = *w;
      for (;;) {
        const unsigned int cur = (old >> sh) & FM;
        if (cur >= cap) break;
        const unsigned int add = min(n, cap - cur);
        const unsigned int seen = atomicCAS(w, old, old + (add << sh));
        if (seen == old) break;
        old = seen;
      }
    }
  }
}

// Packed counter word -> one bit per field (32 / FB bits): is the cell active?
template <int FB, int MODE>
__device__ __forceinline__ unsigned int active_bits(unsigned int x, unsigned int vn) {
  if constexpr (MODE == MODE_UNARY) {
    // thermometer code: the field holds min(votes, vn) ones from bit 0 up, so the cell is active
    // (>= vn votes) iff bit vn-1 of its field is set; vn == 0: every cell is active
    x = vn ? (x >> (vn - 1u)) : 0xffffffffu;
    if constexpr (FB == 2) {
      x &= 0x55555555u; x = (x | (x >> 1)) & 0x33333333u; x = (x | (x >> 2)) & 0x0f0f0f0fu;
      x = (x | (x >> 4)) & 0x00ff00ffu; x = (x | (x >> 8)) & 0x0000ffffu;
    } else if constexpr (FB == 4) {
      x &= 0x11111111u; x = (x | (x >> 3)) & 0x03030303u; x = (x | (x >> 6)) & 0x000f000fu;
      x = (x | (x >> 12)) & 0x000000ffu;
    } else if constexpr (FB == 8) {
      x &= 0x01010101u; x = (x | (x >> 7)) & 0x00030003u; x = (x | (x >> 14)) & 0x0000000fu;
    }
    return x;
  } else {                                   // binary fields (CAS8): count >= vn
    constexpr unsigned int FM = (1u << FB) - 1u;
    unsigned int r = 0u;
#pragma unroll
    for (int i = 0; i < 32 / FB; ++i) r |= ((((x >> (i * FB)) & FM) >= vn) ? 1u : 0u) << i;
    return r;
  }
}

// 64-bit activity mask of cells [firstcell, firstcell + ncell) of the tile (ncell <= 64).
template <int FB, int MODE>
__device__ __forceinline__ unsigned long long mask_word(const unsigned int *cnt, unsigned int firstcell, int ncell,
                                                        unsigned int vn) {
  constexpr int CPW = 32 / FB;               // cells per counter word
  unsigned long long m = 0ull;
  unsigned int wi = (firstcell * (unsigned int)FB) >> 5;
  int rel = (int)(wi * (unsigned int)CPW) - (int)firstcell;       // <= 0: first cell of word wi, relative
  while (rel < ncell) {
    const unsigned long long bits = active_bits<FB, MODE>(cnt[wi], vn);
    m |= (rel >= 0) ? (bits << rel) : (bits >> (-rel));
    rel += CPW;
    ++wi;
  }
  if (ncell < 64) m &= (1ull << ncell) - 1ull;
  return m;
}

// Field-wise saturating sum of two packed counter words (slice combine).
template <int FB, int MODE>
__device__ __forceinline__ unsigned int combine_words(unsigned int a, unsigned int b, unsigned int cap) {
  if constexpr (MODE == MODE_ADD32) {
    return a + b;
  } else {
    if constexpr (MODE == MODE_UNARY && FB == 1) return a | b;   // one bit per cell: the sum saturates at 1
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

// The one result byte of a frame.  `sys` (ScanK::sys_flags, set by the host when `flags` is not device memory — the
// pipe's zero-copy staging, a caller's hipHostMalloc'ed buffer): a SYSTEM-scope store (global_store_byte ... sc0 sc1:
// write-through past the XCD's L2, byte-masked) — the kernel then writes pinned host memory over PCIe next to bytes
// that other workgroups, on other XCDs, write into the same line at other times, so no cache on the way may hold the
// line and merge it back later.  Device memory takes the plain store: a write-through store is only acknowledged from
// the memory side, and a workgroup cannot retire before that — measured against round 4's library in one process,
// system-scope stores for EVERY frame cost 1.2 % on 1080p (16 384 flags per launch) and 4 % on 480p (262 144).
__device__ __forceinline__ void store_flag(unsigned char *flags, unsigned int f, unsigned char v, int sys) {
  if (sys) __hip_atomic_store(&flags[f], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  else flags[f] = v;
}

// Where a spilling workgroup (SPILL) keeps the votes later bands need: the frame's queue, one dword per record of the
// frame, filled from both ends —
//   singles  from the BACK, one dword per vote (or run of same-cell votes): (run - 1) << 30 | gy << 15 | gx
//   spans    from the FRONT, two dwords for three or more queued votes of one wave instruction that fall into
//            CONSECUTIVE cells of one grid row with equal run lengths — what raster-ordered records produce wherever
//            motion is dense:  {(run - 1) << 30 | gy << 15 | gx of the first cell, number of cells}.  A wave
//            instruction is cut into such segments wherever the cells stop being consecutive (end of a grid row, a
//            block that did not move) or the run length changes (a block's records cut by the edge of the
//            instruction); segments of one or two votes go to the singles.
// Round 5 measured what vote-heavy input costs on banded plans (profiles/r05_ab_spill_3_ablations.log): not the votes
// (LDS votes removed: no change), not the replay (3 %), but the queue's STORES — 1-5 % of extra write traffic into a
// saturated read stream cost 8-11 % of its rate, whatever their cache policy, order or instruction form.  A span entry
// is 8 bytes where the same votes took up to 256: the queue of a frame in which every record votes shrinks about
// 30-fold (one record per cell) or 4- to 8-fold (four per cell).  A span stands for >= 3 votes in 2 dwords, a single
// for one vote in one: the two ends never meet.
struct SpillQ {
  unsigned int *q;        // this frame's queue: n dwords
  unsigned int n;         // its size in dwords (records of the frame, at most 2^32 - 1)
  unsigned int *tail;     // LDS: tail[0] singles so far (dwords from the back), tail[1] span dwords so far (from the front)
  int q_lo;               // first grid row a later band tracks (band 0's last centre row)
};

// `r` votes for each of the cells [c0, c0 + n) of the tile (consecutive cells of one row: a replayed span).  Thermometer
// fields: a counter WORD at a time — one look, the next clear bits of every field of the word in ONE returning OR
// (8 cells of a 4-bit form), and only the votes that lost a race to another wave settle cell by cell; other forms
// cell by cell.  `rot`: the word the walk starts at (it wraps around) — the lanes of a wave replay consecutive spans,
// whose words lie a fixed stride apart: started at the same relative word they would all hit a handful of LDS banks.
template <int FB, int MODE>
__device__ __forceinline__ void bump_cells(unsigned int *cnt, unsigned int c0, unsigned int n, unsigned int r, unsigned int cap,
                                           unsigned int rot) {
  if constexpr (MODE == MODE_UNARY && FB >= 2 && FB <= 8) {
    constexpr unsigned int CPW = 32u / FB, FM = (1u << FB) - 1u;
    if (n == 0u) return;
    const unsigned int rr = r < cap ? r : cap;
    constexpr unsigned int EVERY = 0xffffffffu / FM;                     // bit 0 of every field: 0x11111111 for 4-bit fields
    const unsigned int low_rr = EVERY * ((1u << rr) - 1u), cap_ones = EVERY * ((1u << cap) - 1u);
    const unsigned int end = c0 + n, first_w = c0 / CPW, last_w = (end - 1u) / CPW, nw = last_w - first_w + 1u;
    unsigned int at = rot % nw;
    for (unsigned int i = 0; i < nw; ++i) {
      const unsigned int wi = first_w + at;
      at = at + 1u == nw ? 0u : at + 1u;
      const unsigned int lo = wi == first_w ? c0 - first_w * CPW : 0u, hi = wi == last_w ? end - last_w * CPW : CPW;
      const unsigned int x = cnt[wi];
      // every field of the word at once: a thermometer code grows by rr when it is shifted up by rr and its low rr
      // bits are set — what the shift pushes out of a field's top lands in the low rr bits of the next field, which
      // are set anyway — and stays within `cap` ones under the cap pattern; `sel` keeps the fields [lo, hi)
      const unsigned int sel = (hi == CPW ? 0xffffffffu : (1u << (hi * FB)) - 1u) & ~((1u << (lo * FB)) - 1u);
      const unsigned int m = (((x << rr) | low_rr) & cap_ones & sel) & ~x;
      if (m != 0u) {
        const unsigned int lost = m & atomicOr(&cnt[wi], m);       // bits somebody else set between look and OR
        if (lost != 0u) {
#pragma unroll
          for (unsigned int q = 0; q < CPW; ++q) {
            const unsigned int l = (unsigned int)__popc((lost >> (q * FB)) & FM);
            if (l != 0u) bump_n<FB, MODE>(cnt, wi * CPW + q, l, cap);
          }
        }
      }
    }
  } else {
    for (unsigned int c = 0; c < n; ++c) bump_n<FB, MODE>(cnt, c0 + c, r, cap);
  }
}

// `r` votes for each of the cells [c0, c0 + n) of the tile (n <= 64, consecutive cells of one row), by a whole wave:
// lane i takes the i-th counter word the cells touch — the word-at-once step of bump_cells, every word in parallel.
template <int FB, int MODE>
__device__ __forceinline__ void bump_cells_wave(unsigned int *cnt, unsigned int c0, unsigned int n, unsigned int r, unsigned int cap,
                                                unsigned int lane) {
  constexpr unsigned int CPW = 32u / FB, FM = (1u << FB) - 1u;
  const unsigned int rr = r < cap ? r : cap;
  constexpr unsigned int EVERY = 0xffffffffu / FM;                       // bit 0 of every field
  const unsigned int low_rr = EVERY * ((1u << rr) - 1u), cap_ones = EVERY * ((1u << cap) - 1u);
  const unsigned int end = c0 + n, first_w = c0 / CPW, last_w = (end - 1u) / CPW;
  const unsigned int wi = first_w + lane;
  if (wi > last_w) return;
  const unsigned int lo = wi == first_w ? c0 - first_w * CPW : 0u, hi = wi == last_w ? end - last_w * CPW : CPW;
  const unsigned int x = cnt[wi];
  const unsigned int sel = (hi == CPW ? 0xffffffffu : (1u << (hi * FB)) - 1u) & ~((1u << (lo * FB)) - 1u);
  const unsigned int m = (((x << rr) | low_rr) & cap_ones & sel) & ~x;   // (see bump_cells)
  if (m != 0u) {
    const unsigned int lost = m & atomicOr(&cnt[wi], m);                 // bits somebody else set between look and OR
    if (lost != 0u) {
#pragma unroll
      for (unsigned int q = 0; q < CPW; ++q) {
        const unsigned int l = (unsigned int)__popc((lost >> (q * FB)) & FM);
        if (l != 0u) bump_n<FB, MODE>(cnt, wi * CPW + q, l, cap);
      }
    }
  }
}

// Threshold + cell mapping + vote for one record (src/motion_scanner.cpp:246-267).
// [t0,t1) = grid rows this tile tracks.
// DENSE: try the dense-wave shortcut first — only the main streaming loop over 40-byte records asks for it (whole
// waves, one record per lane in stream order; the compact loop holds two records per lane, head and tail calls run
// under divergence): compiled into every call site it tripled the library's code size.
template <int FB, int MODE, bool SPILL, bool DENSE = false>
__device__ __forceinline__ void vote(const MvFields m, const ScanK &k, int t0, int t1,
                                     unsigned int *cnt, const SpillQ &sq) {
  const unsigned int dx = (unsigned int)(m.dst_x - m.src_x);   // |dx| <= 65535
  const unsigned int dy = (unsigned int)(m.dst_y - m.src_y);
  // dx*dx < 2^32 exactly; the sum needs 34 bits.  (The compiler proves the operands fit 17 bits and already
  // emits the full-rate v_mul_i32_i24 for the squares; __mul24() must NOT be used here: its library definition
  // multiplies signed ints, so the optimiser may assume dx*dx < 2^31 and drops the 33rd bit of the sum.)
  const unsigned long long mag =
      (unsigned long long)(dx * dx) + (unsigned long long)(dy * dy);
  const int gx = m.dst_x >> k.shift;
  const int gy = m.dst_y >> k.shift;
  // 0 <= gx < gw and y_lo <= gy < y_hi (:262) as two unsigned compares (y_hi >= y_lo by construction)
  const bool in = (mag >= k.thr) & ((unsigned int)gx < (unsigned int)k.gw) &
                  ((unsigned int)(gy - k.y_lo) < (unsigned int)(k.y_hi - k.y_lo));
  if constexpr (!SPILL && (MODE == MODE_ADD32 || (MODE == MODE_UNARY && FB == 1))) {
    // fire-and-forget LDS atomics (32-bit add, 1-bit or): a single tile tracks every analysed row
    if (in) bump<FB, MODE>(cnt, (unsigned int)((gy - t0) * k.gw + gx), k.vec_need);
  } else {
    // Returning LDS atomics (thermometer / CAS fields) and the spill queue: RUNS of records that one wave
    // instruction maps to the same cell — codecs export several vectors per block (two prediction directions,
    // partitions), in stream order — vote once, with their count: on a frame where every record votes, the 4
    // records of a block otherwise queue up on ONE LDS word four times over (8 cells of a 4-bit form share a
    // word), and each would append its own queue entry.  A wave instruction without a voter costs one ballot.
    const unsigned long long any = __ballot(in);
    if (any == 0ull) return;
    const int lane = (int)(threadIdx.x & 63u);
    const unsigned int key = in ? (((unsigned int)gy << 15) | (unsigned int)gx) : 0xffffffffu;   // gx, gy < 32768
    if constexpr (DENSE && MODE == MODE_UNARY && FB >= 2 && FB <= 8) {
      // DENSE wave instruction — all 64 records vote and fall, in order, into consecutive cells of one grid row, R = 1, 2
      // or 4 records per cell: what dense motion (a camera pan) looks like in raster-ordered records.  The general path
      // below (runs, queue segments: a shuffle, eight ballots and 64-bit bit-scans per wave instruction) makes such input
      // instruction-bound; here the pattern is read off a few lanes (first run: lanes 0..3, run length: the second
      // cell's lanes), verified with ONE ballot, and the wave's votes are: the first cell (R - ph votes: the instruction
      // may start inside a cell — where a frame's records start, and how many head records the line alignment peeled,
      // is data), the cells in between (R each: their counter words voted in parallel, one lane per word), the last
      // cell (the rest).  The queue gets ONE span for the cells in between and a single each for a cut first / last
      // cell — what the general path would have written.  Same counter state and same replayed votes (saturating
      // counts commute).  Needs the whole wave (the calls for head records and tails do not have it).
      if (any == ~0ull && __ballot(true) == ~0ull) {
        const unsigned int key0 = (unsigned int)__builtin_amdgcn_readfirstlane((int)key);
        const unsigned int k1 = (unsigned int)__builtin_amdgcn_readlane((int)key, 1);
        const unsigned int k2 = (unsigned int)__builtin_amdgcn_readlane((int)key, 2);
        const unsigned int k3 = (unsigned int)__builtin_amdgcn_readlane((int)key, 3);
        const unsigned int L1 = k1 != key0 ? 1u : (k2 != key0 ? 2u : (k3 != key0 ? 3u : 4u));     // votes for the first cell
        const unsigned int a1 = (unsigned int)__builtin_amdgcn_readlane((int)key, (int)L1 + 1);
        const unsigned int a2 = (unsigned int)__builtin_amdgcn_readlane((int)key, (int)L1 + 2);
        const unsigned int a3 = (unsigned int)__builtin_amdgcn_readlane((int)key, (int)L1 + 3);
        const unsigned int R = a1 != key0 + 1u ? 1u : (a2 != key0 + 1u ? 2u : (a3 != key0 + 1u ? 3u : 4u));
        if (R != 3u && L1 <= R) {
          const unsigned int sh = R >> 1, ph = R - L1;                                              // log2 R; lanes of the first cell that belong to the instruction before
          if (__ballot(key != key0 + (((unsigned int)lane + ph) >> sh)) == 0ull) {
            const unsigned int n_cells = ((63u + ph) >> sh) + 1u, last = ((63u + ph) & (R - 1u)) + 1u;
            const int gy0 = (int)(key0 >> 15), gx0 = (int)(key0 & 0x7fffu);
            const bool cut = ph != 0u;                                                              // first and last cell hold fewer than R votes
            if (!SPILL || (unsigned int)(gy0 - t0) < (unsigned int)(t1 - t0)) {
              const unsigned int c0 = (unsigned int)((gy0 - t0) * k.gw + gx0);
              if (!cut) {
                bump_cells_wave<FB, MODE>(cnt, c0, n_cells, R, k.vec_need, (unsigned int)lane);
              } else {
                bump_cells_wave<FB, MODE>(cnt, c0 + 1u, n_cells - 2u, R, k.vec_need, (unsigned int)lane);
                if (lane == 63) bump_n<FB, MODE>(cnt, c0, L1, k.vec_need);
                if (lane == 62) bump_n<FB, MODE>(cnt, c0 + n_cells - 1u, last, k.vec_need);
              }
            }
            if constexpr (SPILL) {
              if (gy0 >= sq.q_lo && lane == 0) {
                const unsigned int at = atomicAdd(sq.tail + 1, 2u);                                 // (n_cells - 2 >= 14: always a span)
                sq.q[at] = ((R - 1u) << 30) | (cut ? key0 + 1u : key0);
                sq.q[at + 1u] = cut ? n_cells - 2u : n_cells;
                if (cut) {
                  const unsigned int b1 = atomicAdd(sq.tail, 2u);
                  sq.q[sq.n - 1u - b1] = ((L1 - 1u) << 30) | key0;
                  sq.q[sq.n - 2u - b1] = ((last - 1u) << 30) | (key0 + n_cells - 1u);
                }
              }
            }
            return;
          }
        }
      }
    }
    const unsigned int prev = (unsigned int)__shfl_up((int)key, 1);
    // a queue entry carries a run of at most 4 (two spare bits): where a field can count beyond 4, runs are cut
    // every 4 lanes so that no vote is lost to the entry format
    const unsigned long long forced = (k.vec_need > 4u) ? 0x1111111111111111ull : 1ull;
    // vote() is also called under divergence (head records, tails): a lane that is switched off ends the run
    // below it, and the lane above it starts one (what __shfl_up brings from an inactive lane is undefined)
    const unsigned long long off = ~__ballot(true);
    const unsigned long long heads = __ballot(key != prev) | forced | off | (off << 1);
    const bool head = ((heads >> lane) & 1ull) != 0ull;
    const unsigned long long above = (lane < 63) ? (heads >> (lane + 1)) : 0ull;
    const unsigned int run = above ? (unsigned int)__ffsll((long long)above) : (unsigned int)(64 - lane);
    bool mine = in & head;                       // a band (SPILL) has to test its own rows
    if constexpr (SPILL) mine = mine & ((unsigned int)(gy - t0) < (unsigned int)(t1 - t0));
    if (mine) bump_n<FB, MODE>(cnt, (unsigned int)((gy - t0) * k.gw + gx), run, k.vec_need);
    if constexpr (SPILL) {
      // wave-aggregated append (see SpillQ): one returning LDS add per wave instruction
      const bool qv = in & head & (gy >= sq.q_lo);
      const unsigned long long qm = __ballot(qv);
      if (qm != 0ull) {
        const int leader = __ffsll((long long)qm) - 1;
        const unsigned int rank = (unsigned int)__popcll(qm & ((1ull << lane) - 1ull));   // queued votes below this lane
        const unsigned int r4 = min(run, 4u) - 1u;
        const unsigned int e = (r4 << 30) | key;
        // segments: maximal stretches of queued votes whose entries count up by exactly one from vote to vote —
        // consecutive cells (gy << 15 | gx, gx < 32768: the next cell of the SAME row) with the same run length
        const unsigned long long below = (1ull << lane) - 1ull;
        const unsigned int d = e - rank;
        const unsigned long long qb = qm & below;
        const int prev_lane = qb ? 63 - __clzll((long long)qb) : lane;              // the queued lane below this one
        const unsigned int d_prev = (unsigned int)__shfl((int)d, prev_lane);
        const unsigned long long sh = __ballot(qv && (qb == 0ull || d != d_prev));   // first vote of every segment
        const unsigned long long at_or_below = sh & (below | (1ull << lane));
        const int seg_lo = at_or_below ? 63 - __clzll((long long)at_or_below) : 0;
        const unsigned long long heads_above = (lane < 63) ? (sh >> (lane + 1)) << (lane + 1) : 0ull;
        const int seg_hi = heads_above ? __ffsll((long long)heads_above) - 1 : 64;   // [seg_lo, seg_hi): this lane's segment
        const unsigned long long seg_mask = (seg_hi >= 64 ? ~0ull : ((1ull << seg_hi) - 1ull)) & ~((1ull << seg_lo) - 1ull);
        const unsigned int seg_len = (unsigned int)__popcll(qm & seg_mask);
        const bool span_head = qv && lane == seg_lo && seg_len >= 3u;
        const bool single = qv && seg_len < 3u;
        const unsigned long long spans = __ballot(span_head), singles = __ballot(single);
        unsigned int base_s = 0u, base_1 = 0u;
        if (lane == leader) {
          if (spans != 0ull) base_s = atomicAdd(sq.tail + 1, 2u * (unsigned int)__popcll(spans));
          if (singles != 0ull) base_1 = atomicAdd(sq.tail, (unsigned int)__popcll(singles));
        }
        base_s = (unsigned int)__builtin_amdgcn_readlane((int)base_s, leader);
        base_1 = (unsigned int)__builtin_amdgcn_readlane((int)base_1, leader);
        if (span_head) {
          const unsigned int at = base_s + 2u * (unsigned int)__popcll(spans & below);
          sq.q[at] = e;
          sq.q[at + 1u] = seg_len;
        } else if (single) {
          sq.q[sq.n - 1u - (base_1 + (unsigned int)__popcll(singles & below))] = e;
        }
      }
    }
  }
}

// Compact records: how many records at the start of a frame's array are scanned one by one so that the
// 16-byte pair stream starts on a 128-byte line (align != 0; < 16 records) or at least on a 16-byte boundary.
__device__ __forceinline__ unsigned long long compact_head(const unsigned char *base, unsigned long long n, int align) {
  const unsigned long long a = (unsigned long long)(uintptr_t)base;
  unsigned long long h = align ? (((0ull - a) & 127ull) >> 3) : ((a >> 3) & 1ull);
  return h < n ? h : n;
}

// Compact records, several frames per workgroup (k.group > 1): the first streaming step of the NEXT frame is
// issued while this frame's cluster test and the next frame's zeroing run.  Those phases touch only LDS (their
// barriers wait on lgkmcnt, not on vector-memory loads), so the 4 x 16 bytes per lane stay in flight across
// them and the CU's memory queue never runs dry between two frames of a workgroup.  `have` is workgroup-uniform.
template <int UNROLL>
struct NextStep {
  u32x4 d[UNROLL];
  unsigned int frame;   // the frame these loads belong to (workgroup-uniform, like `have`)
  bool have;
};

// Does frame f have MV side data, and which records are its own (the offsets clamped to the batch).
// has_sd == NULL: a frame has side data iff it has records.
__device__ __forceinline__ bool plan_frame(const unsigned long long *__restrict__ frame_off, const unsigned char *__restrict__ has_sd,
                                           unsigned long long n_records, unsigned long long f, unsigned long long &r0,
                                           unsigned long long &r1) {
  r0 = frame_off[f];
  r1 = frame_off[f + 1];
  r1 = r1 < n_records ? r1 : n_records;
  r0 = r0 < r1 ? r0 : r1;
  return has_sd ? (has_sd[f] != 0) : (r1 > r0);
}

// An entry of the work list with ONE 32-byte load — r0, r1 and f arrive together (read field by field the compiler
// fetches f first, tests it, and only then asks for r0 / r1: two memory round trips at the start of every workgroup's
// life instead of one; the list was just written by another kernel, so the first touch of a line comes from beyond
// this XCD's L2).  The address is workgroup-uniform: a scalar load.
__device__ __forceinline__ WorkItem load_item(const WorkItem *__restrict__ work, unsigned int wi) {
  typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
  const u32x8 raw = *reinterpret_cast<const u32x8 *>(work + wi);
  WorkItem it;
  it.r0 = (unsigned long long)raw[0] | ((unsigned long long)raw[1] << 32);
  it.r1 = (unsigned long long)raw[2] | ((unsigned long long)raw[3] << 32);
  it.f = raw[4];
  it.pad[0] = it.pad[1] = it.pad[2] = 0u;
  return it;
}

// The entries a workgroup that owns several frames needs after its first: fetched by lanes 1 .. group-1 when the
// workgroup starts (one memory latency for all of them, overlapped with the first entry's scalar load), parked in LDS
// behind the tile, and read from there frame by frame.  Tiny frames (SD streams as compact records: 10 KB, about a
// microsecond of streaming each) cannot hide a trip to memory per frame behind the previous frame — asked for one frame
// ahead with scalar loads, 480p compact records ran at 1919 us per 262 144 frames against 1447 for round 5, whose
// workgroups found their eight offsets in ONE cache line.
// (kStageBytes of LDS behind the tile: 64 entries — choose_group never exceeds 64 frames per workgroup)
__device__ __forceinline__ void stage_items(const WorkItem *__restrict__ work, unsigned int first, unsigned int n_items, int group,
                                            unsigned int *stage) {
  const unsigned int j = threadIdx.x;
  if (j >= 1u && j < (unsigned int)group && first + j < n_items) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(work + first + j);
    const u32x4 a = src[0], b = src[1];
    u32x4 *dst = reinterpret_cast<u32x4 *>(stage) + 2u * j;
    dst[0] = a;
    dst[1] = b;
  }
}
__device__ __forceinline__ WorkItem staged_item(const unsigned int *stage, unsigned int j) {
  const u32x4 *src = reinterpret_cast<const u32x4 *>(stage) + 2u * j;      // every lane reads the same address: a broadcast
  const u32x4 a = src[0], b = src[1];
  WorkItem it;
  it.r0 = (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)a.x) |
          ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)a.y) << 32);
  it.r1 = (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)a.z) |
          ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)a.w) << 32);
  it.f = (unsigned int)__builtin_amdgcn_readfirstlane((int)b.x);
  it.pad[0] = it.pad[1] = it.pad[2] = 0u;
  return it;
}

// item -> entry of the work list: bands and slices are never both > 1, and one slice per frame is the common case
template <bool SPILL>
__device__ __forceinline__ unsigned int item_entry(unsigned int item, int slices) {
  return (SPILL || slices == 1) ? item : item / (unsigned int)slices;
}

// One work item (the list entry `me`, or a slice of it) by one workgroup.  `has_next`: the same workgroup scans
// item + 1 (the list entry parked at `stage_next`: readable after this item's first barrier) right after this one.
// Frames without side data (:219-221) never get here: plan_scatter_kernel has answered them.
template <int BLOCK, int UNROLL, int FB, int MODE, int VAR, int REC, bool SPILL>
__device__ __forceinline__ void scan_item(
    const unsigned char *__restrict__ mv, const WorkItem me, const unsigned int *stage_next,
    const unsigned int item, const ScanK &k, unsigned char *__restrict__ flags,
    unsigned int *spill_q, unsigned int *slice_ws, unsigned int *tickets, unsigned int *lds,
    NextStep<UNROLL> &ns, const bool has_next) {
  typedef typename RawOf<REC>::type Raw;
  const int tid = threadIdx.x;
  PT_DECL;
  const unsigned int wi = item_entry<SPILL>(item, k.slices);
  const int slice = (SPILL || k.slices == 1) ? 0 : (int)(item - wi * (unsigned int)k.slices);
  const unsigned int f = me.f;
  // A pre-issued step is consumed only by the frame it was loaded for: whatever early-out a frame takes
  // between here and its streaming loop, a step that was not consumed can never leak its votes into a LATER frame.
  if (ns.have && ns.frame != f) ns.have = false;

  unsigned long long r0 = me.r0, r1 = me.r1;
  const unsigned long long q0 = r0;            // the frame's spill queue: one slot per record
  if (!SPILL && k.slices > 1) {                // this workgroup's share of the frame's records
    const unsigned long long n = r1 - r0, per = (n + (unsigned long long)k.slices - 1ull) / (unsigned long long)k.slices;
    const unsigned long long a = r0 + min(n, per * (unsigned long long)slice);
    const unsigned long long b = r0 + min(n, per * (unsigned long long)(slice + 1));
    r0 = a;
    r1 = b;
  }
  const int W = k.W;
  unsigned int *cnt = lds;                                         // packed [trows][gw] fields
  unsigned long long *mask =
      reinterpret_cast<unsigned long long *>(lds + k.cnt_words);   // [chunk_rows+2][W]
  unsigned int *total = reinterpret_cast<unsigned int *>(mask + (size_t)k.mask_rows * W);
  unsigned int *ticket = total + 1;
  SpillQ sq;
  sq.q = SPILL ? spill_q + q0 : nullptr;
  sq.n = (unsigned int)min(r1 - r0, 0xffffffffull);
  sq.tail = total + 2;                                             // two words: singles, span dwords
  sq.q_lo = min(k.y_hi, k.y_lo + k.band_rows) - 1;                 // band 0's last centre row

  const int n_bands = SPILL ? k.bands : 1;
  unsigned int local = 0;
  for (int band = 0; band < n_bands; ++band) {
    // Band geometry: centres [c0,c1), tracked counter rows [t0,t1) (one halo row each side).
    const int c0 = k.y_lo + band * k.band_rows;
    const int c1 = min(k.y_hi, c0 + k.band_rows);
    const int t0 = max(c0 - 1, 0);
    const int t1 = min(c1 + 1, k.gh);
    const int trows = t1 - t0;                 // may be <= 0 for an empty analysed range

    // ---- phase 0: zero counters (and, once, the centre total and the queue tail)
    {
      u32x4 *c4 = reinterpret_cast<u32x4 *>(cnt);
      const int n4 = k.cnt_words >> 2;
      for (int i = tid; i < n4; i += BLOCK) c4[i] = (u32x4){0u, 0u, 0u, 0u};
      if (band == 0 && tid == 0) { *total = 0u; sq.tail[0] = 0u; sq.tail[1] = 0u; }
    }
    __syncthreads();
    PT_ADD(0);

    // ---- phase 1
    if (band == 0) {                           // stream the records (HBM, exactly once per frame)
      if (trows > 0 && k.vec_need != 0u) {     // vec_need == 0: every cell is active anyway
        const unsigned char *base = mv + r0 * (unsigned long long)REC;
        unsigned long long n = r1 - r0;
        if constexpr (REC == 40) {
          // Line alignment.  A wave instruction of the loops below covers 64 records = 2560 bytes = exactly 20
          // 128-byte lines IF the stream starts on a line; a frame that starts mid-line (real footage: every
          // frame has its own record count) makes every instruction touch 21 lines, and the shared edge lines
          // are fetched twice, by two different instructions (PMC, ragged 960x540 frames: 1.021 x the
          // algorithmic bytes).  40 h = -start (mod 128) has a solution h < 16 whenever the start is 8-byte
          // aligned (5 * 13 = 1 mod 16): the first h records go to lanes 0..h-1, the streams start on a line.
          const unsigned int r = (unsigned int)((uintptr_t)base & 127u);
          if (k.align_lines && (r & 7u) == 0u) {
            unsigned long long h = (unsigned long long)((13u * ((16u - (r >> 3)) & 15u)) & 15u);
            h = h < n ? h : n;
            if ((unsigned long long)tid < h)
              vote<FB, MODE, SPILL>(decode(load_rec<VAR, REC>(base + (unsigned long long)tid * REC)), k, t0, t1, cnt, sq);
            base += h * (unsigned long long)REC;
            n -= h;
          }
        }
        unsigned long long i = tid;
        constexpr unsigned long long STEP = (unsigned long long)UNROLL * BLOCK;
        constexpr unsigned long long LAST = (unsigned long long)(UNROLL - 1) * BLOCK;
        if constexpr (REC == 8) {
          // Compact records: 16-byte loads of TWO records per lane (a wave instruction covers 1 KB),
          // UNROLL pairs in flight per lane — with 8-byte loads a CU keeps too few bytes in flight
          // to cover the HBM latency (measured 4.96 TB/s of compact bytes on 1080p).  The pair
          // stream starts at the first record on a 128-byte line (compact_head: up to 15 head records go to
          // lanes 0..14, so that a wave instruction covers exactly 8 lines); lane 0 takes an odd last record.
          const unsigned long long head = compact_head(base, n, k.align_lines);
          const unsigned char *pbase = base + head * 8ull;
          const unsigned long long np = (n - head) >> 1;            // pairs
          if ((unsigned long long)tid < head)
            vote<FB, MODE, SPILL>(decode(load_compact<VAR>(base + (unsigned long long)tid * 8ull)), k, t0, t1, cnt, sq);
          if (tid == 0 && ((n - head) & 1ull) != 0ull)
            vote<FB, MODE, SPILL>(decode(load_compact<VAR>(base + (n - 1ull) * 8ull)), k, t0, t1, cnt, sq);
          unsigned long long p = tid;
          if (ns.have) {                       // this frame's first step (ns.frame == f, checked on entry) was issued during
            ns.have = false;                   // the previous frame's cluster test
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
              vote<FB, MODE, SPILL>(decode((u32x2){ns.d[u].x, ns.d[u].y}), k, t0, t1, cnt, sq);
              vote<FB, MODE, SPILL>(decode((u32x2){ns.d[u].z, ns.d[u].w}), k, t0, t1, cnt, sq);
            }
            p += STEP;
          }
          for (; p + LAST < np; p += STEP) {
            u32x4 d[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) d[u] = load_pair<VAR>(pbase + (p + (unsigned long long)u * BLOCK) * 16ull);
            __builtin_amdgcn_sched_barrier(0);   // every load of the step is issued before the first one is consumed
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
              vote<FB, MODE, SPILL>(decode((u32x2){d[u].x, d[u].y}), k, t0, t1, cnt, sq);
              vote<FB, MODE, SPILL>(decode((u32x2){d[u].z, d[u].w}), k, t0, t1, cnt, sq);
            }
          }
          {
            // the rest (fewer than one step: at most UNROLL pairs per lane), every load issued before the first
            // vote — a frame smaller than one step (SD streams) costs ONE memory round trip, not one per pair
            u32x4 d[UNROLL];
            bool ok[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
              const unsigned long long q = p + (unsigned long long)u * BLOCK;
              ok[u] = q < np;
              d[u] = ok[u] ? load_pair<VAR>(pbase + q * 16ull) : (u32x4){0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
              if (ok[u]) {
                vote<FB, MODE, SPILL>(decode((u32x2){d[u].x, d[u].y}), k, t0, t1, cnt, sq);
                vote<FB, MODE, SPILL>(decode((u32x2){d[u].z, d[u].w}), k, t0, t1, cnt, sq);
              }
          }
          i = n;                                                    // nothing left for the generic tail loop
        } else if constexpr ((VAR & 8) != 0) {
          // software-pipelined: the next batch of loads is issued before this batch is consumed
          Raw cur[UNROLL], nxt[UNROLL];
          bool have = i + LAST < n;
          if (have) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) cur[u] = load_rec<VAR, REC>(base + (i + (unsigned long long)u * BLOCK) * REC);
          }
          while (have) {
            const unsigned long long j = i + STEP;
            const bool more = j + LAST < n;
            if (more) {
#pragma unroll
              for (int u = 0; u < UNROLL; ++u) nxt[u] = load_rec<VAR, REC>(base + (j + (unsigned long long)u * BLOCK) * REC);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) vote<FB, MODE, SPILL>(decode(cur[u]), k, t0, t1, cnt, sq);
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
            Raw d[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) d[u] = load_rec<VAR, REC>(base + (i + (unsigned long long)u * BLOCK) * REC);
            // (the scheduler sinks loads 2..UNROLL below the wait for load 1; forcing them up front with
            //  a sched_barrier measured -1..-2 % here, +7 % in the compact loop above: left as it is)
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) vote<FB, MODE, SPILL, true>(decode(d[u]), k, t0, t1, cnt, sq);
          }
        }
        if (i < n) {
          // the rest (fewer than one step: at most UNROLL records per lane), every load issued before the first
          // vote — a frame smaller than one step (SD streams) costs ONE memory round trip, not one per record
          constexpr int TU = UNROLL > 4 ? UNROLL : 4;
          Raw d[TU];
          bool ok[TU];
#pragma unroll
          for (int u = 0; u < TU; ++u) {
            const unsigned long long q = i + (unsigned long long)u * BLOCK;
            ok[u] = q < n;
            if (ok[u]) d[u] = load_rec<VAR, REC>(base + q * REC);
          }
#pragma unroll
          for (int u = 0; u < TU; ++u)
            if (ok[u]) vote<FB, MODE, SPILL>(decode(d[u]), k, t0, t1, cnt, sq);
        }
      }
      if constexpr (REC == 8 && !SPILL) {
        if (has_next && k.slices == 1 && k.vec_need != 0u && trows > 0) {   // exactly when the next frame's phase 1 runs
          // next frame of the list (parked in LDS when the workgroup started): same pair alignment as above
          const WorkItem nx = staged_item(stage_next, 0u);
          const unsigned long long a = nx.r0, b = nx.r1;
          const bool sdn = nx.f != kNoFrame;
          const unsigned char *nb = mv + a * 8ull;
          const unsigned long long nn = b - a;
          const unsigned long long nhead = compact_head(nb, nn, k.align_lines);
          const unsigned long long nnp = (nn - nhead) >> 1;
          if (sdn && nnp >= (unsigned long long)UNROLL * BLOCK) {     // the whole first step lies inside the frame: uniform
            const unsigned char *npb = nb + nhead * 8ull;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
              ns.d[u] = load_pair<VAR>(npb + ((unsigned long long)tid + (unsigned long long)u * BLOCK) * 16ull);
            ns.have = true;
            ns.frame = nx.f;
          }
        }
      }
      // queue stores of every wave have left the CU before any wave of this workgroup replays them
      if constexpr (SPILL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {                                   // replay the votes band 0 queued for later bands
      // singles (from the back of the queue): one entry per lane, four loads in flight
      const unsigned int nq = sq.tail[0];
      const unsigned int *qs = sq.q + (sq.n - nq);       // entries nq-1 .. 0 in ascending address order
      unsigned int i = (unsigned int)tid;
      for (; i + 3u * BLOCK < nq; i += 4u * BLOCK) {
        unsigned int e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = qs[i + (unsigned int)u * BLOCK];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int gy = (int)((e[u] >> 15) & 0x7fffu), gx = (int)(e[u] & 0x7fffu);
          if (gy >= t0 && gy < t1) bump_n<FB, MODE>(cnt, (unsigned int)((gy - t0) * k.gw + gx), (e[u] >> 30) + 1u, k.vec_need);
        }
      }
      for (; i < nq; i += BLOCK) {
        const unsigned int e = qs[i];
        const int gy = (int)((e >> 15) & 0x7fffu), gx = (int)(e & 0x7fffu);
        if (gy >= t0 && gy < t1) bump_n<FB, MODE>(cnt, (unsigned int)((gy - t0) * k.gw + gx), (e >> 30) + 1u, k.vec_need);
      }
      // spans (from the front): one span per LANE, which walks the span's cells.  Neighbouring lanes then vote in
      // different rows / far-apart columns — different LDS words — where the cells of ONE span share words (8 cells of
      // a 4-bit form to a word): no same-word serialisation of the returning ORs, and 64 spans in flight per wave.
      const unsigned int nspans = sq.tail[1] >> 1;
      unsigned int sp = (unsigned int)tid;
      bool have = sp < nspans;
      unsigned int h0 = have ? sq.q[2u * sp] : 0u, h1 = have ? sq.q[2u * sp + 1u] : 0u;
      while (have) {
        const unsigned int nx = sp + BLOCK;                // the next header is on its way while this span votes
        const bool more = nx < nspans;
        const unsigned int n0 = more ? sq.q[2u * nx] : 0u, n1 = more ? sq.q[2u * nx + 1u] : 0u;
        const int gy = (int)((h0 >> 15) & 0x7fffu), gx = (int)(h0 & 0x7fffu);
        if (gy >= t0 && gy < t1) {
          bump_cells<FB, MODE>(cnt, (unsigned int)((gy - t0) * k.gw + gx), h1, (h0 >> 30) + 1u, k.vec_need,
                               ((unsigned int)tid >> 2) & 7u);
        }
        h0 = n0; h1 = n1; sp = nx; have = more;
      }
    }
    __syncthreads();
    PT_ADD(band == 0 ? 1 : 2);

    // ---- slices: publish this partial grid; the LAST workgroup of the frame to arrive sums them
    // (no workgroup ever waits for another: nothing to deadlock on).  This is the guide's
    // Guideline 16 hand-off, recipe R1 in its counter form (MI355X_MICROARCH.md "visibility",
    // valid-forms row "ONE lane of each storing workgroup ... an agent-scope atomic add ... the
    // workgroup whose add came last, told by the value its add returned"):
    //   producer  every payload byte leaves with a 16-byte WRITE-THROUGH (sc1) store: it goes
    //             through the XCD's L2 to memory and drops the L2 line, so no release fence
    //             (buffer_wbl2) is needed; EVERY storing wave drains (s_waitcnt vmcnt(0)); the
    //             workgroup barrier orders all drains before ONE lane's agent-scope ticket add
    //             (an atomic executes at the memory side, beyond the per-XCD L2s).
    //   consumer  the last arriver learns it from the add's return value; ONE lane executes the
    //             agent-scope acquire (buffer_inv sc1: drops this CU's L1 lines — the L1 is
    //             shared by the CU's waves, the invalidate is not per wave) and waits for it
    //             (vmcnt(0)); the barrier keeps every other wave's loads behind that wait; then
    //             plain loads.  The workspace is ordinary coarse-grained device memory
    //             (the context's scratch ring) that earlier launches may have cached on this CU: the
    //             acquire is what makes those stale L1 lines unreachable.  A line of another
    //             workgroup's tile cannot sit stale in THIS XCD's L2 from inside the launch (no
    //             wave reads a tile before the ticket says it is complete), and lines cached by
    //             earlier launches were dropped by the kernel-boundary invalidate.
    // Results therefore do not depend on dispatch order or XCD placement; exercised with a warm
    // L1 and a reused workspace by tests/test_gpu_parity.py::test_scan_frame_slices_under_load.
    if (!SPILL && k.slices > 1) {
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
      if (*ticket != (unsigned int)(k.slices - 1)) { PT_ADD(3); PT_FLUSH(); return; }   // not the last: done
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

    PT_ADD(3);
    // ---- phase 2: chunks of centre rows [c0+q0, c0+q0+qn); mask row j <-> grid row c0+q0-1+j
    const int crows = c1 - c0;
    for (int q0r = 0; q0r < crows; q0r += k.chunk_rows) {
      const int qn = min(k.chunk_rows, crows - q0r);
      if constexpr (FB == 32) {
        // 2a (32-bit counters, small grids): FOUR lanes per (mask row, word), 16 cells each, joined
        // by two xor-shuffles.  Cells are visited in a rotated order so that the 64 lanes of a wave
        // (16 words that lie 64 counters apart x 4 quarters) hit 64 different LDS banks per step.
        const int lane = tid & 63;
        const int sub = lane & 3, rot = (lane >> 2) & 15;
        const int ntask = (qn + 2) * W * 4;
        for (int t0q = 0; t0q < ntask; t0q += BLOCK) {             // uniform trip count: shuffles below
          const int t = t0q + tid;
          const int tw = t >> 2;
          const int j = tw / W, w = tw - j * W;
          const int g = c0 + q0r - 1 + j;                // grid row of this mask row
          const int ncell = min(64, k.gw - w * 64) - sub * 16;   // cells of this lane's quarter inside the grid
          unsigned int q = 0u;                           // the quarter's 16 "active" bits
          if (t < ntask && g >= t0 && g < t1 && ncell > 0) {     // outside the grid = inactive
            const unsigned int *row = cnt + (size_t)(g - t0) * k.gw + w * 64 + sub * 16;
            const int last = min(ncell, 16) - 1;
            // four LDS reads in flight per step; the loop is NOT fully unrolled on purpose: unrolled,
            // the per-step lane constants (rotated column, its address, its 64-bit bit) of all 16
            // steps were hoisted out of the task loop and held ~64 VGPRs for the whole kernel
#pragma unroll 1
            for (int c = 0; c < 16; c += 4) {
              int cc[4];
              unsigned int v[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                cc[u] = (c + u + rot) & 15;
                v[u] = row[min(cc[u], last)];            // always inside the row: no branch around the read
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) q |= (unsigned int)(cc[u] <= last && v[u] >= k.active_min) << cc[u];
            }
          }
          unsigned long long m = (unsigned long long)q << (sub * 16);
          m |= __shfl_xor(m, 1);
          m |= __shfl_xor(m, 2);
          if (t < ntask && sub == 0) mask[(size_t)j * W + w] = m;
        }
      } else {
        // 2a (packed counters, big grids): one LANE per (mask row, word) — it reads the 2*FB
        // counter words that hold its 64 cells and squeezes the "active" bit of every field into
        // the mask (the wave-per-word ballot form took 140 us per 960x540 frame: 17 % of the
        // workgroup's life, un-overlapped whenever one workgroup owns the CU)
        const int ntask = (qn + 2) * W;
        for (int t = tid; t < ntask; t += BLOCK) {
          const int j = t / W, w = t - j * W;
          const int g = c0 + q0r - 1 + j;
          unsigned long long m = 0ull;
          if (g >= t0 && g < t1)
            m = mask_word<FB, MODE>(cnt, (unsigned int)((g - t0) * k.gw + w * 64), min(64, k.gw - w * 64), k.vec_need);
          mask[(size_t)j * W + w] = m;
        }
      }
      __syncthreads();
      {  // 2b: centre cells with an active 4-neighbour
        const int ntask = qn * W;
        for (int t = tid; t < ntask; t += BLOCK) {
          const int r = t / W, w = t - r * W;            // centre row c0+q0r+r -> mask row r+1
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
      __syncthreads();                                   // masks / counters are rewritten next
    }
    PT_ADD(4);
  }
  if (local) atomicAdd(total, local);
  __syncthreads();

  if (tid == 0) store_flag(flags, f, (*total >= k.clust_need) ? 1 : 0, k.sys_flags);
  PT_FLUSH();
}

// Grid: one workgroup per k.group consecutive items of the work list.  Small frames (below ~128 KB: a few
// microseconds of work) are grouped: the dispatcher starts ~19 workgroups per microsecond, which
// keeps too few such workgroups alive per CU to overlap their zeroing and cluster tests; a
// workgroup that scans a few frames in a row lives long enough (choose_group, mtgpu_api.hip).
// The grid is sized for "every frame has side data"; the workgroups past the end of the list find a kNoFrame entry
// and leave — all of them at the END of the grid, after the last workgroup with work, whatever the stream's key-frame
// period is.
template <int BLOCK, int UNROLL, int FB, int MODE, int VAR, int REC, bool SPILL>
__global__ __launch_bounds__(BLOCK) void scan_frames_kernel(
    const unsigned char *__restrict__ mv, const WorkItem *__restrict__ work,
    unsigned int item0, unsigned int n_items, ScanK k, unsigned char *__restrict__ flags,
    unsigned int *spill_q, unsigned int *slice_ws, unsigned int *tickets, unsigned int *next_ticket) {
  extern __shared__ __attribute__((aligned(16))) unsigned int lds[];
  NextStep<UNROLL> ns;
  ns.have = false;
  ns.frame = 0u;
  unsigned int first = item0 + blockIdx.x * (unsigned int)k.group;
  unsigned int nextv = 0u;
  unsigned int *slot = lds + (k.cnt_words + 2 * k.mask_rows * k.W + 4);       // (resident form: one word past the kernel's own LDS use)
  const bool resident = kExperiments && k.resident > 0;
  if (resident && threadIdx.x == 0)
    nextv = __hip_atomic_fetch_add(next_ticket, (unsigned int)k.group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (;;) {
    if (resident) {
      // A/B form (experiments build, MTGPU_RESIDENT = workgroups per CU): a fixed grid of workgroups pulls k.group
      // items at a time with one agent-scope atomic; the next ticket is on its way while the current items are
      // scanned.  The loop ends for every workgroup: tickets only grow, and the first one at or past n_items (or
      // the end of the list) is the last this workgroup takes.
      __syncthreads();
      if (threadIdx.x == 0) *slot = nextv;
      __syncthreads();
      first = *slot;
      if (first < n_items && threadIdx.x == 0)
        nextv = __hip_atomic_fetch_add(next_ticket, (unsigned int)k.group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (first >= n_items) return;
    // the first entry is asked for before anything else is set up: its latency overlaps the kernel's scalar prologue;
    // a workgroup that owns several frames (then slices == 1: items are list entries) fetches the others now, too
    WorkItem me = load_item(work, item_entry<SPILL>(first, k.slices));
    unsigned int *stage = lds + k.stage_word;
    if (k.group > 1) {
      if (resident) __syncthreads();                          // (the previous ticket's last reads of the staged entries)
      stage_items(work, first, n_items, k.group, stage);      // published by the first frame's first barrier
    }
    for (int g = 0; g < k.group; ++g) {
      const unsigned int item = first + (unsigned int)g;
      if (item >= n_items || me.f == kNoFrame) return;       // kNoFrame: the list has ended, every later item is past its end too
      const bool more = (g + 1 < k.group) && (item + 1u < n_items);
      // (no barrier between items: every LDS read of an item precedes its last barrier, and the
      //  next item's writes start with its own zeroing)
      scan_item<BLOCK, UNROLL, FB, MODE, VAR, REC, SPILL>(mv, me, stage + 8u * ((unsigned int)g + 1u), item, k, flags, spill_q,
                                                           slice_ws, tickets, lds, ns, more && k.prefetch);
      if (!more) break;
      me = staged_item(stage, (unsigned int)g + 1u);            // (parked before this workgroup's first barrier)
    }
    if (!resident) return;
  }
}

// ------------------------------------------------------------------ the work list (plan_frames)
//
// Ahead of every scan, over the batch's offsets only (8 bytes per frame, never the records): every frame is ranked
// among the frames WITH side data before it (ballot + popcount inside a wave, 16 wave totals through LDS, the blocks
// before this one added up) and written to
//     work[rank]                              = {r0, r1, f}                 a frame with side data
//     work[n_frames - 1 - #empty before f]    = kNoFrame, flags[f] = 0      one without (:219-221)
// so the list is: every frame with side data in stream order, then kNoFrame up to n_frames (+ one more: the entry the
// last frame's "next frame" read finds).  Deterministic: no atomics on global memory, no block waits for another.
// How a block learns the count of the blocks before it:
//   up to kPlanFused blocks (32 768 frames)   ONE kernel: every block counts the frames before its own again — at most
//                                             31 coalesced 8-byte loads per lane, cheaper than a second launch
//   more                                      plan_count_kernel writes one count per block, plan_scatter_kernel adds up
//                                             at most 1024 of them
// has_sd == NULL: a frame has side data iff it has records; has_sd[f] != 0 with no records is a frame whose side data
// is empty: it reaches the scan, which then runs the cluster test on an all-zero grid (vectors_needed == 0 matters).
__global__ __launch_bounds__(kPlanBlock) void plan_count_kernel(
    const unsigned long long *__restrict__ frame_off, const unsigned char *__restrict__ has_sd, unsigned long long n_records,
    unsigned int n_frames, unsigned int per, unsigned int *__restrict__ blk_cnt) {
  __shared__ unsigned int total;
  if (threadIdx.x == 0) total = 0u;
  __syncthreads();
  unsigned int mine = 0u;
  for (unsigned int i = 0; i < per; ++i) {
    const unsigned long long f = ((unsigned long long)blockIdx.x * per + i) * kPlanBlock + threadIdx.x;
    unsigned long long r0, r1;
    const bool sd = f < n_frames && plan_frame(frame_off, has_sd, n_records, f, r0, r1);
    mine += (unsigned int)__popcll(__ballot(sd));        // every lane of the wave holds the wave's count
  }
  if ((threadIdx.x & 63u) == 0u && mine != 0u) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = total;
}

__global__ __launch_bounds__(kPlanBlock) void plan_scatter_kernel(
    const unsigned long long *__restrict__ frame_off, const unsigned char *__restrict__ has_sd, unsigned long long n_records,
    unsigned long long rebase, unsigned int n_frames, unsigned int per,
    const unsigned int *__restrict__ blk_cnt, WorkItem *__restrict__ work, unsigned char *__restrict__ flags, int sys_flags,
    unsigned int *next_ticket) {
  constexpr unsigned int WAVES = kPlanBlock / 64u;
  __shared__ unsigned int wave_before[WAVES];    // frames with side data in the blocks before this one, as each wave counted them
  __shared__ unsigned int wave_cnt[2][WAVES];    // ... among this iteration's frames, per wave (double-buffered: one barrier per iteration)
  const unsigned int tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  // this block's first frames: loaded before the counting below, so that both wait for memory once
  unsigned long long f = (unsigned long long)blockIdx.x * per * kPlanBlock + tid;
  unsigned long long r0 = 0ull, r1 = 0ull;
  bool valid = f < n_frames;
  bool sd = valid && plan_frame(frame_off, has_sd, n_records, f, r0, r1);
  {
    unsigned int mine = 0u;                      // every lane ends up with its wave's count
    if (blk_cnt) {
      for (unsigned int j = tid; j < blockIdx.x; j += kPlanBlock) mine += blk_cnt[j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mine += (unsigned int)__shfl_xor((int)mine, o);
    } else {                                     // fused form (per == 1): count the frames of the blocks before this one
      // (blockIdx.x whole blocks of kPlanBlock frames: the trip count is uniform; four blocks' loads in flight per step)
      for (unsigned int j = 0; j < blockIdx.x; j += 4u) {
        bool s4[4];
#pragma unroll
        for (unsigned int u = 0; u < 4u; ++u) {
          unsigned long long a0, a1;
          s4[u] = (j + u < blockIdx.x) && plan_frame(frame_off, has_sd, n_records, (unsigned long long)(j + u) * kPlanBlock + tid, a0, a1);
        }
#pragma unroll
        for (unsigned int u = 0; u < 4u; ++u) mine += (unsigned int)__popcll(__ballot(s4[u]));
      }
    }
    if (lane == 0u) wave_before[wave] = mine;
  }
  unsigned int running = 0u;                     // frames with side data before the frames of this iteration
  for (unsigned int i = 0; i < per; ++i) {
    const unsigned long long b = __ballot(sd);
    if (lane == 0u) wave_cnt[i & 1u][wave] = (unsigned int)__popcll(b);
    __syncthreads();                             // (also publishes wave_before, first iteration)
    if (i == 0u) {
#pragma unroll
      for (unsigned int w = 0; w < WAVES; ++w) running += wave_before[w];
    }
    unsigned int below = 0u, all = 0u;           // waves before this one; the whole block
#pragma unroll
    for (unsigned int w = 0; w < WAVES; ++w) {
      const unsigned int c = wave_cnt[i & 1u][w];
      below += w < wave ? c : 0u;
      all += c;
    }
    const unsigned int rank = running + below + (unsigned int)__popcll(b & ((1ull << lane) - 1ull));   // side-data frames before f
    if (valid) {
      WorkItem it;
      it.pad[0] = it.pad[1] = it.pad[2] = 0u;
      if (sd) {
        it.r0 = r0 > rebase ? r0 - rebase : 0ull; it.r1 = r1 > rebase ? r1 - rebase : 0ull; it.f = (unsigned int)f;
        work[rank] = it;
      } else {
        it.r0 = it.r1 = 0ull; it.f = kNoFrame;
        work[(unsigned long long)(n_frames - 1u) - (f - rank)] = it;      // f - rank frames without side data before f
        store_flag(flags, (unsigned int)f, 0, sys_flags);                 // :219-221 — no side data: false
      }
    }
    running += all;
    if (i + 1u < per) {                          // the next kPlanBlock frames of this block
      f += kPlanBlock;
      valid = f < n_frames;
      sd = valid && plan_frame(frame_off, has_sd, n_records, f, r0, r1);
    }
  }
  if (blockIdx.x == 0 && tid == 0) {
    WorkItem it;
    it.r0 = it.r1 = 0ull; it.f = kNoFrame; it.pad[0] = it.pad[1] = it.pad[2] = 0u;
    work[n_frames] = it;                         // what the last frame's "next frame" read finds
    *next_ticket = 0u;                           // (resident form, experiments build)
  }
}

static hipError_t launch_plan(const ScanLaunch &L, WorkItem *work, unsigned int *blk_cnt, unsigned int *next_ticket) {
  const unsigned int per = plan_per(L.n_frames), blocks = plan_blocks(L.n_frames);
  const bool fused = blocks <= kPlanFused;       // (then per == 1)
  if (!fused) {
    hipLaunchKernelGGL(plan_count_kernel, dim3(blocks), dim3(kPlanBlock), 0, L.stream, L.frame_off, L.has_sd, L.n_records,
                       L.n_frames, per, blk_cnt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(plan_scatter_kernel, dim3(blocks), dim3(kPlanBlock), 0, L.stream, L.frame_off, L.has_sd, L.n_records,
                     L.rebase, L.n_frames, per, fused ? nullptr : blk_cnt, work, L.flags, L.k.sys_flags, next_ticket);
  return hipGetLastError();
}

// Opt-in precondition check of the device entry points (MTGPU_CHECK_OFFSETS=1): frame_off must be
// non-decreasing — frames are then disjoint record ranges.  A banded plan keeps frame f's spill queue at
// spill_q + frame_off[f], one slot per record, so frames that overlap would race on it.  *first_bad ends up as
// the smallest f with frame_off[f] > frame_off[f + 1] (0xffffffff: none).
__global__ __launch_bounds__(256) void check_offsets_kernel(const unsigned long long *__restrict__ frame_off,
                                                            unsigned int n_frames, unsigned int *first_bad) {
  for (unsigned long long f = (unsigned long long)blockIdx.x * 256ull + threadIdx.x; f < n_frames;
       f += (unsigned long long)gridDim.x * 256ull)
    if (frame_off[f] > frame_off[f + 1]) atomicMin(first_bad, (unsigned int)f);
}

hipError_t launch_check_offsets(const unsigned long long *frame_off, unsigned int n_frames, unsigned int *first_bad,
                                hipStream_t stream) {
  if (n_frames == 0) return hipSuccess;
  const unsigned int blocks = (n_frames + 255u) / 256u;
  hipLaunchKernelGGL(check_offsets_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, stream, frame_off,
                     n_frames, first_bad);
  return hipGetLastError();
}

// ------------------------------------------------------------------ launchers

template <int BLOCK, int FB, int MODE, int REC, bool SPILL, int UNROLL = 4, int VAR = 0>
static hipError_t launch_one(const ScanLaunch &L) {
  auto kern = scan_frames_kernel<BLOCK, UNROLL, FB, MODE, VAR, REC, SPILL>;
  // Dynamic-LDS ceiling: set ONCE per instantiation and device to the device maximum (host
  // threads sharing an instantiation must not race each other with per-launch values).
  static std::atomic<unsigned long long> ready{0ull};
  const unsigned long long bit = 1ull << (L.device & 63);
  if ((ready.load(std::memory_order_acquire) & bit) == 0ull) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_max);
    if (e != hipSuccess) return e;
    ready.fetch_or(bit, std::memory_order_release);
  }
  const unsigned long long items = (unsigned long long)L.n_frames * (unsigned long long)(SPILL ? 1 : L.k.slices);
  const unsigned long long group = (unsigned long long)(L.k.group > 0 ? L.k.group : 1);
  WorkItem *work = static_cast<WorkItem *>(L.plan_ws);
  unsigned int *next_ticket = reinterpret_cast<unsigned int *>(work + (size_t)L.n_frames + 1u) + plan_blocks(L.n_frames);
  if (kExperiments && L.k.resident > 0) {
    const unsigned long long want = (unsigned long long)(L.cu_count > 0 ? L.cu_count : 256) * (unsigned long long)L.k.resident;
    const unsigned long long wgs = (items + group - 1) / group;
    hipLaunchKernelGGL(kern, dim3((unsigned int)(wgs < want ? wgs : want)), dim3(BLOCK), L.lds_bytes + 16, L.stream, L.mv, work,
                       0u, (unsigned int)items, L.k, L.flags, L.spill_q, L.slice_ws, L.tickets, next_ticket);
    return hipGetLastError();
  }
  const unsigned long long chunk = L.item_chunk ? L.item_chunk : (1ull << 30);   // workgroups per launch: grid.x stays < 2^31
  for (unsigned long long i0 = 0; i0 < items; i0 += chunk * group) {
    const unsigned long long left = items - i0;
    const unsigned long long wgs = (left + group - 1) / group;
    const unsigned int n = (unsigned int)(wgs < chunk ? wgs : chunk);
    hipLaunchKernelGGL(kern, dim3(n), dim3(BLOCK), L.lds_bytes, L.stream, L.mv, work, (unsigned int)i0,
                       (unsigned int)items, L.k, L.flags, L.spill_q, L.slice_ws, L.tickets, next_ticket);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

#ifdef MTGPU_EXPERIMENTS
// Experiment variants of the ADD32 kernel on 40-byte records (MTGPU_VARIANT): bit0 UNROLL 8,
// bit1 dwordx4 loads, bit2 no nt hint, bit3 software-pipelined loop.  Experiments build only.
template <int BLOCK>
static hipError_t launch_variant(const ScanLaunch &L) {
  switch (L.variant & 15) {
#define MT_VARIANT_CASE(v) case v: return launch_one<BLOCK, 32, MODE_ADD32, 40, false, ((v) & 1) ? 8 : 4, (v)>(L);
    MT_VARIANT_CASE(1) MT_VARIANT_CASE(2) MT_VARIANT_CASE(3) MT_VARIANT_CASE(4) MT_VARIANT_CASE(5)
    MT_VARIANT_CASE(6) MT_VARIANT_CASE(7) MT_VARIANT_CASE(8) MT_VARIANT_CASE(9) MT_VARIANT_CASE(10)
    MT_VARIANT_CASE(11) MT_VARIANT_CASE(12) MT_VARIANT_CASE(13) MT_VARIANT_CASE(14) MT_VARIANT_CASE(15)
#undef MT_VARIANT_CASE
    default: return launch_one<BLOCK, 32, MODE_ADD32, 40, false>(L);
  }
}
#endif

template <int BLOCK, int REC, bool SPILL>
static hipError_t launch_form(const ScanLaunch &L) {
  const int key = L.k.mode * 100 + L.k.fb;
  switch (key) {
    case MODE_ADD32 * 100 + 32: return launch_one<BLOCK, 32, MODE_ADD32, REC, SPILL>(L);
    case MODE_UNARY * 100 + 1: return launch_one<BLOCK, 1, MODE_UNARY, REC, SPILL>(L);
    case MODE_UNARY * 100 + 2: return launch_one<BLOCK, 2, MODE_UNARY, REC, SPILL>(L);
    case MODE_UNARY * 100 + 4: return launch_one<BLOCK, 4, MODE_UNARY, REC, SPILL>(L);
    case MODE_UNARY * 100 + 8: return launch_one<BLOCK, 8, MODE_UNARY, REC, SPILL>(L);
    case MODE_CAS * 100 + 8: return launch_one<BLOCK, 8, MODE_CAS, REC, SPILL>(L);
    default: return hipErrorInvalidValue;
  }
}

// Default build: the instantiations a plan can select — single tiles with 512 or 1024 threads, banded plans with
// 1024 (make_plan, mtgpu_api.hip) — 36 kernels.  The experiments build adds 256-thread workgroups, 512-thread
// banded ones and the MTGPU_VARIANT load variants (117 kernels, three times the compile time and code size).
template <int BLOCK>
static hipError_t launch_block(const ScanLaunch &L) {
  const bool spill = L.k.bands > 1;
#ifdef MTGPU_EXPERIMENTS
  const int key = L.k.mode * 100 + L.k.fb;
  if (L.rec_bytes == 40 && !spill && key == MODE_ADD32 * 100 + 32 && (L.variant & 15) != 0 && BLOCK != 1024)
    return launch_variant<BLOCK>(L);
  constexpr bool kSpillHere = true;
#else
  constexpr bool kSpillHere = BLOCK == 1024;
#endif
  if constexpr (kSpillHere) {
    if (spill) return L.rec_bytes == 8 ? launch_form<BLOCK, 8, true>(L) : launch_form<BLOCK, 40, true>(L);
  } else {
    if (spill) return hipErrorInvalidValue;
  }
  return L.rec_bytes == 8 ? launch_form<BLOCK, 8, false>(L) : launch_form<BLOCK, 40, false>(L);
}

#ifdef MTGPU_PHASE_TIMES
hipError_t debug_set_phase_times(unsigned long long *p) { return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_times), &p, sizeof p); }
#endif

hipError_t launch_scan(const ScanLaunch &L) {
  if (L.n_frames == 0) return hipSuccess;
  if (L.rec_bytes != 40 && L.rec_bytes != 8) return hipErrorInvalidValue;
  hipError_t e;
  if (L.k.bands > 1 && (L.k.slices != 1 || !L.spill_q)) return hipErrorInvalidValue;
  // work items (frames x slices) are 32-bit inside the kernel (item0 + blockIdx.x * group)
  if ((unsigned long long)L.n_frames * (unsigned long long)(L.k.slices > 0 ? L.k.slices : 1) >= (1ull << 32))
    return hipErrorInvalidValue;
  if (L.k.slices > 1) {
    if (!L.slice_ws || !L.tickets) return hipErrorInvalidValue;
    e = hipMemsetAsync(L.tickets, 0, sizeof(unsigned int) * (size_t)L.n_frames, L.stream);
    if (e != hipSuccess) return e;
  }
  if (!L.plan_ws || ((uintptr_t)L.plan_ws & 31u) != 0u || !L.frame_off || L.rebase > L.n_records) return hipErrorInvalidValue;
  {
    WorkItem *work = static_cast<WorkItem *>(L.plan_ws);
    unsigned int *blk_cnt = reinterpret_cast<unsigned int *>(work + (size_t)L.n_frames + 1u);
    e = launch_plan(L, work, blk_cnt, blk_cnt + plan_blocks(L.n_frames));
    if (e != hipSuccess) return e;
  }
  // (profiling: the event between planning and scan)
  if (L.ev_planned && (e = hipEventRecord(L.ev_planned, L.stream)) != hipSuccess) return e;
  switch (L.block) {
#ifdef MTGPU_EXPERIMENTS
    case 256: e = launch_block<256>(L); break;
#endif
    case 512: e = launch_block<512>(L); break;
    case 1024: e = launch_block<1024>(L); break;
    default: return hipErrorInvalidValue;
  }
  return e;
}

}  // namespace mtgpu
