// pack_simd.cpp — the copy-out loop of the host dispatcher (pipe.hip: mtgpu_batch_add_frame) and of
// mtgpu_pack_records: bytes 6..13 (src_x, src_y, dst_x, dst_y — all check_frame reads,
// reference src/motion_scanner.cpp:246-256) of every 40-byte AVMotionVector -> one 8-byte compact
// record.  Data movement only: nothing is decided here.  It exists because the MV side data dies when
// the AVFrame is reused (src/motion_scanner.cpp:347), so every batched backend must copy it out first.
//
// A separate host translation unit (g++, no HIP): three loops chosen once per process by CPU feature
// (__builtin_cpu_supports) — the build machine's -march does not matter, each loop carries its own
// target attribute:
//   AVX-512BW  8 records per step: five 64-byte loads (320 bytes = exactly 8 records, never a byte
//              beyond them), two two-source word permutes + a blend + one masked permute gather the
//              8 x 4 int16 fields into ONE 64-byte line, written with one full-line store;
//   AVX2       4 records per 32-byte store, two stores per step (16-byte loads at +6 stay inside
//              their own record);
//   scalar     one 8-byte load and store per record (the round-3 loop).
// The destination is pinned staging that the GPU reads over PCIe next; it is never read by the CPU
// again, so the vector loops write it with NON-TEMPORAL full-line stores (no read-for-ownership of
// the destination line, no staging in the cache hierarchy); MTGPU_PACK_NT=0 selects ordinary
// stores, MTGPU_PACK=scalar|avx2|avx512 pins the loop, MTGPU_PACK_PREFETCH=<bytes> sets the
// software-prefetch distance on the source (0 = off).  Head records are packed one by one until the
// destination sits on a 64-byte line.
#include <immintrin.h>

#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <cstring>

#include "knobs.h"
#include "pack_simd.h"

namespace mtgpu {
namespace {

constexpr uint64_t kRec = 40, kOut = 8;

inline void pack_scalar(const unsigned char *mv, uint64_t n, unsigned char *out) {
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t v;
    std::memcpy(&v, mv + i * kRec + 6, 8);
    std::memcpy(out + i * kOut, &v, 8);
  }
}

// ---- AVX2: records r..r+3 -> one 32-byte vector
__attribute__((target("avx2"))) inline __m256i gather4_avx2(const unsigned char *s) {
  const __m128i r0 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 6));          // bytes 6..21 of record 0
  const __m128i r1 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + kRec + 6));
  const __m128i r2 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 2 * kRec + 6));
  const __m128i r3 = _mm_loadu_si128(reinterpret_cast<const __m128i *>(s + 3 * kRec + 6));
  return _mm256_set_m128i(_mm_unpacklo_epi64(r2, r3), _mm_unpacklo_epi64(r0, r1));
}

template <bool NT>
__attribute__((target("avx2"))) void pack_avx2(const unsigned char *mv, uint64_t n, unsigned char *out, uint64_t pf) {
  uint64_t i = 0;
  if ((reinterpret_cast<uintptr_t>(out) & 7u) == 0) {           // NT stores need their natural alignment
    const uint64_t head = ((64u - (reinterpret_cast<uintptr_t>(out) & 63u)) & 63u) / kOut;
    const uint64_t h = head < n ? head : n;
    pack_scalar(mv, h, out);
    i = h;
    for (; i + 8 <= n; i += 8) {
      const unsigned char *s = mv + i * kRec;
      if (pf) {
        _mm_prefetch(reinterpret_cast<const char *>(s + pf), _MM_HINT_NTA);
        _mm_prefetch(reinterpret_cast<const char *>(s + pf + 64), _MM_HINT_NTA);
        _mm_prefetch(reinterpret_cast<const char *>(s + pf + 128), _MM_HINT_NTA);
        _mm_prefetch(reinterpret_cast<const char *>(s + pf + 192), _MM_HINT_NTA);
        _mm_prefetch(reinterpret_cast<const char *>(s + pf + 256), _MM_HINT_NTA);
      }
      const __m256i a = gather4_avx2(s), b = gather4_avx2(s + 4 * kRec);
      __m256i *d = reinterpret_cast<__m256i *>(out + i * kOut);
      if (NT) { _mm256_stream_si256(d, a); _mm256_stream_si256(d + 1, b); }
      else { _mm256_store_si256(d, a); _mm256_store_si256(d + 1, b); }
    }
    if (NT) _mm_sfence();
  } else {
    for (; i + 4 <= n; i += 4)
      _mm256_storeu_si256(reinterpret_cast<__m256i *>(out + i * kOut), gather4_avx2(mv + i * kRec));
  }
  pack_scalar(mv + i * kRec, n - i, out + i * kOut);
}

// ---- AVX-512BW: 8 records (five source lines' worth, 160 words) -> one 64-byte line.
// Output word 4r+j (r = record 0..7, j = field 0..3) is source word 20r+3+j:
//   r0: c0 w3..6   r1: c0 w23..26   r2: c1 w11..14   r3: c1 w31, c2 w0..2
//   r4: c2 w19..22 r5: c3 w7..10    r6: c3 w27..30   r7: c4 w15..18
alignas(64) const uint16_t kIdxA[32] = {            // from (c0 | 32 + c1): output words 0..12
    3, 4, 5, 6, 23, 24, 25, 26, 32 + 11, 32 + 12, 32 + 13, 32 + 14, 32 + 31,
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
alignas(64) const uint16_t kIdxB[32] = {            // from (c2 | 32 + c3): output words 13..27
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 1, 2, 19, 20, 21, 22, 32 + 7, 32 + 8, 32 + 9, 32 + 10, 32 + 27, 32 + 28, 32 + 29, 32 + 30,
    0, 0, 0, 0};
alignas(64) const uint16_t kIdxC[32] = {            // from c4: output words 28..31
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 15, 16, 17, 18};

template <bool NT>
__attribute__((target("avx512f,avx512bw"))) void pack_avx512(const unsigned char *mv, uint64_t n, unsigned char *out,
                                                             uint64_t pf) {
  const __m512i ia = _mm512_load_si512(kIdxA), ib = _mm512_load_si512(kIdxB), ic = _mm512_load_si512(kIdxC);
  const __mmask32 from_b = 0x0FFFE000u, from_c4 = 0xF0000000u;
  const bool aligned = (reinterpret_cast<uintptr_t>(out) & 7u) == 0;
  uint64_t i = 0;
  if (aligned) {
    const uint64_t head = ((64u - (reinterpret_cast<uintptr_t>(out) & 63u)) & 63u) / kOut;
    i = head < n ? head : n;
    pack_scalar(mv, i, out);
  }
  for (; i + 8 <= n; i += 8) {
    const unsigned char *s = mv + i * kRec;
    if (pf) {
      _mm_prefetch(reinterpret_cast<const char *>(s + pf), _MM_HINT_NTA);
      _mm_prefetch(reinterpret_cast<const char *>(s + pf + 64), _MM_HINT_NTA);
      _mm_prefetch(reinterpret_cast<const char *>(s + pf + 128), _MM_HINT_NTA);
      _mm_prefetch(reinterpret_cast<const char *>(s + pf + 192), _MM_HINT_NTA);
      _mm_prefetch(reinterpret_cast<const char *>(s + pf + 256), _MM_HINT_NTA);
    }
    const __m512i c0 = _mm512_loadu_si512(s), c1 = _mm512_loadu_si512(s + 64), c2 = _mm512_loadu_si512(s + 128),
                  c3 = _mm512_loadu_si512(s + 192), c4 = _mm512_loadu_si512(s + 256);
    __m512i r = _mm512_permutex2var_epi16(c0, ia, c1);
    r = _mm512_mask_blend_epi16(from_b, r, _mm512_permutex2var_epi16(c2, ib, c3));
    r = _mm512_mask_permutexvar_epi16(r, from_c4, ic, c4);
    void *d = out + i * kOut;
    if (!aligned) _mm512_storeu_si512(d, r);
    else if (NT) _mm512_stream_si512(reinterpret_cast<__m512i *>(d), r);
    else _mm512_store_si512(d, r);
  }
  if (NT && aligned) _mm_sfence();
  pack_scalar(mv + i * kRec, n - i, out + i * kOut);
}

struct Choice {
  int impl = MT_PACK_SCALAR;
  bool nt = true;
  uint64_t prefetch = 0;
};

bool cpu_has(int impl) {
  switch (impl) {
    case MT_PACK_SCALAR: return true;
    case MT_PACK_AVX2: return __builtin_cpu_supports("avx2") != 0;
    case MT_PACK_AVX512: return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    default: return false;
  }
}

const Choice &choice() {          // read once; thread-safe (function-local static)
  static const Choice c = [] {
    Choice ch;
    __builtin_cpu_init();
    ch.impl = cpu_has(MT_PACK_AVX512) ? MT_PACK_AVX512 : cpu_has(MT_PACK_AVX2) ? MT_PACK_AVX2 : MT_PACK_SCALAR;
    if (const char *e = std::getenv("MTGPU_PACK")) {
      const int want = !std::strcmp(e, "scalar") ? MT_PACK_SCALAR : !std::strcmp(e, "avx2") ? MT_PACK_AVX2
                       : !std::strcmp(e, "avx512") ? MT_PACK_AVX512 : -1;
      if (want > 0 && cpu_has(want)) ch.impl = want;     // a loop this CPU cannot run is never selected
    }
    ch.nt = exp_int("MTGPU_PACK_NT", 1) != 0;                                  // experiments build only
    ch.prefetch = (uint64_t)std::max(0, exp_int("MTGPU_PACK_PREFETCH", 0));    // experiments build only
    return ch;
  }();
  return c;
}

}  // namespace

int pack_records_with(int impl_flags, const unsigned char *mv, uint64_t n, unsigned char *out, uint64_t prefetch) {
  const int impl = impl_flags & MT_PACK_IMPL_MASK;
  const bool nt = (impl_flags & MT_PACK_NT) != 0;
  // (the software prefetch may run past the end of the source: a prefetch never faults)
  if (!cpu_has(impl)) return -1;
  if (impl == MT_PACK_SCALAR || n < 16) { pack_scalar(mv, n, out); return 0; }
  if (impl == MT_PACK_AVX2) { nt ? pack_avx2<true>(mv, n, out, prefetch) : pack_avx2<false>(mv, n, out, prefetch); return 0; }
  nt ? pack_avx512<true>(mv, n, out, prefetch) : pack_avx512<false>(mv, n, out, prefetch);
  return 0;
}

int pack_selected() { const Choice &c = choice(); return c.impl | (c.nt && c.impl != MT_PACK_SCALAR ? MT_PACK_NT : 0); }

void pack_records(const unsigned char *mv, uint64_t n, unsigned char *out) {
  const Choice &c = choice();
  (void)pack_records_with(c.impl | (c.nt ? MT_PACK_NT : 0), mv, n, out, c.prefetch);
}

}  // namespace mtgpu
