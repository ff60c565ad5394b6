// pack_simd.h — the host copy-out loops (pack_simd.cpp).  Internal; the public face is
// mtgpu_pack_records / mtgpu_pack_records_with / mtgpu_pack_selected in include/mtgpu.h.
#pragma once
#include <stdint.h>

#include "../../include/mtgpu.h"   // MT_PACK_* constants

namespace mtgpu {
// bytes 6..13 of each 40-byte record -> 8-byte compact records, with the loop chosen for this CPU
// (MTGPU_PACK / MTGPU_PACK_NT / MTGPU_PACK_PREFETCH override, read once).
void pack_records(const unsigned char *mv, uint64_t n, unsigned char *out);
// the same with an explicit loop: impl_flags = MT_PACK_SCALAR | MT_PACK_AVX2 | MT_PACK_AVX512, optionally
// | MT_PACK_NT; prefetch = software-prefetch distance on the source in bytes (0 = none).
// Returns 0, or -1 when this CPU cannot run the requested loop (nothing is written then).
int pack_records_with(int impl_flags, const unsigned char *mv, uint64_t n, unsigned char *out, uint64_t prefetch);
// MT_PACK_* (| MT_PACK_NT) that pack_records uses in this process
int pack_selected();
}  // namespace mtgpu
