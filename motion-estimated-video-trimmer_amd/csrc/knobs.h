// knobs.h — environment knobs of libmtgpu.so.  Internal.
//
// Two classes (the supported ones are listed in include/mtgpu.h, "Environment"; the experiment ones below):
//   env_int()  supported knobs: production settings and the FORCE_* / INJECT_* switches the test-suite needs to
//              reach every kernel form and error path of the DEFAULT build;
//   exp_int()  A/B switches of measurements whose losing side is documented as losing: compiled in
//              only with `make EXTRA=-DMTGPU_EXPERIMENTS`; in the default build they are constants, the code
//              they guard folds away and the kernels they select are not instantiated.
//
// Experiment knobs (exp_int; IGNORED unless the library was built with `make -C csrc experiments`, whose
// mtgpu_version() ends in "+experiments"):
//    experiments   MTGPU_VARIANT (load variants of the 32-bit kernel), MTGPU_ALIGN=0 (streams start wherever the frame
//                  starts), MTGPU_PREFETCH=0 (no next-frame prefetch), MTGPU_RESIDENT=K (K ticketed resident workgroups
//                  per CU instead of one workgroup per work item), MTGPU_PIPE_STREAMS,
//                  MTGPU_PIPE_EAGER, MTGPU_EVENT_BLOCKING, MTGPU_MAX_TILE_KB, MTGPU_BAND_LDS_KB, MTGPU_MIN_LDS_KB,
//                  MTGPU_FORCE_CHUNK, MTGPU_PACK_NT, MTGPU_PACK_PREFETCH, MTGPU_FORCE_BLOCK=256
#pragma once
#include <cstdlib>

namespace mtgpu {

inline int env_int(const char *name, int dflt) {
  const char *v = std::getenv(name);
  return v ? std::atoi(v) : dflt;
}

#ifdef MTGPU_EXPERIMENTS
inline int exp_int(const char *name, int dflt) { return env_int(name, dflt); }
constexpr bool kExperiments = true;
#else
constexpr int exp_int(const char *, int dflt) { return dflt; }
constexpr bool kExperiments = false;
#endif

}  // namespace mtgpu
