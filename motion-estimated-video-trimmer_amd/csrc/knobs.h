// knobs.h — environment knobs of libmtgpu.so.  Internal.
//
// Two classes (the table a user sees is in include/mtgpu.h, "Environment"):
//   env_int()  supported knobs: production settings and the FORCE_* / INJECT_* switches the test-suite needs to
//              reach every kernel form and error path of the DEFAULT build;
//   exp_int()  A/B switches of measurements whose losing side is documented as losing: compiled in
//              only with `make EXTRA=-DMTGPU_EXPERIMENTS`; in the default build they are constants, the code
//              they guard folds away and the kernels they select are not instantiated.
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
