// mp_diag.h — A/B and test switches live in the DIAGNOSTICS build only.
//
// Rounds 1-4 left some twenty getenv("MP_...") switches in the library a host program links: other kernels for the same step (A/B
// measurements), forced code paths for tests, one (MP_K1_MT_FLAGS) after which reads returned undefined values.  The product library
// reads NO environment variable: every such switch goes through mp_diag_env(), which is getenv() when the library is compiled with
// -DMP_DIAGNOSTICS (modppl_amd/build.py build_diag -> libmodppl_hip_diag.so: same sources, same kernels; loaded by the tests and tools that
// need a switch — tests/conftest.py `diag`, tools/ab_env.sh — never by the product path) and a constant null otherwise, so that the
// code behind it compiles to its default in libmodppl_hip.so.
#pragma once
#include <cstdlib>

static inline const char* mp_diag_env(const char* name) {
#ifdef MP_DIAGNOSTICS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
