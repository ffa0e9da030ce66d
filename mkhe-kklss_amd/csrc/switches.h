// switches.h -- the two kinds of environment variables this library reads.
//
//  * CONFIGURATION (MKHE_CFG_INT): a documented handful, listed in include/mkhe.h ("Environment"), read by every build.
//  * A/B INSTRUMENTATION (MKHE_AB_INT): every other MKHE_* variable of DESIGN.md section 6.  They select kernels and
//    fusions that a measurement decided between; results are bit-identical either way.  They exist only in the
//    -DMKHE_SWITCHES build (`make switches` -> ../lib/libmkhe_hip_switches.so, what tools/switch_matrix.sh and
//    tests/test_gpu_forced_paths.py load through MKHE_LIB): in the product library the macro IS its default and the
//    variable's name never reaches the binary, so a stray variable in a user's environment cannot select a round-1 kernel set.
#pragma once
#include <cstdlib>

namespace mkhe {
inline int env_int_raw(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
}

#define MKHE_CFG_INT(name, dflt) ::mkhe::env_int_raw(name, dflt)
#ifdef MKHE_SWITCHES
#define MKHE_AB_INT(name, dflt) ::mkhe::env_int_raw(name, dflt)
#else
#define MKHE_AB_INT(name, dflt) (dflt)
#endif
