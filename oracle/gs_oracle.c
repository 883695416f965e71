/*
 * oracle/gs_oracle.c -- instantiates oracle/gs_oracle_impl.h for float (gsf_*: parity oracle) and double (gsd_*: finite-
 * difference reference for the analytic backward).  TEST INFRASTRUCTURE ONLY.  See gs_oracle_impl.h for the citations and
 * the "parity unpinned" statement.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int IMIN(int a, int b) { return a < b ? a : b; }
static inline int IMAX(int a, int b) { return a > b ? a : b; }

#define REAL float
#define FN(name) gsf_##name
#define SQRT sqrtf
#define EXP expf
#define CEIL ceilf
#define FMIN fminf
#define FMAX fmaxf
#define FMA fmaf
#include "gs_oracle_impl.h"
#undef REAL
#undef FN
#undef SQRT
#undef EXP
#undef CEIL
#undef FMIN
#undef FMAX
#undef FMA
#undef GS_BLEND_POWER

#define REAL double
#define FN(name) gsd_##name
#define SQRT sqrt
#define EXP exp
#define CEIL ceil
#define FMIN fmin
#define FMAX fmax
#define FMA fma
#include "gs_oracle_impl.h"


/* thread count of the OpenMP loops of the whole oracle library (0 = all host cores); returns the previous maximum */
int oracle_set_threads(int n) {
    const int before = omp_get_max_threads();
    omp_set_num_threads(n > 0 ? n : omp_get_num_procs());
    return before;
}
