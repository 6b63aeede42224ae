/* tv_oracle_c.c -- C / OpenMP restatement of the reference's CPU twin.  TEST INFRASTRUCTURE, NOT PRODUCT CODE:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it (same rule as tv_oracle.py).
 *
 * Why it exists next to the NumPy oracle: (1) a second, independently written implementation of the per-voxel
 * definitions that the NumPy oracle -- itself pinned to the reference's golden vectors -- is
 * cross-checked against (tests/test_oracle_c.py); (2) the multi-core CPU baseline SURVEY 8d asks for next to the
 * single-threaded NumPy one (the reference's NumPy path is single-threaded).
 *
 * Build: gcc -O3 -fopenmp -shared -fPIC oracle/tv_oracle_c.c -o oracle/_build/libtv_oracle_c.so  (oracle/build_c.py) */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct tvc_geom {
    long nz, m, ny, nx;
    int scheme;        /* 0 upwind, 1 downwind, 2 central, 3 hybrid */
    int nd, za, ta;    /* channels, z / time axis active */
    double wz, wt, sf; /* sqrt(reg_z_over_reg), sqrt(reg_time), sqrt(factor_reg_static) */
    const uint8_t* mask;
} tvc_geom;

#define REAL double
#define SUF _f64
#include "tv_oracle_c.inc"
#undef REAL
#undef SUF

#define REAL float
#define SUF _f32
#include "tv_oracle_c.inc"
#undef REAL
#undef SUF
