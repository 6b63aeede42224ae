// tv_host.h -- host-side helpers shared by the translation units of libpytv4d_hip.so:
// error reporting, tv_geom -> DG, launch geometry, partial-sum reduction, dispatch tables.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <mutex>
#include <string>

#include "../../include/pytv4d.h"
#include "tv_device.h"

using namespace tv;

namespace tv { __global__ void k_reduce(const double* in, long long n, double* out); }

inline thread_local std::string g_err;
inline int fail(int code, const char* msg) {
    g_err = msg;
    return code;
}
inline int hipfail(hipError_t e, const char* where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return (int)e;
}
#define HIP_TRY(call)                                         \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return hipfail(e__, #call);    \
    } while (0)

// pitch_ok: does the calling entry point take pitched arrays (tv_geom::row_pitch / frame_pitch)?  One that does not refuses
// them here -- nothing reads a pitched array as a dense one.
inline int make_dg(const tv_geom* g, DG& d, bool pitch_ok = false) {
    if (g == nullptr) return fail(TV_E_ARG, "tv_geom is NULL");
    if (g->struct_size != (uint32_t)sizeof(tv_geom) || g->abi_version != TV_ABI_VERSION)
        return fail(TV_E_ARG, "tv_geom was built against another version of pytv4d.h (struct_size / abi_version mismatch): "
                              "rebuild the host against this library's header and call tv_geom_init()");
    if (g->nz < 1 || g->m < 1 || g->ny < 1 || g->nx < 1) return fail(TV_E_ARG, "every dimension must be >= 1");
    if (g->nz_global < g->nz || g->z0 < 0 || g->z0 + g->nz > g->nz_global)
        return fail(TV_E_ARG, "slab [z0, z0+nz) must lie inside [0, nz_global)");
    if (g->scheme < 0 || g->scheme > 3) return fail(TV_E_ARG, "unknown scheme");
    if (g->dtype != TV_F32 && g->dtype != TV_F64) return fail(TV_E_ARG, "unknown dtype");
    if (g->nz_global > 60000 || g->m > 65535 || g->ny > (1 << 24) || g->nx > (1 << 24))
        return fail(TV_E_ARG, "dimension too large for the launch grid");
    if ((long long)g->ny * g->nx > (1ll << 31)) return fail(TV_E_ARG, "a frame larger than 2^31 pixels does not fit the launch grid");
    if (!(g->reg_z_over_reg >= 0.0) || !(g->reg_time >= 0.0) || !(g->factor_reg_static >= 0.0))
        return fail(TV_E_ARG, "weights must be non-negative numbers");
    d.nz = (int)g->nz; d.m = (int)g->m; d.ny = (int)g->ny; d.nx = (int)g->nx;
    d.nzg = (int)g->nz_global; d.z0 = (int)g->z0;
    d.za = (g->nz_global > 1 && g->reg_z_over_reg > 0.0) ? 1 : 0;
    d.ta = (g->m > 1 && g->reg_time > 0.0) ? 1 : 0;
    const int per = (g->scheme == TV_HYBRID) ? 2 : 1;
    d.nd = per * (2 + d.za + d.ta);
    d.ch_z = 2 * per;
    d.ch_t = d.ch_z + (d.za ? per : 0);
    d.vl = (g->dtype == TV_F32) ? 4 : 2;
    d.z_two = (g->scheme == TV_CENTRAL && g->nz_global == 2) ? 1 : 0;
    d.t_two = (g->scheme == TV_CENTRAL && g->m == 2) ? 1 : 0;
    d.rp = (int)g->nx;
    d.pitched = 0;
    d.s_t = (long long)g->ny * g->nx;
    if (g->row_pitch != 0 || g->frame_pitch != 0) {
        const long long lane = (g->dtype == TV_F32) ? 4 : 2;                // elements per 16 bytes
        const long long rp = g->row_pitch != 0 ? g->row_pitch : g->nx;
        const long long fp = g->frame_pitch != 0 ? g->frame_pitch : (long long)g->ny * rp;
        if (rp < g->nx || rp > (1 << 24) || rp % lane != 0) return fail(TV_E_ARG, "row_pitch must be >= nx and a multiple of 16 bytes");
        if (fp < (long long)g->ny * rp || fp % lane != 0 || fp > (1ll << 31))
            return fail(TV_E_ARG, "frame_pitch must be >= ny * row_pitch, a multiple of 16 bytes and <= 2^31 elements");
        if (!pitch_ok) return fail(TV_E_ARG, "this entry point does not take pitched arrays (tv_geom::row_pitch / frame_pitch must be 0)");
        d.rp = (int)rp;
        d.s_t = fp;
        d.pitched = 1;
    }
    d.s_z = d.s_t * g->m;
    d.s_dz = d.s_z * d.nd;
    d.mask = g->mask_static;
    d.tf = g->time_factor;
    d.wv = g->time_weight_vol;
    d.wvp = g->time_weight_vol ? g->time_weight_prev : nullptr;
    d.wvn = g->time_weight_vol ? g->time_weight_next : nullptr;
    return 0;
}

template <typename T> inline WT<T> make_w(const tv_geom* g) {
    WT<T> w;
    w.wz = (T)std::sqrt(g->reg_z_over_reg);
    w.wt = (T)std::sqrt(g->reg_time);
    w.sf = (T)std::sqrt(g->factor_reg_static);
    return w;
}

struct LC { dim3 grid, block; long long nblocks; };
inline LC launch_cfg(const DG& d, int V, int planes) {
    const int nxv = (d.nx + V - 1) / V;
    int bx = 1;
    while (bx < nxv && bx < 64) bx <<= 1;
    const int by = 256 / bx;
    const long long tx = (nxv + bx - 1) / bx, ty = (d.ny + by - 1) / by;
    LC lc;
    lc.block = dim3(bx, by, 1);
    lc.grid = dim3((unsigned)(tx * ty), (unsigned)d.m, (unsigned)planes);
    lc.nblocks = tx * ty * d.m * planes;
    return lc;
}

static const int kFlatBlocks = 2048;    // grid-stride kernels: 256 CUs x 8 blocks
static const int kStage = 256;          // second-level partials

// layout of the scratch buffer: [partials ... nmax][stage kStage]
inline long long max_partials(const DG& d) {
    LC lc = launch_cfg(d, 1, d.nz + 2);
    // the one-sweep fix-up launches up to four classes whose block counts add up to ~3 x (256-column tiles) x (4-row
    // groups) x m x nz when the z-chunks are short and the frame is narrow: bound them explicitly
    const long long tiles = (long long)(((d.nx + 3) / (d.vl > 0 ? d.vl : 4) + 63) / 64 + 1) * ((d.ny + 3) / 4 + 1);
    long long n = 4 * tiles * (d.m + 2) * (d.nz + 2) + 4096;
    if (lc.nblocks > n) n = lc.nblocks;
    return n > kFlatBlocks ? n : kFlatBlocks;
}

inline int reduce_partials(double* ws, long long n, long long nmax, double* result, hipStream_t st) {
    if (n <= 4096) {
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, st, ws, n, result);
    } else {
        double* stage = ws + nmax;
        hipLaunchKernelGGL(k_reduce, dim3(kStage), dim3(256), 0, st, ws, n, stage);
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, st, stage, (long long)kStage, result);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

inline bool aligned16(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15u) != 0) return false;
    return true;
}

// call f.template operator()<S, T, V>() for the run-time (scheme, dtype, vec)
// lanes of a 16-byte vector for this dtype: `vec` at the call sites means "Nx is a multiple of it and every
// pointer is 16-byte aligned"
inline int vec_lanes(const tv_geom* g) { return g->dtype == TV_F32 ? 4 : 2; }
// may the 16-byte-lane instantiations run on this geometry?  Dense arrays: Nx must be a multiple of the lane; pitched arrays:
// the row pitch is one by construction (make_dg), the last lane of a row then holds pad columns (zeros in, zeros out)
inline bool rows_vectorisable(const tv_geom* g, const DG& d) { return d.pitched || d.nx % vec_lanes(g) == 0; }

template <typename F> inline int dispatch(int scheme, int dtype, bool vec, F&& f) {
#define TV_CASE(SC)                                                                  \
    case SC:                                                                         \
        if (dtype == TV_F32) {                                                       \
            if (vec) return f.template operator()<SC, float, 4>();                   \
            return f.template operator()<SC, float, 1>();                            \
        }                                                                            \
        if (vec) return f.template operator()<SC, double, 2>();                      \
        return f.template operator()<SC, double, 1>();
    switch (scheme) {
        TV_CASE(0) TV_CASE(1) TV_CASE(2) TV_CASE(3)
    }
#undef TV_CASE
    return fail(TV_E_ARG, "unknown scheme");
}

// ---- plane-marching fast path (tv_march.h): fp32, 16-byte lanes, M in {1,2,3,4,8,16} ---------------
// Tuning / debugging options: ONE explicit process-wide table.  Each entry is initialised once, when
// the library first looks at the table, from the environment variable of the same name; after that only
// tv_set_option() changes it -- no call reads the environment again, so a running host program cannot have its
// dispatch changed behind its back.  An option that was never set (neither way) takes the call site's default.
struct TvOption { const char* name; std::atomic<int> has; std::atomic<int> value; };
inline TvOption g_options[] = {
    {"TV_NO_MARCH", 0, 0}, {"TV_MARCH_MIN_PLANE_KB", 0, 0}, {"TV_ZCHUNK", 0, 0}, {"TV_NS_ZCHUNK", 0, 0}, {"TV_MARCH_D", 0, 0},
    {"TV_NO_MARCH_SUBGRAD", 0, 0}, {"TV_NO_MARCH_NORMAL", 0, 0}, {"TV_SCALAR_GATHER", 0, 0}, {"TV_NO_FUSED", 0, 0},
    {"TV_NO_FUSED_TWIN", 0, 0}, {"TV_FUSED_XW", 0, 0}, {"TV_FUSED_FORCE_TWIN", 0, 0}, {"TV_NO_FUSED_SUBGRAD", 0, 0},
    {"TV_NORMAL_KERNEL", 0, 0}, {"TV_D_KERNEL", 0, 0}, {"TV_DT_KERNEL", 0, 0}, {"TV_FUSED_MIN_KVOXELS", 0, 0}, {"TV_SG_KERNEL", 0, 0}, {"TV_SPARE", 0, 0}, {"TV_SG_ALIGNED", 0, 0},
    {"TV_SMALL_MAX_KVOXELS", 0, 0}, {"TV_NO_SMALL", 0, 0}, {"TV_SMALL_BLOCKS_PER_CU", 0, 0}, {"TV_SMALL_GENERIC", 0, 0}, {"TV_SMALL_TILES", 0, 0}, {"TV_SMALL_SITES", 0, 0}, {"TV_NS_NO_FIRST", 0, 0},
};
inline std::once_flag g_options_once;
inline TvOption* find_option(const char* name) {
    std::call_once(g_options_once, [] {
        for (TvOption& o : g_options) {
            const char* v = getenv(o.name);
            if (v && *v) { o.value = atoi(v); o.has = 1; }
        }
    });
    for (TvOption& o : g_options)
        if (strcmp(o.name, name) == 0) return &o;
    return nullptr;
}
inline int env_int(const char* name, int dflt) {
    const TvOption* o = find_option(name);
    return (o && o->has.load(std::memory_order_relaxed)) ? o->value.load(std::memory_order_relaxed) : dflt;
}
inline bool march_ok(const tv_geom* g, const DG& d, bool vec) {
    if (g->dtype != TV_F32 || !vec || d.nx < 128) return false;
    if (d.wv != nullptr) return false;                    // weight volume: one-site-per-thread kernels
    if (d.pitched) return false;                          // pitched arrays (interface version 4): streaming / one-sweep / one-site kernels
    if (env_int("TV_NO_MARCH", 0)) return false;
    // small planes (z/t neighbours one plane away stay L2-resident) are served better by the
    // one-site-per-thread kernels: measured on 512x512xM=1 (BASELINE config 1)
    if ((long long)d.s_z * 4 < (long long)env_int("TV_MARCH_MIN_PLANE_KB", 4096) * 1024) return false;
    return (d.m >= 1 && d.m <= 8) || d.m == 16;
}
// the marching ADJOINT (k_DT_march) is instantiated for double as well (round 4): the same rule with the plane measured in bytes -- for the
// case where it wins: hybrid with a plain store (tv_DT, 32x8x1024x1024 fp64: 3.6 - 3.8 ms = 0.64 - 0.68 against 4.4 - 4.5 one-site; the Nd = 4
// schemes are 10 - 20 % SLOWER marching, tv_DT_axpy is the same either way: profiles/r4_op_rooflines_f64_dt.txt)
inline bool march_dt_ok(const tv_geom* g, const DG& d, bool vec, bool plain_store) {
    // fp32: everything but the plain adjoint of the one-sided schemes, where the one-site kernel is 3 - 5 % ahead (64x8x1024x1024: upwind 2.16 - 2.18
    // against 2.23 - 2.31 ms, downwind 2.15 - 2.19 against 2.30 - 2.39; hybrid 3.6 against 4.3 and central 2.4 - 2.5 against 2.75 the other way)
    if (g->dtype == TV_F32) return march_ok(g, d, vec) && !(plain_store && (g->scheme == TV_UPWIND || g->scheme == TV_DOWNWIND));
    if (g->scheme != TV_HYBRID || !plain_store) return false;
    if (!vec || d.nx < 128 || d.wv != nullptr || d.pitched || env_int("TV_NO_MARCH", 0)) return false;
    if ((long long)d.s_z * 8 < (long long)env_int("TV_MARCH_MIN_PLANE_KB", 4096) * 1024) return false;
    return (d.m >= 1 && d.m <= 8) || d.m == 16;
}
inline int march_zchunk(const DG& d) {
    // planes per z-chunk: long chunks amortise the chunk prologue (and, for the one-sweep CP kernel, the
    // chunk-edge fix-up planes); short ones keep >= ~4096 blocks in flight.  TV_ZCHUNK overrides.
    int zc = env_int("TV_ZCHUNK", 0);
    if (zc <= 0) {
        long long tiles = (long long)((d.nx / d.vl + 63) / 64) * ((d.ny + 3) / 4);      // d.vl columns per 16-byte lane: 4 (fp32) / 2 (fp64)
        if (tiles < 1) tiles = 1;                        // frames narrower than one 4-column vector (tools/abi_validation.py)
        const long long want_chunks = (4096 + tiles - 1) / tiles;
        zc = (int)(d.nz / (want_chunks > 0 ? want_chunks : 1));
        if (zc > 32) zc = 32;
        if (zc < 4) zc = 4;
    }
    if (zc < 1) zc = 1;
    if (zc > d.nz) zc = d.nz;
    return zc;
}
inline LC march_cfg(const DG& d, int zchunk) {
    const long long tx = (d.nx / d.vl + 63) / 64, ty = (d.ny + 3) / 4, nch = (d.nz + zchunk - 1) / zchunk;
    LC lc;
    lc.block = dim3(64, 4, 1);
    lc.grid = dim3((unsigned)(tx * ty), (unsigned)nch, 1);
    lc.nblocks = tx * ty * nch;
    return lc;
}
template <typename F> inline int dispatch_sm(int scheme, int m, F&& f) {
#define TV_CASE_M(SC)                                              \
    case SC:                                                       \
        switch (m) {                                               \
            case 1: return f.template operator()<SC, 1>();         \
            case 2: return f.template operator()<SC, 2>();         \
            case 3: return f.template operator()<SC, 3>();         \
            case 4: return f.template operator()<SC, 4>();         \
            case 5: return f.template operator()<SC, 5>();         \
            case 6: return f.template operator()<SC, 6>();         \
            case 7: return f.template operator()<SC, 7>();         \
            case 8: return f.template operator()<SC, 8>();         \
            case 16: return f.template operator()<SC, 16>();       \
        }                                                          \
        break;
    switch (scheme) { TV_CASE_M(0) TV_CASE_M(1) TV_CASE_M(2) TV_CASE_M(3) }
#undef TV_CASE_M
    return fail(TV_E_ARG, "unsupported (scheme, M) for the marching path");
}
inline int check_x_halos(const tv_geom* g, const DG& d, const void* xp, const void* xn) {
    if (!d.za) return 0;
    const bool need_prev = (g->scheme != TV_UPWIND), need_next = (g->scheme != TV_DOWNWIND);
    if (need_prev && g->z0 > 0 && xp == nullptr) return fail(TV_E_HALO, "previous-slab halo plane required");
    if (need_next && g->z0 + g->nz < g->nz_global && xn == nullptr) return fail(TV_E_HALO, "next-slab halo plane required");
    return 0;
}
inline int check_y_halos(const tv_geom* g, const DG& d, const void* yp, const void* yn) {
    if (!d.za) return 0;
    // backward-looking adjoint (upwind-type, central) reads the previous slab; forward-looking the next
    const bool need_prev = (g->scheme != TV_DOWNWIND), need_next = (g->scheme != TV_UPWIND);
    if (need_prev && g->z0 > 0 && yp == nullptr) return fail(TV_E_HALO, "previous-slab gradient halo required");
    if (need_next && g->z0 + g->nz < g->nz_global && yn == nullptr) return fail(TV_E_HALO, "next-slab gradient halo required");
    return 0;
}

// ---- plane-marching launchers, defined in tv_march_D.hip / tv_march_DT.hip ---------------------------
namespace tvm {
int D_store(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb, float* dout);
int D_cp_dual(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
              float* q, float sigma, float inv_lambda, double* partials);
// sub-gradient passes on the marching path: pass 1 = |Dx| (0 -> +inf) on the local planes plus ghost planes,
// pass 2 = G from x and |Dx| (radius-1 schemes only); x halo buffers hold TWO planes each
int D_norms(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
            float* norms_ext, double* partials, int ghost_lo, int ghost_hi);
int subgrad_pass2(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st,
                  const float* norms_ext, float* G);
bool subgrad_pass2_ok(const tv_geom* g, const DG& d);
// x + rho D^T D x (radius-1 schemes, M <= 8): LIGHT marching kernel with the hybrid stencil; two-plane halo buffers
int D_normal_op(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
                float* out, float rho, double* partials);
int D_admm_zu(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
              float* z, float* u, float thresh, double* partials, int tform);
// streaming normal operator (tv_nstream.h): radius-1 schemes, fp32; two dot products
bool N_stream_ok(const tv_geom* g, const DG& d, bool vec);
// optional Chebyshev form of the epilogue (tv_cheb_step): out = [add +] x + alpha (b - A x) + beta (x - y)
struct NCheb { const void* y; const void* add; const void* ref; double alpha, beta, yscale; };
int N_stream(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, const void* b, void* out, void* out2,
             double rho, hipStream_t st, long long* nblocks, double* part0, double* part1, const NCheb* cheb = nullptr);
// streaming forward kernel (tv_dstream.h): d = D x without LDS tile or barrier, every load one plane ahead of its use
bool D_stream_ok(const tv_geom* g, const DG& d, bool vec);
int D_stream(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, void* dout);
// (fp32 and fp64: the arrays are of g->dtype)
int DT_store(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb, void* out);
int DT_axpy(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb,
            void* out, const void* base, double alpha, const void* base2 = nullptr, double beta = 0.0);
int DT_cp_primal(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb,
                 float* x, const float* x0, float* p, float tau, float sigma_a, float inv_1p_sigma_a, double* partials);
}  // namespace tvm
