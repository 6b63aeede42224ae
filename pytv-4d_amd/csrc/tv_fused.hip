// tv_fused.hip -- instantiations + C-ABI of the one-sweep Chambolle-Pock iteration (tv_fused.h).
#include "tv_fused_launch.h"

TV_FUSED_INSTANTIATE(float, ALG_CP)

// every M <= 8 has its own instantiation; more frames run as time windows of CP_TWN = 8 frames (tv_fused.h)
static bool fused_m_ok(int m) { return m >= 1; }

// planes per z-chunk of the one-sweep path (sweep, fix-up and the host's interior-first schedule share it): its blocks are
// CP_TR rows x CP_BC columns, half as many per plane as the marching kernels' -- with march_zchunk's rule (round 1) BASELINE
// config 2 ran on 8-plane chunks and config 1 on 16, 3 - 5 % slower than with 16 - 32 / 32 - 64 (tools/ab_cp.py --shape).
// Long chunks mean fewer chunk-edge fix-up planes; >= ~1024 blocks keep the 256 CUs busy for several rounds; a slab (one rank
// of a sharded volume) keeps >= 4 chunks so that the halo exchanges can hide behind interior chunks.  TV_ZCHUNK overrides.
static int fused_zchunk(const DG& d) {
    int zc = env_int("TV_ZCHUNK", 0);
    if (zc <= 0) {
        // (blocks per plane set: tiles x time windows -- counted since round 5: the configs[4] slab, 32 planes x 16 frames, ran in chunks of 16
        // where 32 is 2 - 4 % faster, ADMM hybrid 14.48 -> 14.18 ms per outer iteration, profiles/r5d_zchunk_sensitivity.txt)
        const long long nwin = (d.m > CP_TWN) ? (d.m + CP_TWN - 1) / CP_TWN : 1;
        const long long tiles = (long long)(((d.nx + d.vl - 1) / d.vl + CP_NW * CP_TL - 1) / (CP_NW * CP_TL)) * ((d.ny + CP_TR - 1) / CP_TR) * nwin;
        const long long want = (1024 + tiles - 1) / (tiles > 0 ? tiles : 1);
        zc = (int)(d.nz / (want > 0 ? want : 1));
        if (zc > 32) zc = 32;
        if (zc < 8) zc = 8;
        // a few planes of small frames (the reference's own shapes): below one block per CU shorter chunks win although every chunk
        // edge is fix-up work -- ADMM + Chebyshev on 20x4x100x100 0.29 -> 0.19 ms per outer iteration with 2-plane chunks, 256x4x100x100
        // (416 blocks at 8 planes) unchanged: profiles/r5_small_frames.txt
        while (zc > 2 && tiles * ((d.nz + zc - 1) / zc) < 256) zc -= 2;
        if (d.nz < d.nzg) {                              // a slab: interior chunks for the overlap
            const int q4 = (int)(d.nz / 4);
            if (zc > q4) zc = q4 < 4 ? 4 : q4;
        }
    }
    if (zc < 1) zc = 1;
    if (zc > d.nz) zc = (int)d.nz;
    return zc;
}

namespace tvm {
bool subgrad_pass2_ok(const tv_geom* g, const DG& d) { return g->scheme != TV_CENTRAL && d.m >= 1 && d.m <= 8; }
int subgrad_pass2(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st,
                  const float* norms_ext, float* G) {
    const int zc = march_zchunk(d);
    const LC lc = march_cfg(d, zc);
    return dispatch_fused(g->scheme, d.m, [&]<int S, int M>() -> int {
        if constexpr (S != CENTRAL && M >= 1) {
            hipLaunchKernelGGL((k_subgrad_march<S, M>), lc.grid, lc.block, 0, st, d, make_w<float>(g), (const float*)x,
                               (const float*)xp, (const float*)xn, norms_ext, G, zc);
            HIP_TRY(hipGetLastError());
            return 0;
        } else {
            return fail(TV_E_ARG, "central: radius-2 gather, not a marching kernel");
        }
    });
}
}  // namespace tvm

// launch geometry of a sweep over the chunks [chunk_begin, chunk_begin + chunk_count) (count < 0: all); empty: nothing to launch
struct SweepPlan {
    LC lc;
    int zc, chunk0;
    long long nmax;
    bool xw, force_win;
};
static int sweep_plan(const tv_geom* g, const DG& d, const void* x_in, const void* x_prev, const void* x_next, int64_t chunk_begin,
                      int64_t chunk_count, SweepPlan& sp, bool& empty) {
    sp.nmax = max_partials(d);
    sp.zc = fused_zchunk(d);
    const int zc = sp.zc;
    {   // halos are only needed by the chunks that touch the slab boundary
        const long long nch_all = (d.nz + zc - 1) / zc;
        const long long cb = (chunk_count < 0) ? 0 : chunk_begin, ce = (chunk_count < 0) ? nch_all : chunk_begin + chunk_count;
        const bool first = (cb == 0 && ce > 0), last = (ce == nch_all && ce > cb);
        if (int rc = check_x_halos(g, d, first ? x_prev : x_in, last ? x_next : x_in)) return rc;
    }
    LC& lc = sp.lc;
    lc = march_cfg(d, zc);
    {   // block tile of the sweep: CP_TR rows x CP_BC columns (tv_fused.h)
        const long long tx = ((d.nx + d.vl - 1) / d.vl + CP_NW * CP_TL - 1) / (CP_NW * CP_TL), ty = (d.ny + CP_TR - 1) / CP_TR;
        lc.grid.x = (unsigned)(tx * ty);
        lc.block = dim3(64, CP_NW, 1);
        lc.nblocks = tx * ty * lc.grid.y;
    }
    const long long nch = lc.grid.y;
    if (chunk_count < 0) { chunk_begin = 0; chunk_count = nch; }
    if (chunk_begin < 0 || chunk_begin + chunk_count > nch) return fail(TV_E_ARG, "chunk range outside the slab");
    empty = (chunk_count == 0);
    if (empty) return 0;
    const int nwin = (d.m > CP_TWN) ? (d.m + CP_TWN - 1) / CP_TWN : 1;      // time windows (grid z)
    lc.grid.y = (unsigned)chunk_count;
    lc.grid.z = (unsigned)nwin;
    lc.nblocks = (long long)lc.grid.x * chunk_count * nwin;
    if (lc.nblocks > sp.nmax) return fail(TV_E_ARG, "internal: sweep partials exceed the workspace");
    sp.chunk0 = (int)chunk_begin;
    sp.xw = env_int("TV_FUSED_XW", 1) != 0;
    // M == 8, hybrid: the windowed instantiation (one window) needs 234 VGPRs and no scratch where the plain one sits at
    // 256 + 8 B/lane, and is 1 ms faster per sweep on the north-star volume (33.7 vs 34.8 ms); the other schemes are
    // 2 % faster with the plain one (measured).  TV_FUSED_FORCE_TWIN=0/1 overrides.
    // central (round 3, once its windowed form had the dual prefetch): 253 VGPRs / no scratch against 256 + 20 B/lane, 22.7 vs 23.5 ms
    sp.force_win = (d.m == CP_TWN) && env_int("TV_FUSED_FORCE_TWIN", (g->scheme == TV_HYBRID || g->scheme == TV_CENTRAL) ? 1 : 0);
    return 0;
}

// launch geometry of the fix-up over the local planes [z_begin, z_begin + z_count) (count < 0: all)
static int fixup_plan(const tv_geom* g, const DG& d, const void* q, const void* q_prev, const void* q_next, int64_t z_begin, int64_t z_count,
                      FixPlan& fp, long long& nmax, bool& empty) {
    nmax = max_partials(d);
    const int zc = fused_zchunk(d);
    if (z_count < 0) { z_begin = 0; z_count = d.nz; }
    if (z_begin < 0 || z_begin + z_count > d.nz) return fail(TV_E_ARG, "plane range outside the slab");
    empty = (z_count == 0);
    if (empty) return 0;
    if (int rc = check_y_halos(g, d, (z_begin == 0) ? q_prev : q, (z_begin + z_count == d.nz) ? q_next : q)) return rc;
    const int zb = (int)z_begin, zn = (int)z_count;
    const int chunk_lo = zb / zc, chunk_hi = (zb + zn - 1) / zc;           // chunks intersecting the plane range
    const long long tiles_x = ((d.nx + d.vl - 1) / d.vl + 63) / 64, tiles_y = (d.ny + 3) / 4;
    const long long ngrp = (g->scheme == TV_HYBRID || g->scheme == TV_CENTRAL) ? (d.ny + 2 * CP_TR - 1) / (2 * CP_TR)
                                                                                : (d.ny + 4 * CP_TR - 1) / (4 * CP_TR);
    const bool xw = env_int("TV_FUSED_XW", 1) != 0;
    const long long bc = (long long)CP_NW * CP_TL * d.vl, wc = (long long)CP_TL * d.vl;      // block / wave tile width in columns
    const long long ncand = xw ? 2ll * ((d.nx + bc - 1) / bc) : 2ll * ((d.nx + wc - 1) / wc);
    fp.g0 = dim3((unsigned)(tiles_x * ngrp), (unsigned)d.m, (unsigned)zn);
    fp.g1 = dim3((unsigned)(tiles_x * tiles_y), (unsigned)d.m, (unsigned)(2 * (chunk_hi - chunk_lo + 1)));
    fp.g2 = dim3((unsigned)(((long long)d.ny * ncand + 255) / 256), (unsigned)d.m, (unsigned)zn);
    const int nwin = (d.m > CP_TWN && d.ta) ? (d.m + CP_TWN - 1) / CP_TWN : 0;     // time-window seams (M > 8)
    fp.g3 = dim3((unsigned)(tiles_x * tiles_y), (unsigned)(2 * (nwin > 0 ? nwin : 1)), (unsigned)zn);
    fp.n0 = (long long)fp.g0.x * fp.g0.y * fp.g0.z;
    fp.n1 = d.za ? (long long)fp.g1.x * fp.g1.y * fp.g1.z : 0;
    fp.n2 = (long long)fp.g2.x * fp.g2.y * fp.g2.z;
    fp.n3 = nwin > 0 ? (long long)fp.g3.x * fp.g3.y * fp.g3.z : 0;
    if (fp.n0 + fp.n1 + fp.n2 + fp.n3 > nmax) return fail(TV_E_ARG, "internal: fix-up partials exceed the workspace");
    fp.chunk_lo = chunk_lo; fp.zc = zc; fp.zb = zb; fp.zn = zn; fp.xw = xw;
    return 0;
}

extern "C" {

int tv_cp_fused_supported(const tv_geom* g) {
    DG d;
    if (make_dg(g, d, true)) return 0;
    // fp32, and fp64 since round 3 (2 columns per lane); a dense array needs whole lanes, a pitched one has them by construction
    // (the last lane of a ragged row holds pad columns: zeros in, zeros out -- round 4)
    if ((!d.pitched && d.nx % d.vl != 0) || d.nx < 64 || !fused_m_ok(d.m)) return 0;
    if (d.m > CP_TWN && env_int("TV_NO_FUSED_TWIN", 0)) return 0;
    // frames below 2^31 bytes: per-lane byte offsets are 32-bit, the PFX prefetch addresses x0 / p through a per-frame buffer descriptor
    // (num_records = frame bytes as a 32-bit number: 2^32 would wrap to 0) and BUF_OOB = 0x80000000 must lie outside the frame (round-5 advice)
    if (d.s_t * (16 / d.vl) >= (1ll << 31)) return 0;
    if (env_int("TV_NO_FUSED", 0)) return 0;
    return 1;
}

int tv_cp_zchunk(const tv_geom* g) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    return fused_zchunk(d);
}

static int cp_sweep_impl(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, const void* q_in, void* q, const void* x0,
                         void* p, void* x_out, double sigma_D, double lambda, double tau, double sigma_A, int32_t flags, int64_t chunk_begin,
                         int64_t chunk_count, double* tvout, double* fid, void* ws, void* stream);

int tv_cp_fused(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, void* q, const void* x0,
                void* p, void* x_out, double sigma_D, double lambda, double tau, double sigma_A, int64_t chunk_begin,
                int64_t chunk_count, double* tvout, double* fid, void* ws, void* stream) {
    return cp_sweep_impl(g, x_in, x_prev, x_next, q, q, x0, p, x_out, sigma_D, lambda, tau, sigma_A, 0, chunk_begin, chunk_count, tvout, fid, ws, stream);
}

int tv_cp_sweep(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, const void* q_in, void* q_out, const void* x0,
                void* p, void* x_out, double sigma_D, double lambda, double tau, double sigma_A, int32_t flags, int64_t chunk_begin,
                int64_t chunk_count, double* tvout, double* fid, void* ws, void* stream) {
    if (flags & ~(TV_CP_FID_OF_INPUT | TV_CP_FID_BOTH)) return fail(TV_E_ARG, "tv_cp_sweep: unknown flag");
    if ((flags & TV_CP_FID_BOTH) && !(flags & TV_CP_FID_OF_INPUT)) return fail(TV_E_ARG, "tv_cp_sweep: TV_CP_FID_BOTH extends TV_CP_FID_OF_INPUT");
    if (q_in == nullptr) return fail(TV_E_ARG, "NULL array");
    if (!aligned16({q_in})) return fail(TV_E_ARG, "arrays must be 16-byte aligned");
    return cp_sweep_impl(g, x_in, x_prev, x_next, q_in, q_out, x0, p, x_out, sigma_D, lambda, tau, sigma_A, flags, chunk_begin, chunk_count, tvout, fid, ws, stream);
}

static int cp_sweep_impl(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, const void* q_in, void* q, const void* x0,
                         void* p, void* x_out, double sigma_D, double lambda, double tau, double sigma_A, int32_t flags, int64_t chunk_begin,
                         int64_t chunk_count, double* tvout, double* fid, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!x_in || !q || !x0 || !p || !x_out || !tvout || !fid || !ws) return fail(TV_E_ARG, "NULL array");
    if (x_in == x_out) return fail(TV_E_ARG, "x_in and x_out must be different buffers (ping-pong)");
    if (!(lambda > 0.0)) return fail(TV_E_ARG, "lambda must be > 0");
    if (!tv_cp_fused_supported(g)) return fail(TV_E_ARG, "geometry not supported by the one-sweep path");
    if (!aligned16({x_in, x_prev, x_next, q, x0, p, x_out, d.wv})) return fail(TV_E_ARG, "arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SweepPlan sp;
    bool empty = false;
    if (int rc = sweep_plan(g, d, x_in, x_prev, x_next, chunk_begin, chunk_count, sp, empty)) return rc;
    if (empty) {
        HIP_TRY(hipMemsetAsync(tvout, 0, sizeof(double), st));
        HIP_TRY(hipMemsetAsync(fid, 0, ((flags & TV_CP_FID_BOTH) ? 2 : 1) * sizeof(double), st));
        return 0;
    }
    double* w0 = (double*)ws;
    double* w1 = w0 + sp.nmax + kStage + 16;
    double* w2 = w1 + sp.nmax + kStage + 16;          // third partial array (tv_workspace_bytes): TV_CP_FID_BOTH only
    auto sweep = [&]<typename T>() -> int {
        FusedArgsT<T> a{(const T*)x_in, (const T*)x_prev, (const T*)x_next, (T*)q, (const T*)x0, (T*)p,
                        (T*)x_out, (T)sigma_D, (T)(1.0 / lambda), (T)tau, (T)sigma_A, (T)(1.0 / (1.0 + sigma_A)), w0, w1,
                        ((flags & TV_CP_FID_OF_INPUT) ? 2 : 0) | ((flags & TV_CP_FID_BOTH) ? 4 : 0), (const T*)(q_in ? q_in : q), w2};
        return tvm::fused_sweep<T, ALG_CP>(g, d, sp.lc, st, a, sp.zc, sp.chunk0, sp.xw, sp.force_win);
    };
    const int rc = (g->dtype == TV_F32) ? sweep.template operator()<float>() : sweep.template operator()<double>();
    if (rc) return rc;
    if (int r2 = reduce_partials(w0, sp.lc.nblocks, sp.nmax, tvout, st)) return r2;
    if (int r3 = reduce_partials(w1, sp.lc.nblocks, sp.nmax, fid, st)) return r3;
    if (flags & TV_CP_FID_BOTH) return reduce_partials(w2, sp.lc.nblocks, sp.nmax, fid + 1, st);
    return 0;
}

int tv_cp_fixup(const tv_geom* g, const void* q, const void* q_prev, const void* q_next, void* x_out, const void* x0,
                double tau, int64_t z_begin, int64_t z_count, double* fid, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!q || !x_out || !fid || !ws) return fail(TV_E_ARG, "NULL array");      // x0 may be NULL: no fidelity (*fid = 0)
    if (!tv_cp_fused_supported(g)) return fail(TV_E_ARG, "geometry not supported by the one-sweep path");
    if (!aligned16({q, q_prev, q_next, x_out, x0})) return fail(TV_E_ARG, "arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    FixPlan fp;
    long long nmax = 0;
    bool empty = false;
    if (int rc = fixup_plan(g, d, q, q_prev, q_next, z_begin, z_count, fp, nmax, empty)) return rc;
    if (empty) {
        HIP_TRY(hipMemsetAsync(fid, 0, sizeof(double), st));
        return 0;
    }
    double* w0 = (double*)ws;
    auto fix = [&]<typename T>() -> int {
        FixupArgsT<T> a{(const T*)q, (const T*)q_prev, (const T*)q_next, (T*)x_out, (const T*)x0, (T)tau, fp.chunk_lo};
        return tvm::fused_fixup<T, ALG_CP>(g, d, st, a, fp, w0);
    };
    if (int rc = (g->dtype == TV_F32) ? fix.template operator()<float>() : fix.template operator()<double>()) return rc;
    if (x0 == nullptr) {                 // no fidelity asked for (tv_cp_sweep with TV_CP_FID_OF_INPUT delivers it): the partials are meaningless
        HIP_TRY(hipMemsetAsync(fid, 0, sizeof(double), st));
        return 0;
    }
    return reduce_partials(w0, fp.n0 + fp.n1 + fp.n2 + fp.n3, nmax, fid, st);
}

// One-sweep ADMM (tv_fused.h, ALG_ADMM): the z / u update of the outer iteration that ends and the residual of the x-solve that
// starts, from ONE pass over u (build-defined, the reference ships no ADMM; README.md:26,135).
int tv_admm_fused(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* u, void* t, const void* x0, void* r,
                  double thresh, double rho, int32_t full_store, int64_t chunk_begin, int64_t chunk_count, double* tvout, double* rr, void* ws,
                  void* stream) {
    return tv_admm_sweep(g, x, x_prev, x_next, u, u, t, x0, r, thresh, rho, full_store, chunk_begin, chunk_count, tvout, rr, ws, stream);
}

// tv_admm_fused with the dual variable read from u_in and written to u_out (round 5; equal: in place).  A solver that keeps both -- swapping
// them every outer iteration -- can rebuild z = shrink(D x + u_in) on demand, so the sweep need not store every sample of t' (Nd words less)
int tv_admm_sweep(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, const void* u_in, void* u, void* t, const void* x0,
                  void* r, double thresh, double rho, int32_t full_store, int64_t chunk_begin, int64_t chunk_count, double* tvout, double* rr,
                  void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!x || !u_in || !u || !t || !x0 || !r || !tvout || !rr || !ws) return fail(TV_E_ARG, "NULL array");
    if (x == r || x0 == r) return fail(TV_E_ARG, "r must not alias x or x0");
    if (u == t || u_in == t) return fail(TV_E_ARG, "u and t must be different arrays");
    if (!(thresh >= 0.0)) return fail(TV_E_ARG, "thresh must be >= 0");
    if (!tv_cp_fused_supported(g)) return fail(TV_E_ARG, "geometry not supported by the one-sweep path");
    if (!aligned16({x, x_prev, x_next, u_in, u, t, x0, r, d.wv})) return fail(TV_E_ARG, "arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SweepPlan sp;
    bool empty = false;
    if (int rc = sweep_plan(g, d, x, x_prev, x_next, chunk_begin, chunk_count, sp, empty)) return rc;
    if (empty) {
        HIP_TRY(hipMemsetAsync(tvout, 0, sizeof(double), st));
        HIP_TRY(hipMemsetAsync(rr, 0, sizeof(double), st));
        return 0;
    }
    double* w0 = (double*)ws;
    double* w1 = w0 + sp.nmax + kStage + 16;
    auto sweep = [&]<typename T>() -> int {
        FusedArgsT<T> a{(const T*)x, (const T*)x_prev, (const T*)x_next, (T*)u, (const T*)x0, (T*)t,
                        (T*)r, (T)thresh, (T)0, (T)rho, (T)0, (T)0, w0, w1, (int)(full_store & 3), (const T*)u_in};
        return tvm::fused_sweep<T, ALG_ADMM>(g, d, sp.lc, st, a, sp.zc, sp.chunk0, sp.xw, sp.force_win);
    };
    const int rc = (g->dtype == TV_F32) ? sweep.template operator()<float>() : sweep.template operator()<double>();
    if (rc) return rc;
    if (int r2 = reduce_partials(w0, sp.lc.nblocks, sp.nmax, tvout, st)) return r2;
    return reduce_partials(w1, sp.lc.nblocks, sp.nmax, rr, st);
}

int tv_admm_fixup(const tv_geom* g, const void* t, const void* t_prev, const void* t_next, void* r, double rho, int64_t z_begin,
                  int64_t z_count, double* rr, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!t || !r || !rr || !ws) return fail(TV_E_ARG, "NULL array");
    if (!tv_cp_fused_supported(g)) return fail(TV_E_ARG, "geometry not supported by the one-sweep path");
    if (!aligned16({t, t_prev, t_next, r})) return fail(TV_E_ARG, "arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    FixPlan fp;
    long long nmax = 0;
    bool empty = false;
    if (int rc = fixup_plan(g, d, t, t_prev, t_next, z_begin, z_count, fp, nmax, empty)) return rc;
    if (empty) {
        HIP_TRY(hipMemsetAsync(rr, 0, sizeof(double), st));
        return 0;
    }
    double* w0 = (double*)ws;
    auto fix = [&]<typename T>() -> int {
        FixupArgsT<T> a{(const T*)t, (const T*)t_prev, (const T*)t_next, (T*)r, nullptr, (T)(-rho), fp.chunk_lo};
        return tvm::fused_fixup<T, ALG_ADMM>(g, d, st, a, fp, w0);
    };
    if (int rc = (g->dtype == TV_F32) ? fix.template operator()<float>() : fix.template operator()<double>()) return rc;
    return reduce_partials(w0, fp.n0 + fp.n1 + fp.n2 + fp.n3, nmax, rr, st);
}

// One-sweep Chambolle-Pock with a data-fidelity operator (tv_fused.h, ALG_CPOP): q <- proj(q + sigma_D D x_in) and
// x_out <- x_in - tau atp - tau D^T q in one pass over q; atp = A^T p is the caller's (solvers.ChambollePockOperator).
int tv_cpop_fused(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, void* q, const void* atp, void* x_out,
                  double sigma_D, double lambda, double tau, int64_t chunk_begin, int64_t chunk_count, double* tvout, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!x_in || !q || !atp || !x_out || !tvout || !ws) return fail(TV_E_ARG, "NULL array");
    if (x_in == x_out || atp == x_out) return fail(TV_E_ARG, "x_out must be a buffer of its own (ping-pong)");
    if (!(lambda > 0.0)) return fail(TV_E_ARG, "lambda must be > 0");
    if (!tv_cp_fused_supported(g)) return fail(TV_E_ARG, "geometry not supported by the one-sweep path");
    if (!aligned16({x_in, x_prev, x_next, q, atp, x_out, d.wv})) return fail(TV_E_ARG, "arrays must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    SweepPlan sp;
    bool empty = false;
    if (int rc = sweep_plan(g, d, x_in, x_prev, x_next, chunk_begin, chunk_count, sp, empty)) return rc;
    if (empty) {
        HIP_TRY(hipMemsetAsync(tvout, 0, sizeof(double), st));
        return 0;
    }
    double* w0 = (double*)ws;
    double* w1 = w0 + sp.nmax + kStage + 16;
    auto sweep = [&]<typename T>() -> int {
        FusedArgsT<T> a{(const T*)x_in, (const T*)x_prev, (const T*)x_next, (T*)q, nullptr, (T*)const_cast<void*>(atp),
                        (T*)x_out, (T)sigma_D, (T)(1.0 / lambda), (T)tau, (T)0, (T)1, w0, w1, 0, (const T*)q};
        return tvm::fused_sweep<T, ALG_CPOP>(g, d, sp.lc, st, a, sp.zc, sp.chunk0, sp.xw, sp.force_win);
    };
    const int rc = (g->dtype == TV_F32) ? sweep.template operator()<float>() : sweep.template operator()<double>();
    if (rc) return rc;
    return reduce_partials(w0, sp.lc.nblocks, sp.nmax, tvout, st);
}

// its fix-up: x_out -= tau (missing adjoint terms); the ALG_ADMM instantiation with the coefficient tau (r += rho ... with rho = -tau)
int tv_cpop_fixup(const tv_geom* g, const void* q, const void* q_prev, const void* q_next, void* x_out, double tau, int64_t z_begin,
                  int64_t z_count, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (ws == nullptr) return fail(TV_E_ARG, "NULL array");
    // the fix-up's reduction (unused here) goes to the last word of the workspace: the pad behind the second partial array, which
    // this call does not use
    double* unused = (double*)ws + 2 * (max_partials(d) + kStage + 16) - 1;
    return tv_admm_fixup(g, q, q_prev, q_next, x_out, -tau, z_begin, z_count, unused, ws, stream);
}

}  // extern "C"
