// tv_subgrad_host.h -- launcher shared by the two translation units that instantiate k_subgrad_one
// (tv_subgrad.hip: MODE 0, G is stored; tv_sgstep.hip: MODE 1, the descent step is applied in the epilogue).
#pragma once
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_subgrad.h"

// every M <= 8 has its own instantiation; more frames run as overlapping time windows of 8 frames (tv_subgrad.h)
inline bool sg_m_ok(int m) { return m >= 1; }

inline int sg_supported(const tv_geom* g) {
    DG d;
    if (make_dg(g, d)) return 0;
    if (g->dtype != TV_F32 || d.nx % 4 != 0 || !sg_m_ok(d.m)) return 0;
    if (d.wv != nullptr) return 0;                                  // weight volume: two-pass kernels
    if (g->scheme == TV_CENTRAL && ((d.za && d.z_two) || (d.ta && d.t_two))) return 0;   // two-point axes: forward stencil
    if ((long long)d.ny * d.nx > (1ll << 30)) return 0;          // 32-bit per-lane byte offsets inside a frame
    if (d.m > SG_TWN && env_int("TV_NO_FUSED_TWIN", 0)) return 0;
    if (env_int("TV_NO_FUSED_SUBGRAD", 0)) return 0;
    return 1;
}

template <typename F> inline int dispatch_sg(int scheme, int m, F&& f) {
#define TV_CASE_G(SC)                                              \
    case SC:                                                       \
        switch (m) {                                               \
            case 0: return f.template operator()<SC, 0>();         \
            case 1: return f.template operator()<SC, 1>();         \
            case 2: return f.template operator()<SC, 2>();         \
            case 3: return f.template operator()<SC, 3>();         \
            case 4: return f.template operator()<SC, 4>();         \
            case 5: return f.template operator()<SC, 5>();         \
            case 6: return f.template operator()<SC, 6>();         \
            case 7: return f.template operator()<SC, 7>();         \
            case 8: return f.template operator()<SC, 8>();         \
        }                                                          \
        break;
    switch (scheme) { TV_CASE_G(0) TV_CASE_G(1) TV_CASE_G(2) TV_CASE_G(3) }
#undef TV_CASE_G
    return fail(TV_E_ARG, "unsupported (scheme, M) for the one-pass sub-gradient");
}

// common argument checks + launch geometry; MODE 1 passes the step arguments, MODE 0 an empty struct
template <int MODE>
inline int sg_launch(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* G, double* tvout, double* fidout,
                     void* ws, void* stream, SgStepArgs sa, const char* who) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (x == nullptr || tvout == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (!sg_supported(g)) return fail(TV_E_ARG, "geometry not supported by the one-pass sub-gradient");
    if (!aligned16({x, x_prev, x_next, G, sa.x0, sa.x_out})) return fail(TV_E_ARG, "arrays must be 16-byte aligned");
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr))) return fail(TV_E_HALO, who);
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    constexpr int NW = 4, UR = 4 * NW - 2, UC = 14;
    const long long tx = (d.nx / 4 + UC - 1) / UC, ty = (d.ny + UR - 1) / UR;
    // planes per z-chunk: every chunk computes two extra planes of norms (and loads four), so chunks are as long as
    // keeping >= ~2048 blocks (4 rounds of 256 CUs x 2) allows; TV_ZCHUNK overrides
    int zc = env_int("TV_ZCHUNK", 0);
    if (zc <= 0) {
        const long long want = (2048 + tx * ty - 1) / (tx * ty);
        zc = (int)(d.nz / (want > 0 ? want : 1));
        if (zc > 32) zc = 32;
        if (zc < 8) zc = 8;
    }
    if (zc > d.nz) zc = d.nz;
    const long long nch = (d.nz + zc - 1) / zc;
    const long long nwin = (d.m > SG_TWN) ? (d.m + SG_TWU - 1) / SG_TWU : 1;      // time windows (M > 8)
    const long long nb = tx * ty * nch * nwin, per_xcd = (nb + 7) / 8;       // XCD-aware logical ids: see the kernel
    const dim3 grid((unsigned)(8 * per_xcd), 1, 1), block(64, NW, 1);
    if (nb > nmax) return fail(TV_E_ARG, "internal: partials exceed the workspace");
    double* w0 = (double*)ws;
    double* w1 = w0 + nmax + kStage + 16;
    sa.part_fid = w1;
    int rc = dispatch_sg(g->scheme, d.m > SG_TWN ? 0 : d.m, [&]<int S, int M>() -> int {
        if constexpr (M == 0)        // M > 8: overlapping windows of 8 frames
            hipLaunchKernelGGL((k_subgrad_one<S, SG_TWN, NW, MODE, true>), grid, block, 0, st, d, make_w<float>(g), (const float*)x,
                               (const float*)x_prev, (const float*)x_next, (float*)G, zc, (int)nch, w0, sa);
        else
            hipLaunchKernelGGL((k_subgrad_one<S, M, NW, MODE>), grid, block, 0, st, d, make_w<float>(g), (const float*)x,
                               (const float*)x_prev, (const float*)x_next, (float*)G, zc, (int)nch, w0, sa);
        HIP_TRY(hipGetLastError());
        return 0;
    });
    if (rc) return rc;
    if (int r2 = reduce_partials(w0, nb, nmax, tvout, st)) return r2;
    if (MODE == 1) return reduce_partials(w1, nb, nmax, fidout, st);
    return 0;
}
