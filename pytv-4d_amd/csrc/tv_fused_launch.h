// tv_fused_launch.h -- launch templates of the one-sweep kernels (tv_fused.h).  One (dtype, ALG) pair per translation unit
// (tv_fused.hip: float CP, tv_fused_f64.hip: double CP, tv_fused_admm.hip / tv_fused_admm_f64.hip: ADMM) so that the four
// sets of instantiations compile next to each other; the extern "C" entry points and the launch geometry live in tv_fused.hip.
#pragma once
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_fused.h"

// launch geometry of the fix-up classes (tv_fused.h, k_cp_fixup)
struct FixPlan {
    dim3 g0, g1, g2, g3;
    long long n0, n1, n2, n3;
    int chunk_lo, zc, zb, zn;
    bool xw;
};

template <typename F> static int dispatch_fused(int scheme, int m, F&& f) {
#define TV_CASE_F(SC)                                              \
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
    switch (scheme) { TV_CASE_F(0) TV_CASE_F(1) TV_CASE_F(2) TV_CASE_F(3) }
#undef TV_CASE_F
    return fail(TV_E_ARG, "unsupported (scheme, M) for the one-sweep path");
}

// the sweep: every M <= 8 has its own instantiation, more frames run as time windows of CP_TWN = 8 frames (M == 0 below).
// ADMM is built with the in-block column hand-off only (XW; TV_FUSED_XW=0 is a debugging switch of the CP sweep).
template <typename T, int ALG>
static int fused_sweep_launch(const tv_geom* g, const DG& d, const LC& lc, hipStream_t st, const FusedArgsT<T>& a, int zc, int chunk0, bool xw,
                              bool force_win) {
    if (ALG != ALG_CP && !xw) return fail(TV_E_ARG, "the ADMM / operator sweeps are built with TV_FUSED_XW=1 only");
    return dispatch_fused(g->scheme, (d.m > CP_TWN || force_win) ? 0 : d.m, [&]<int S, int M>() -> int {
        if constexpr (M == 0) {          // M > 8: windows of 8 frames
            if (xw) hipLaunchKernelGGL((k_cp_fused<S, CP_TWN, true, true, T, ALG>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
            else if constexpr (ALG == ALG_CP)
                hipLaunchKernelGGL((k_cp_fused<S, CP_TWN, false, true, T, ALG>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
        } else if (xw) hipLaunchKernelGGL((k_cp_fused<S, M, true, false, T, ALG>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
        else if constexpr (ALG == ALG_CP)
            hipLaunchKernelGGL((k_cp_fused<S, M, false, false, T, ALG>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

template <typename T, int ALG>
static int fused_fixup_launch(const tv_geom* g, const DG& d, hipStream_t st, const FixupArgsT<T>& a, const FixPlan& p, double* w0) {
    if (ALG != ALG_CP && !p.xw) return fail(TV_E_ARG, "the ADMM / operator sweeps are built with TV_FUSED_XW=1 only");
    const dim3 blk(64, 4, 1);
    auto launch = [&]<int S, bool XW>() -> int {
        if constexpr (ALG == ALG_CP || XW) {
            hipLaunchKernelGGL((k_cp_fixup<S, 0, XW, T, ALG>), p.g0, blk, 0, st, d, make_w<T>(g), a, p.zc, p.zb, p.zn, w0);
            if (d.za) hipLaunchKernelGGL((k_cp_fixup<S, 1, XW, T, ALG>), p.g1, blk, 0, st, d, make_w<T>(g), a, p.zc, p.zb, p.zn, w0 + p.n0);
            hipLaunchKernelGGL((k_cp_fixup<S, 2, XW, T, ALG>), p.g2, blk, 0, st, d, make_w<T>(g), a, p.zc, p.zb, p.zn, w0 + p.n0 + p.n1);
            if (p.n3 > 0)
                hipLaunchKernelGGL((k_cp_fixup<S, 3, XW, T, ALG>), p.g3, blk, 0, st, d, make_w<T>(g), a, p.zc, p.zb, p.zn, w0 + p.n0 + p.n1 + p.n2);
            HIP_TRY(hipGetLastError());
        }
        return 0;
    };
    const bool xw = p.xw;
    switch (g->scheme) {
        case TV_UPWIND: return xw ? launch.template operator()<UPWIND, true>() : launch.template operator()<UPWIND, false>();
        case TV_DOWNWIND: return xw ? launch.template operator()<DOWNWIND, true>() : launch.template operator()<DOWNWIND, false>();
        case TV_CENTRAL: return xw ? launch.template operator()<CENTRAL, true>() : launch.template operator()<CENTRAL, false>();
        default: return xw ? launch.template operator()<HYBRID, true>() : launch.template operator()<HYBRID, false>();
    }
}

// one definition per (dtype, ALG), each in its own translation unit
namespace tvm {
template <typename T, int ALG>
int fused_sweep(const tv_geom* g, const DG& d, const LC& lc, hipStream_t st, const FusedArgsT<T>& a, int zc, int chunk0, bool xw, bool force_win);
template <typename T, int ALG> int fused_fixup(const tv_geom* g, const DG& d, hipStream_t st, const FixupArgsT<T>& a, const FixPlan& p, double* w0);
}  // namespace tvm

#define TV_FUSED_INSTANTIATE(T, ALG)                                                                                                       \
    namespace tvm {                                                                                                                        \
    template <>                                                                                                                            \
    int fused_sweep<T, ALG>(const tv_geom* g, const DG& d, const LC& lc, hipStream_t st, const FusedArgsT<T>& a, int zc, int chunk0, bool xw, \
                            bool force_win) {                                                                                              \
        return fused_sweep_launch<T, ALG>(g, d, lc, st, a, zc, chunk0, xw, force_win);                                                     \
    }                                                                                                                                      \
    template <> int fused_fixup<T, ALG>(const tv_geom* g, const DG& d, hipStream_t st, const FixupArgsT<T>& a, const FixPlan& p, double* w0) { \
        return fused_fixup_launch<T, ALG>(g, d, st, a, p, w0);                                                                             \
    }                                                                                                                                      \
    }
