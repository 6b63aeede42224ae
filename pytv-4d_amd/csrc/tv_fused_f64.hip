// tv_fused_f64.hip -- the fp64 instantiations of the one-sweep Chambolle-Pock iteration (tv_fused.h, round 3): a lane holds
// 2 doubles instead of 4 floats, everything else -- tile in lanes, LDS state, in-block column hand-off, fix-up classes -- is
// the same code.  A translation unit of its own so that it compiles next to tv_fused.hip.
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_fused.h"

template <typename F> static int dispatch_fused64(int scheme, int m, F&& f) {
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

namespace tvm {

int cp_fused_f64(const tv_geom* g, const DG& d, const LC& lc, hipStream_t st, const void* x_in, const void* x_prev, const void* x_next, void* q,
                 const void* x0, void* p, void* x_out, double sigma_D, double lambda, double tau, double sigma_A, int zc, int chunk0, bool xw,
                 bool force_win, double* w0, double* w1) {
    using T = double;
    FusedArgsT<T> a{(const T*)x_in, (const T*)x_prev, (const T*)x_next, (T*)q, (const T*)x0, (T*)p,
                    (T*)x_out, sigma_D, 1.0 / lambda, tau, sigma_A, 1.0 / (1.0 + sigma_A), w0, w1};
    return dispatch_fused64(g->scheme, (d.m > CP_TWN || force_win) ? 0 : d.m, [&]<int S, int M>() -> int {
        if constexpr (M == 0) {          // M > 8: windows of 8 frames
            if (xw) hipLaunchKernelGGL((k_cp_fused<S, CP_TWN, true, true, T>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
            else hipLaunchKernelGGL((k_cp_fused<S, CP_TWN, false, true, T>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
        } else if (xw) hipLaunchKernelGGL((k_cp_fused<S, M, true, false, T>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
        else hipLaunchKernelGGL((k_cp_fused<S, M, false, false, T>), lc.grid, lc.block, 0, st, d, make_w<T>(g), a, zc, chunk0);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int cp_fixup_f64(const tv_geom* g, const DG& d, hipStream_t st, const void* q, const void* q_prev, const void* q_next, void* x_out, const void* x0,
                 double tau, int chunk_lo, int zc, int zb, int zn, bool xw, dim3 g0, dim3 g1, dim3 g2, dim3 g3, long long n0, long long n1,
                 long long n2, long long n3, double* w0) {
    using T = double;
    FixupArgsT<T> a{(const T*)q, (const T*)q_prev, (const T*)q_next, (T*)x_out, (const T*)x0, tau, chunk_lo};
    const dim3 blk(64, 4, 1);
    auto launch = [&]<int S, bool XW>() -> int {
        hipLaunchKernelGGL((k_cp_fixup<S, 0, XW, T>), g0, blk, 0, st, d, make_w<T>(g), a, zc, zb, zn, w0);
        if (d.za) hipLaunchKernelGGL((k_cp_fixup<S, 1, XW, T>), g1, blk, 0, st, d, make_w<T>(g), a, zc, zb, zn, w0 + n0);
        hipLaunchKernelGGL((k_cp_fixup<S, 2, XW, T>), g2, blk, 0, st, d, make_w<T>(g), a, zc, zb, zn, w0 + n0 + n1);
        if (n3 > 0) hipLaunchKernelGGL((k_cp_fixup<S, 3, XW, T>), g3, blk, 0, st, d, make_w<T>(g), a, zc, zb, zn, w0 + n0 + n1 + n2);
        HIP_TRY(hipGetLastError());
        return 0;
    };
    switch (g->scheme) {
        case TV_UPWIND: return xw ? launch.template operator()<UPWIND, true>() : launch.template operator()<UPWIND, false>();
        case TV_DOWNWIND: return xw ? launch.template operator()<DOWNWIND, true>() : launch.template operator()<DOWNWIND, false>();
        case TV_CENTRAL: return xw ? launch.template operator()<CENTRAL, true>() : launch.template operator()<CENTRAL, false>();
        default: return xw ? launch.template operator()<HYBRID, true>() : launch.template operator()<HYBRID, false>();
    }
}

}  // namespace tvm
