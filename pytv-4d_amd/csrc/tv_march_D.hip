// tv_march_D.hip -- instantiations + launchers of the plane-marching FORWARD kernels (tv_march.h).
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_march.h"

template <template <int, typename, int> class EpiT, typename... Args>
static int launch_D_march(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st,
                          long long* nblocks, Args... args) {
    const int zc = march_zchunk(d);
    const LC lc = march_cfg(d, zc);
    *nblocks = lc.nblocks;
    return dispatch_sm(g->scheme, d.m, [&]<int S, int M>() -> int {
        EpiT<S, float, 4> epi{args...};
        hipLaunchKernelGGL((k_D_march<S, M, EpiT<S, float, 4>>), lc.grid, lc.block, 0, st, d, make_w<float>(g), (const float*)x,
                           (const float*)xp, (const float*)xn, zc, epi);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

namespace tvm {
int D_store(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb, float* dout) {
    return launch_D_march<StoreD>(g, d, x, xp, xn, st, nb, dout, (double*)nullptr);
}
int D_cp_dual(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
              float* q, float sigma, float inv_lambda, double* partials) {
    return launch_D_march<CpDual>(g, d, x, xp, xn, st, nb, q, sigma, inv_lambda, partials);
}
int D_admm_zu(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
              float* z, float* u, float thresh, double* partials, int tform) {
    return launch_D_march<AdmmZU>(g, d, x, xp, xn, st, nb, z, u, thresh, partials, tform);
}
int D_norms(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
            float* norms_ext, double* partials, int ghost_lo, int ghost_hi) {
    const int zc = march_zchunk(d);
    LC lc = march_cfg(d, zc);
    const int planes = d.nz + ghost_lo + ghost_hi;
    lc.grid.y = (unsigned)((planes + zc - 1) / zc);
    lc.nblocks = (long long)lc.grid.x * lc.grid.y;
    *nb = lc.nblocks;
    return dispatch_sm(g->scheme, d.m, [&]<int S, int M>() -> int {
        NormEpi<S, float, 4> epi{norms_ext, partials};
        // low-traffic epilogue: M = 8 requests the whole next plane and the halo rows at the top of the z step
        // (PF = 2, 2 waves/SIMD); smaller M take the LIGHT variant (3 waves/SIMD, single-buffered tile), 15-20 %
        // faster than the default one here (measured); M = 16 fits neither
        if constexpr (M == 8)
            hipLaunchKernelGGL((k_D_march<S, M, NormEpi<S, float, 4>, false, 2>), lc.grid, lc.block, 0, st, d, make_w<float>(g),
                               (const float*)x, (const float*)xp, (const float*)xn, zc, epi, 2, -ghost_lo, d.nz + ghost_hi);
        else if constexpr (M < 8)
            hipLaunchKernelGGL((k_D_march<S, M, NormEpi<S, float, 4>, true>), lc.grid, lc.block, 0, st, d, make_w<float>(g),
                               (const float*)x, (const float*)xp, (const float*)xn, zc, epi, 2, -ghost_lo, d.nz + ghost_hi);
        else
            hipLaunchKernelGGL((k_D_march<S, M, NormEpi<S, float, 4>>), lc.grid, lc.block, 0, st, d, make_w<float>(g), (const float*)x,
                               (const float*)xp, (const float*)xn, zc, epi, 2, -ghost_lo, d.nz + ghost_hi);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}
int D_normal_op(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, long long* nb,
                float* out, float rho, double* partials) {
    const int zc = march_zchunk(d);
    const LC lc = march_cfg(d, zc);
    *nb = lc.nblocks;
    const WT<float> w = make_w<float>(g);
    NormalEpi<HYBRID, float, 4> epi{out, rho, w.wz, w.wt, w.sf, d.mask, partials};
    return dispatch_sm(TV_HYBRID, d.m, [&]<int S, int M>() -> int {
        if constexpr (S == HYBRID && M == 8) {
            // whole next plane + halo rows requested at the top of the z step: 1.61 -> 1.28 ms on 64x8x1024x1024
            hipLaunchKernelGGL((k_D_march<HYBRID, M, NormalEpi<HYBRID, float, 4>, false, 2>), lc.grid, lc.block, 0, st, d, w,
                               (const float*)x, (const float*)xp, (const float*)xn, zc, epi, 2, 0, -1);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        if constexpr (S == HYBRID && M <= 8) {
            hipLaunchKernelGGL((k_D_march<HYBRID, M, NormalEpi<HYBRID, float, 4>, true>), lc.grid, lc.block, 0, st, d, w,
                               (const float*)x, (const float*)xp, (const float*)xn, zc, epi, 2, 0, -1);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        return fail(TV_E_ARG, "marching normal operator: M > 8");
    });
}
}  // namespace tvm
