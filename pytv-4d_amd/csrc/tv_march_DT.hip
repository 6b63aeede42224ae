// tv_march_DT.hip -- instantiations + launchers of the plane-marching TRANSPOSED kernels (tv_march.h).
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_march.h"

template <template <typename, int> class EpiT, typename T, typename... Args>
static int launch_DT_march(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st,
                           long long* nblocks, Args... args) {
    constexpr int V = 16 / (int)sizeof(T);
    const int zc = march_zchunk(d);
    const LC lc = march_cfg(d, zc);
    *nblocks = lc.nblocks;
    return dispatch_sm(g->scheme, d.m, [&]<int S, int M>() -> int {
        EpiT<T, V> epi{args...};
        hipLaunchKernelGGL((k_DT_march<S, M, EpiT<T, V>, T>), lc.grid, lc.block, 0, st, d, make_w<T>(g), (const T*)q,
                           (const T*)qp, (const T*)qn, zc, epi);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

namespace tvm {
int DT_store(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb, void* out) {
    if (g->dtype == TV_F64)        // as AxpyDT with no base and alpha = 1 (tv_march.h: the StoreDT instantiation for double takes 256 VGPRs or spills)
        return launch_DT_march<AxpyDT, double>(g, d, q, qp, qn, st, nb, (double*)out, (const double*)nullptr, 1.0, (double*)nullptr, (const double*)nullptr, 0.0);
    return launch_DT_march<StoreDT, float>(g, d, q, qp, qn, st, nb, (float*)out, (double*)nullptr);
}
int DT_axpy(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb,
            void* out, const void* base, double alpha, const void* base2, double beta) {
    if (g->dtype == TV_F64)
        return launch_DT_march<AxpyDT, double>(g, d, q, qp, qn, st, nb, (double*)out, (const double*)base, alpha, (double*)nullptr, (const double*)base2, beta);
    return launch_DT_march<AxpyDT, float>(g, d, q, qp, qn, st, nb, (float*)out, (const float*)base, (float)alpha, (double*)nullptr, (const float*)base2, (float)beta);
}
int DT_cp_primal(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb,
                 float* x, const float* x0, float* p, float tau, float sigma_a, float inv_1p_sigma_a, double* partials) {
    return launch_DT_march<CpPrimal, float>(g, d, q, qp, qn, st, nb, x, x0, p, tau, sigma_a, inv_1p_sigma_a, partials);
}
}  // namespace tvm
