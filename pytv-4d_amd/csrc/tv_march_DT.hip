// tv_march_DT.hip -- instantiations + launchers of the plane-marching TRANSPOSED kernels (tv_march.h).
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_march.h"

template <template <typename, int> class EpiT, typename... Args>
static int launch_DT_march(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st,
                           long long* nblocks, Args... args) {
    const int zc = march_zchunk(d);
    const LC lc = march_cfg(d, zc);
    *nblocks = lc.nblocks;
    return dispatch_sm(g->scheme, d.m, [&]<int S, int M>() -> int {
        EpiT<float, 4> epi{args...};
        hipLaunchKernelGGL((k_DT_march<S, M, EpiT<float, 4>>), lc.grid, lc.block, 0, st, d, make_w<float>(g), (const float*)q,
                           (const float*)qp, (const float*)qn, zc, epi);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

namespace tvm {
int DT_store(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb, float* out) {
    return launch_DT_march<StoreDT>(g, d, q, qp, qn, st, nb, out, (double*)nullptr);
}
int DT_axpy(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb,
            float* out, const float* base, float alpha, const float* base2, float beta) {
    return launch_DT_march<AxpyDT>(g, d, q, qp, qn, st, nb, out, base, alpha, (double*)nullptr, base2, beta);
}
int DT_cp_primal(const tv_geom* g, const DG& d, const void* q, const void* qp, const void* qn, hipStream_t st, long long* nb,
                 float* x, const float* x0, float* p, float tau, float sigma_a, float inv_1p_sigma_a, double* partials) {
    return launch_DT_march<CpPrimal>(g, d, q, qp, qn, st, nb, x, x0, p, tau, sigma_a, inv_1p_sigma_a, partials);
}
}  // namespace tvm
