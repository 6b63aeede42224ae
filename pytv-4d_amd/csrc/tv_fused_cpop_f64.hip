// tv_fused_cpop_f64.hip -- fp64 instantiations of the one-sweep Chambolle-Pock iteration with a user-supplied data-fidelity operator (tv_fused.h,
// ALG_CPOP; round 3).  Only the sweep: its fix-up is the ALG_ADMM one (tv_fused_admm*.hip) with the coefficient tau.
#include "tv_fused_launch.h"
namespace tvm {
template <>
int fused_sweep<double, ALG_CPOP>(const tv_geom* g, const DG& d, const LC& lc, hipStream_t st, const FusedArgsT<double>& a, int zc, int chunk0, bool xw,
                               bool force_win) {
    return fused_sweep_launch<double, ALG_CPOP>(g, d, lc, st, a, zc, chunk0, xw, force_win);
}
}  // namespace tvm
