// tv_sgstep3.hip -- the MODE 1 (descent step in the epilogue) instantiations of k_subgrad_pair, behind tv_subgrad_step_fused.
#include "tv_subgrad3_host.h"

int sg3_launch_step(const tv_geom* g, const DG& d, const void* x, const void* x_prev, const void* x_next, void* G, double* tvout,
                    double* fidout, void* ws, hipStream_t st, const SgHostArgs& so) {
    return sg3_launch<1>(g, d, x, x_prev, x_next, G, tvout, fidout, ws, st, so);
}
