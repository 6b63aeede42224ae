// tv_subgrad3.hip -- the MODE 0 (G stored) instantiations of k_subgrad_pair (tv_subgrad3.h), behind tv_subgrad_fused (tv_subgrad.hip).
#include "tv_subgrad3_host.h"

int sg3_launch_g(const tv_geom* g, const DG& d, const void* x, const void* x_prev, const void* x_next, void* G, double* tvout,
                 double* fidout, void* ws, hipStream_t st, const SgHostArgs& so) {
    return sg3_launch<0>(g, d, x, x_prev, x_next, G, tvout, fidout, ws, st, so);
}
