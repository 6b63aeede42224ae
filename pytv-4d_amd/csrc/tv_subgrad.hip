// tv_subgrad.hip -- C-ABI of the one-pass TV value + sub-gradient (tv_subgrad.h), G stored.
#include "tv_subgrad_host.h"

extern "C" {

int tv_subgrad_fused_supported(const tv_geom* g) { return sg_supported(g); }

int tv_subgrad_fused(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* G, double* tvout,
                     void* ws, void* stream) {
    if (G == nullptr) return fail(TV_E_ARG, "NULL array");
    return sg_launch<0>(g, x, x_prev, x_next, G, tvout, nullptr, ws, stream, SgHostArgs{},
                        "tv_subgrad_fused on a slab needs two halo planes on each interior side");
}

}  // extern "C"
