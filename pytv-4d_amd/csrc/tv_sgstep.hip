// tv_sgstep.hip -- C-ABI of the sub-gradient DESCENT step fused into the one-pass sub-gradient kernel (tv_subgrad.h,
// MODE 1): x_out = x - step ((x - x0) + lambda G(x)) without ever writing G  (README.md:118-124 of the reference).
#include "tv_subgrad_host.h"

extern "C" {

int tv_subgrad_step_fused(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, const void* x0, void* x_out,
                          double step, double lambda, double* tvout, double* fid, void* ws, void* stream) {
    if (x0 == nullptr || x_out == nullptr || fid == nullptr) return fail(TV_E_ARG, "NULL array");
    if (x == x_out) return fail(TV_E_ARG, "x and x_out must be different buffers (ping-pong)");
    SgHostArgs sa{x0, x_out, step, lambda, nullptr};
    return sg_launch<1>(g, x, x_prev, x_next, nullptr, tvout, fid, ws, stream, sa,
                        "tv_subgrad_step_fused on a slab needs two halo planes on each interior side");
}

}  // extern "C"
