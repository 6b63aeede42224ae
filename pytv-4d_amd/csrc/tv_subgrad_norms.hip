// tv_subgrad_norms.hip -- C-ABI of the one-pass TV value + sub-gradient WITH the per-voxel norms |Dx| (tv_subgrad2.h, MODE 2; the
// reference's return_grad_norms=True, pytv/tv_GPU.py:88,135-139).  A translation unit of its own: the MODE 0 / MODE 2 instantiations
// of k_subgrad_col together were the longest compile of the build.
#include "tv_subgrad_host.h"

extern "C" {

int tv_subgrad_fused_norms(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* G, void* norms,
                           double* tvout, void* ws, void* stream) {
    if (G == nullptr || norms == nullptr) return fail(TV_E_ARG, "NULL array");
    SgHostArgs sa{};
    sa.norms = norms;
    return sg_launch<2>(g, x, x_prev, x_next, G, tvout, nullptr, ws, stream, sa,
                        "tv_subgrad_fused_norms on a slab needs two halo planes on each interior side");
}

}  // extern "C"
