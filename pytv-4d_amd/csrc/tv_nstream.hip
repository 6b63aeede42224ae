// tv_nstream.hip -- instantiations + launcher of the streaming normal operator (tv_nstream.h).
#ifdef TV_NSTREAM_NOCONTRACT
#pragma clang fp contract(off)
#endif
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_nstream.h"

namespace tvm {

bool N_stream_ok(const tv_geom* g, const DG& d, bool vec) {
    // pitched arrays incl. ragged rows (the radius-1 kernel masks the differences that touch pad columns; the central one has
    // per-element masks anyway); `vec` says the rows start on 16-byte boundaries
    if (!vec || d.nx < 64 || d.wv != nullptr) return false;
    if (g->scheme == TV_CENTRAL && ((d.za && d.z_two) || (d.ta && d.t_two))) return false;   // two-point axes: forward stencil
    const long long eb = (g->dtype == TV_F32) ? 4 : 8;
    if (d.s_z * eb >= (1ll << 31)) return false;     // plane descriptors: num_records and the out-of-range offset (tv_fused.h, BUF_OOB)
    if (env_int("TV_NO_MARCH", 0) || env_int("TV_NO_MARCH_NORMAL", 0)) return false;
    // small planes: the z / t neighbours of the one-site kernel stay in L2 (same threshold as the other streaming kernels)
    return (long long)d.s_z * eb >= (long long)env_int("TV_MARCH_MIN_PLANE_KB", 4096) * 1024;
}

int N_stream(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, const void* b, void* out, void* out2,
             double rho, hipStream_t st, long long* nblocks, double* part0, double* part1, const NCheb* cheb) {
    const int V = (g->dtype == TV_F32) ? 4 : 2;
    const long long tx = ((d.nx + V - 1) / V + ST_BCV - 1) / ST_BCV, ty = (d.ny + ST_BR - 1) / ST_BR;
    const long long nwin = (d.m > NS_TWN) ? (d.m + NS_TWN - 1) / NS_TWN : 1;
    const bool ragged = (d.m > NS_TWN) && (d.m % NS_TWN != 0);
    int zc = env_int("TV_NS_ZCHUNK", 0);          // (this kernel family alone; TV_ZCHUNK: every z-chunked kernel)
    if (zc <= 0) zc = env_int("TV_ZCHUNK", 0);
    if (zc <= 0) {
        // >= ~2048 blocks (4 rounds of 256 CUs x 2), counting the time windows.  A chunk reads two planes it does not write and starts with
        // a prologue nothing overlaps: on the configs[4] slab (nz = 32, two windows) the Chebyshev solve takes 6.54 / 6.15 / 5.93 ms in
        // chunks of 8 / 16 / 32 planes (profiles/r5d_nstream_zchunk.txt); until round 5 the rule asked for 4096 blocks without counting
        // the windows and chose 8.
        // The central kernel (two plane lattices per chunk) on ONE window is the exception: 64x8x1024x1024 takes 1.68 ms per Chebyshev
        // step in chunks of 16 and 1.92 in chunks of 32 (profiles/r5d_zchunk_sensitivity.txt) -- it keeps the 4096.
        const long long target = (g->scheme == TV_CENTRAL && nwin == 1) ? 4096 : 2048;
        const long long want = (target + tx * ty * nwin - 1) / (tx * ty * nwin);
        zc = (int)(d.nz / (want > 0 ? want : 1));
        if (zc > 32) zc = 32;
        if (zc < 8) zc = 8;
    }
    if (zc > d.nz) zc = d.nz;
    const long long nch = (d.nz + zc - 1) / zc;
    const long long nb = tx * ty * nch * nwin, per_xcd = (nb + 7) / 8;
    const dim3 grid((unsigned)(8 * per_xcd), 1, 1), block(64, ST_NWX * ST_NWY, 1);
    *nblocks = 8 * per_xcd;
    if (*nblocks > max_partials(d)) return fail(TV_E_ARG, "internal: normal-operator partials exceed the workspace");
    const WT<float> w = make_w<float>(g);
    const NCheb c0{nullptr, nullptr, nullptr, 0.0, 0.0, 0.0};
    const NCheb& c = cheb ? *cheb : c0;
    NormalArgs a{(const float*)x, (const float*)xp, (const float*)xn, (const float*)b, (float*)out, (float*)out2, (float)rho, part0, part1,
                 (const float*)c.y, (const float*)c.add, (const float*)c.ref, (float)c.alpha, (float)c.beta, cheb ? 1 : 0, (float)c.yscale};
    NormalArgsT<double> ad{(const double*)x, (const double*)xp, (const double*)xn, (const double*)b, (double*)out, (double*)out2, rho, part0, part1,
                           (const double*)c.y, (const double*)c.add, (const double*)c.ref, c.alpha, c.beta, cheb ? 1 : 0, c.yscale};
#define TV_NS_LAUNCH1(MM, TW, CH)                                                                                       \
    do {                                                                                                               \
        if (g->scheme == TV_CENTRAL && TW && ragged) {                                                                 \
            if (g->dtype == TV_F64) hipLaunchKernelGGL((k_normal_stream_cen<MM, TW, CH, double, TW>), grid, block, 0, st, d, make_w<double>(g), ad, zc, (int)nch); \
            else hipLaunchKernelGGL((k_normal_stream_cen<MM, TW, CH, float, TW>), grid, block, 0, st, d, w, a, zc, (int)nch); \
        }                                                                                                              \
        else if (g->dtype == TV_F64 && g->scheme == TV_CENTRAL)                                                        \
            hipLaunchKernelGGL((k_normal_stream_cen<MM, TW, CH, double>), grid, block, 0, st, d, make_w<double>(g), ad, zc, (int)nch); \
        else if (g->scheme == TV_CENTRAL) hipLaunchKernelGGL((k_normal_stream_cen<MM, TW, CH, float>), grid, block, 0, st, d, w, a, zc, (int)nch); \
        else if (TW && ragged) {          /* tv_nstream.h, RAGGED: the last window is short */                       \
            if (g->dtype == TV_F64) hipLaunchKernelGGL((k_normal_stream<MM, TW, double, CH, TW>), grid, block, 0, st, d, make_w<double>(g), ad, zc, (int)nch); \
            else hipLaunchKernelGGL((k_normal_stream<MM, TW, float, CH, TW>), grid, block, 0, st, d, w, a, zc, (int)nch); \
        }                                                                                                              \
        else if (g->dtype == TV_F64) hipLaunchKernelGGL((k_normal_stream<MM, TW, double, CH>), grid, block, 0, st, d, make_w<double>(g), ad, zc, (int)nch); \
        else hipLaunchKernelGGL((k_normal_stream<MM, TW, float, CH>), grid, block, 0, st, d, w, a, zc, (int)nch);       \
    } while (0)
    // the Chebyshev epilogue is its own instantiation (tv_nstream.h, ns_epilogue), and so is the first step of a solve (no operand streams)
    const bool cheb_first = cheb != nullptr && b == x && c.y == nullptr && c.add == nullptr && c.ref == nullptr && !env_int("TV_NS_NO_FIRST", 0);
#define TV_NS_LAUNCH(MM, TW)                                                                                            \
    do {                                                                                                               \
        if (cheb_first) TV_NS_LAUNCH1(MM, TW, 2);                                                                      \
        else if (cheb) TV_NS_LAUNCH1(MM, TW, 1);                                                                       \
        else TV_NS_LAUNCH1(MM, TW, 0);                                                                                 \
    } while (0)
    switch (d.m > NS_TWN ? 0 : d.m) {
        case 0: TV_NS_LAUNCH(NS_TWN, true); break;
        case 1: TV_NS_LAUNCH(1, false); break;
        case 2: TV_NS_LAUNCH(2, false); break;
        case 3: TV_NS_LAUNCH(3, false); break;
        case 4: TV_NS_LAUNCH(4, false); break;
        case 5: TV_NS_LAUNCH(5, false); break;
        case 6: TV_NS_LAUNCH(6, false); break;
        case 7: TV_NS_LAUNCH(7, false); break;
        default: TV_NS_LAUNCH(8, false); break;
    }
#undef TV_NS_LAUNCH
#undef TV_NS_LAUNCH1
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace tvm
