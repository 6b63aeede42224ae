// tv_subgrad_host.h -- launcher shared by the translation units that instantiate k_subgrad_col
// (tv_subgrad.hip: MODE 0, G is stored; tv_sgstep.hip: MODE 1, the descent step is applied in the epilogue).
#pragma once
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_subgrad2.h"

// every M <= 8 has its own instantiation; more frames run as overlapping time windows of 8 frames (tv_subgrad.h)
inline bool sg_m_ok(int m) { return m >= 1; }

inline int sg_supported(const tv_geom* g) {
    DG d;
    if (make_dg(g, d, true)) return 0;
    if (!sg_m_ok(d.m)) return 0;
    // the round-3 kernel holds ONE column per lane (4- / 8-byte buffer loads): any Nx, any element-aligned pointer (late round 3;
    // before, ragged Nx took the two-pass path at 0.11 - 0.17 of the roofline).  The round-1 kernel (TV_SG_KERNEL=1, frames of
    // 2^31 bytes and more) is fp32 with 16-byte lanes.
    const bool k2 = env_int("TV_SG_KERNEL", 2) != 1 && d.s_t * (g->dtype == TV_F32 ? 4 : 8) < (1ll << 31);
    if (!k2) return 0;              // frames of 2^31 bytes and more: the two-pass path (the round-1 kernel that took them is a variant build now)
    if (g->scheme == TV_CENTRAL && ((d.za && d.z_two) || (d.ta && d.t_two))) return 0;   // two-point axes: forward stencil
    if (d.s_t > (1ll << 30)) return 0;                           // 32-bit per-lane byte offsets inside a frame
#if TV_SG2_PLANE_DESC
    // one buffer descriptor per PLANE (experiment): the plane's M frames must stay below 2^31 bytes (num_records is 32-bit and the "not my lane" offset 2^31
    // must lie outside it); larger planes take the two-pass kernels
    if (d.m <= SG2_TWN && d.s_t * (long long)(g->dtype == TV_F32 ? 4 : 8) * d.m >= (1ll << 31)) return 0;
#endif
    if (d.m > SG2_TWN && env_int("TV_NO_FUSED_TWIN", 0)) return 0;
    if (env_int("TV_NO_FUSED_SUBGRAD", 0)) return 0;
    return 1;
}

template <typename F> inline int dispatch_sg(int scheme, int m, F&& f) {
#define TV_CASE_G(SC)                                              \
    case SC:                                                       \
        switch (m) {                                               \
            case 0: return f.template operator()<SC, 0>();         \
            case 1: return f.template operator()<SC, 1>();         \
            case 2: return f.template operator()<SC, 2>();         \
            case 3: return f.template operator()<SC, 3>();         \
            case 4: return f.template operator()<SC, 4>();         \
            case 5: return f.template operator()<SC, 5>();         \
            case 6: return f.template operator()<SC, 6>();         \
            case 7: return f.template operator()<SC, 7>();         \
            case 8: return f.template operator()<SC, 8>();         \
        }                                                          \
        break;
    switch (scheme) { TV_CASE_G(0) TV_CASE_G(1) TV_CASE_G(2) TV_CASE_G(3) }
#undef TV_CASE_G
    return fail(TV_E_ARG, "unsupported (scheme, M) for the one-pass sub-gradient");
}

// what the C entry points hand to the launcher (dtype-neutral)
struct SgHostArgs {
    const void* x0 = nullptr;
    void* x_out = nullptr;
    double step = 0.0, lambda = 0.0;
    void* norms = nullptr;
};

// ---- round-3 kernel (tv_subgrad2.h): a lane = R rows x 1 column -----------------------------------------------------------
// fp32: R = 4 rows per lane, 4 waves per block (tile 16 rows x 64 columns, 14 x 60 / 62 stored), row hand-off through LDS.
// fp64: R = 2 (the same registers), tile 8 x 64 (6 x 60 / 62 stored), neighbour rows from memory (XLD) so that two blocks still
// fit a CU's LDS.  TV_SG_KERNEL=1 selects the round-1 kernel (k_subgrad_one, fp32 only) for A/B.
template <typename T, int MODE>
inline int sg2_launch(const tv_geom* g, const DG& d, const void* x, const void* x_prev, const void* x_next, void* G, double* tvout,
                      double* fidout, void* ws, hipStream_t st, const SgHostArgs& so) {
    constexpr bool F64 = sizeof(T) == 8;
#ifndef TV_SG2_R32
#define TV_SG2_R32 4           // fp32: rows per lane.  EXPERIMENT -DTV_SG2_R32=2 -DTV_SG2_NW32=8 -DTV_SG2_XLD32=1: 16 waves of 2-row strips per block (4 waves per SIMD)
#endif
#ifndef TV_SG2_NW32
#define TV_SG2_NW32 4
#endif
    constexpr int R = F64 ? 2 : TV_SG2_R32;
#ifndef TV_SG2_NWX
#define TV_SG2_NWX 2
#endif
#ifndef TV_SG2_XLD32
#define TV_SG2_XLD32 0
#endif
#ifndef TV_SG2_AL
#define TV_SG2_AL 1            // round 5: aligned 64-column tiles with a ring slot (tv_subgrad2.h, AL): 1 = instantiated for the one-sided schemes (where they win), 2 = for all four (experiments), 0 = not at all
#endif
    constexpr bool XLD = F64 || TV_SG2_XLD32;
    // ALIGNED tiles (round 5): fp32, every frame in registers (M <= 8), no per-voxel weight volume (the ring slot does not read one),
    // a plane of M frames below 2^31 bytes (one buffer descriptor per plane); TV_SG_ALIGNED=0 switches back at run time (A/B).
    // A block is then 8 waves STACKED (32 rows x 64 columns, 30 x 64 stored); otherwise the round-3 tile: 4 x 2 waves, 16 x 128.
    // Used for UPWIND / DOWNWIND: same-box A/B (profiles/r5_subgrad_aligned_ab.txt), descent loop at 256x8x1024x1024 148 -> 166 it/s, the
    // norms variant at 64 planes 1.61 - 1.79 -> 1.49 ms, tv_subgrad_fused unchanged (1.17 - 1.26 ms).  NOT for hybrid / central: their
    // instantiations are at the register limit already (245 - 248 VGPRs), the ring slot's ~30 registers spill (128 / 28 B per lane) and
    // cost more than the whole lines gain (hybrid 1.61 -> 1.95 ms, loop 142 -> 99 it/s).  TV_SG_ALIGNED=0: off; =2: every scheme a
    // -DTV_SG2_AL=2 build instantiates.
    const int al_opt = env_int("TV_SG_ALIGNED", 1);
    const bool al_scheme = (g->scheme == TV_UPWIND || g->scheme == TV_DOWNWIND) ? (al_opt != 0) : (TV_SG2_AL == 2 && al_opt == 2);
    // fp64 (R = 2 rows of doubles per lane, neighbour rows from memory): the same aligned tile, 8 waves stacked = 16 rows x 64 columns, 14 x 64
    // stored in whole lines (64 doubles = four 128-byte lines per row) instead of 8 x 64 with 6 x 60 / 62 stored
    // (fp64 A/B, profiles/r5_f64_aligned_ab.txt: the norms variant gains 3 - 11 %, tv_subgrad_fused itself loses 2 - 5 %: MODE 1 / 2 only)
    const bool al = TV_SG2_AL && al_scheme && d.m <= SG2_TWN && d.wv == nullptr && d.s_t * (long long)sizeof(T) * d.m < (1ll << 31) &&
                    (!F64 || MODE != 0);
    const int NW = al ? 8 : (F64 ? 4 : TV_SG2_NW32), NWX = (F64 || al) ? 1 : TV_SG2_NWX;
    const long long nmax = max_partials(d);
    const bool halo = (g->scheme == TV_HYBRID || g->scheme == TV_CENTRAL);
    const int UR = R * NW - 2, UC = al ? 64 : (halo ? 60 : 62);
    const long long tx = (d.nx + UC - 1) / UC, ty = (d.ny + UR - 1) / UR;
    const long long nwin = (d.m > SG2_TWN) ? (d.m + SG2_TWU - 1) / SG2_TWU : 1;
    // planes per z-chunk (TV_ZCHUNK overrides): 16.  A chunk computes two extra planes of norms, so longer chunks waste less
    // -- and still measured slower on the north-star volume (256 planes: 8.2 / 8.3 / 8.8 / 8.9 ms per descent step with 16 /
    // 32 / 64 / 86 planes per chunk, profiles/r3_sg2_zchunk.txt): many short blocks balance the 512 block slots better than
    // few long ones.  Shorter only when the volume would not give 2048 blocks otherwise.
    int zc = env_int("TV_ZCHUNK", 0);
    if (zc <= 0) {
        // 32 planes where that still leaves >= 2048 blocks (with two wave columns per block the longer chunk wins: 256 planes,
        // hybrid 8.58 / 7.97 / 8.13 ms with 16 / 32 / 64), else 16, shorter only when the volume would not give 1024 blocks
        const long long per_plane_set = ((tx + NWX - 1) / NWX) * ty * nwin;
        zc = (per_plane_set * ((d.nz + 31) / 32) >= 2048) ? 32 : 16;
        while (zc > 8 && per_plane_set * ((d.nz + zc - 1) / zc) < 1024) zc -= 4;
        // volumes of a few planes of small frames (the reference's own shapes: pytv/tests.py:48 N = 100, README.md:76-79
        // rand(20, 4, 100, 100)): below one block per CU the chunk overlap is cheaper than idle CUs -- 20x4x100x100 hybrid 0.063 -> 0.038 ms
        // per descent step with 2-plane chunks, 256x4x100x100 (256 blocks at 8 planes) best as it is: profiles/r5_small_frames.txt
        while (zc > 2 && per_plane_set * ((d.nz + zc - 1) / zc) < 256) zc -= 2;
    }
    if (zc > d.nz) zc = d.nz;
    const long long nch = (d.nz + zc - 1) / zc;
    // interior rectangle of the tile grid (tv_subgrad2.h, SgTiles): tiles with every site, ring included, strictly inside the frame
    const int RING = halo ? 2 : 1, RB = R * NW;
    // tiles of the launch = BLOCK tiles: NWX wave tiles side by side (the last block of a row may hold a wave tile beyond the frame:
    // all its lanes are out of range).  A block runs the variant without border multipliers iff all its wave tiles are interior.
    const long long txb = (tx + NWX - 1) / NWX;
    SgTiles tm{};
    tm.tx = (int)txb; tm.ty = (int)ty;
    int ix0 = 1, ix1 = (int)((d.nx - 2 - 63 + RING) / UC), iy0 = 1, iy1 = (int)((d.ny - RB) / UR);
    if (al) ix1 = (d.nx - 1) / 64 - 1;          // lanes 0 .. 63 of the tile have both column neighbours: c0 >= 1 and c0 + 64 <= nx - 1
    if (d.nx < 66 || d.ny < RB + 2 || d.mask != nullptr || d.tf != nullptr || d.wv != nullptr || ix1 < ix0 || iy1 < iy0 || env_int("TV_SPARE", 0) == 7) { ix0 = iy0 = 1; ix1 = iy1 = 0; }   // (TV_SPARE=7: experiment, every block generic)
    if (ix1 >= ix0) {            // wave-tile columns [ix0, ix1] -> block-tile columns whose NWX wave tiles all lie inside
        const int b0 = (ix0 + NWX - 1) / NWX, b1 = (ix1 + 1) / NWX - 1;
        ix0 = b0; ix1 = b1;
        if (ix1 < ix0) { ix0 = iy0 = 1; ix1 = iy1 = 0; }
    }
    tm.ix0 = ix0; tm.ix1 = ix1; tm.iy0 = iy0; tm.iy1 = iy1;
    tm.nfast = (long long)(ix1 - ix0 + 1) * (iy1 - iy0 + 1);
    tm.nborder = txb * ty - tm.nfast;
    const long long nbf = tm.nfast * nch * nwin, nbb = tm.nborder * nch * nwin, nb = nbf + nbb;
    if (nb > nmax) return fail(TV_E_ARG, "internal: partials exceed the workspace");
    const long long ngrid = (nbb + 7) / 8 * 8 + (nbf + 7) / 8 * 8;
    if (ngrid > 0x7fffffffll) return fail(TV_E_ARG, "volume too large for the one-pass sub-gradient grid");
    const dim3 block(64, NW * NWX, 1);
    double* w0 = (double*)ws;
    double* w1 = w0 + nmax + kStage + 16;
    SgArgs2<T> sa{(const T*)so.x0, (T*)so.x_out, (T)so.step, (T)so.lambda, w1, (T*)so.norms};
    int rc = dispatch_sg(g->scheme, d.m > SG2_TWN ? 0 : d.m, [&]<int S, int M>() -> int {
        constexpr int MM = (M == 0) ? SG2_TWN : M;
        constexpr bool TW = (M == 0);
        if constexpr (!TW && (!F64 || MODE != 0) && (TV_SG2_AL == 2 || (TV_SG2_AL == 1 && (S == UPWIND || S == DOWNWIND)))) {
            if (al) {
                hipLaunchKernelGGL((k_subgrad_col<S, T, MM, R, 8, MODE, false, XLD, 1, true>), dim3((unsigned)ngrid), block, 0, st, d, make_w<T>(g),
                                   (const T*)x, (const T*)x_prev, (const T*)x_next, (T*)G, zc, (int)nch, w0, sa, tm);
                HIP_TRY(hipGetLastError());
                return 0;
            }
        }
        hipLaunchKernelGGL((k_subgrad_col<S, T, MM, R, (F64 ? 4 : TV_SG2_NW32), MODE, TW, XLD, (F64 ? 1 : TV_SG2_NWX)>), dim3((unsigned)ngrid), block, 0, st, d,
                           make_w<T>(g), (const T*)x, (const T*)x_prev, (const T*)x_next, (T*)G, zc, (int)nch, w0, sa, tm);
        HIP_TRY(hipGetLastError());
        return 0;
    });
    if (rc) return rc;
    if (int r2 = reduce_partials(w0, nb, nmax, tvout, st)) return r2;
    if (MODE == 1) return reduce_partials(w1, nb, nmax, fidout, st);
    return 0;
}


// common argument checks + launch geometry; MODE 1 passes the step arguments, MODE 0 an empty struct
template <int MODE>
inline int sg_launch(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* G, double* tvout, double* fidout,
                     void* ws, void* stream, const SgHostArgs& ha, const char* who) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;      // pitched arrays: the round-3 kernel (one column per lane: pads are never touched)
    if (x == nullptr || tvout == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (!sg_supported(g)) return fail(TV_E_ARG, "geometry not supported by the one-pass sub-gradient");
    const bool vec16 = (d.nx % 4 == 0) && aligned16({x, x_prev, x_next, G, ha.x0, ha.x_out, ha.norms});      // what the round-1 kernel needs
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr))) return fail(TV_E_HALO, who);
    hipStream_t st = (hipStream_t)stream;
    // round-3 kernel unless switched off; its descent step divides by step * lambda (tv_subgrad2.h): tiny or zero products
    // take the round-1 kernel; frames must stay below 2^31 bytes for its buffer addressing
    const bool step_ok = (MODE != 1 || ha.step * ha.lambda >= 1e-6);
    if (g->dtype == TV_F64) {
        if (!step_ok) return fail(TV_E_ARG, "the fp64 one-pass descent step needs step * lambda >= 1e-6: use tv_subgrad + tv_subgrad_step");
        return sg2_launch<double, MODE>(g, d, x, x_prev, x_next, G, tvout, fidout, ws, st, ha);
    }
    if (env_int("TV_SG_KERNEL", 2) != 1 && d.s_t * 4 < (1ll << 31) && step_ok)
        return sg2_launch<float, MODE>(g, d, x, x_prev, x_next, G, tvout, fidout, ws, st, ha);
    // the product library holds ONE generation of the one-pass kernel (round-4 verdict, item 7): what it cannot take goes to the two-pass
    // entry points (tv_subgrad + tv_subgrad_step), as solvers.SubgradientDescent does on its own
    (void)vec16;
    return fail(TV_E_ARG, "the one-pass sub-gradient kernel needs step * lambda >= 1e-6 and frames below 2^31 bytes: use tv_subgrad + tv_subgrad_step");
}
