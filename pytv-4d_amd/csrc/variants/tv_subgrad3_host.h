// tv_subgrad3_host.h -- launcher of k_subgrad_pair (tv_subgrad3.h: a lane = 2 rows x 2 columns), one translation unit per MODE
// (tv_subgrad3.hip, tv_subgrad3_norms.hip, tv_sgstep3.hip: each instantiates 4 schemes x 9 frame counts x 3 block variants).
#pragma once
#include "tv_subgrad_host.h"
#include "tv_subgrad3.h"

template <int MODE>
inline int sg3_launch(const tv_geom* g, const DG& d, const void* x, const void* x_prev, const void* x_next, void* G, double* tvout,
                      double* fidout, void* ws, hipStream_t st, const SgHostArgs& so) {
    using T = float;
    constexpr int R = 2, NW = 8, LC = 2, RB = R * NW, UR = RB - 2, UC = 64 * LC - 2 * LC, WC = 64 * LC;
    const long long nmax = max_partials(d);
    const long long tx = (d.nx + UC - 1) / UC, ty = (d.ny + UR - 1) / UR;
    const long long nwin = (d.m > SG2_TWN) ? (d.m + SG2_TWU - 1) / SG2_TWU : 1;
    // planes per z-chunk: as for k_subgrad_col (32 where that still leaves >= 2048 blocks, else 16, shorter for small volumes)
    int zc = env_int("TV_ZCHUNK", 0);
    if (zc <= 0) {
        const long long per_plane_set = tx * ty * nwin;
        zc = (per_plane_set * ((d.nz + 31) / 32) >= 2048) ? 32 : 16;
        while (zc > 8 && per_plane_set * ((d.nz + zc - 1) / zc) < 1024) zc -= 4;
    }
    if (zc > d.nz) zc = d.nz;
    const long long nch = (d.nz + zc - 1) / zc;
    // interior rectangle of the tile grid: every site of the tile, ring lanes and the rows read above / below it included, has
    // both neighbours inside the frame
    SgTiles tm{};
    tm.tx = (int)tx; tm.ty = (int)ty;
    int ix0 = 1, ix1 = (int)((d.nx - 2 - (WC - 1) + LC) / UC), iy0 = 1, iy1 = (int)((d.ny - RB) / UR);
    if (d.nx < WC + 2 || d.ny < RB + 2 || d.mask != nullptr || d.tf != nullptr || d.wv != nullptr || ix1 < ix0 || iy1 < iy0 || env_int("TV_SPARE", 0) == 7) { ix0 = iy0 = 1; ix1 = iy1 = 0; }
    tm.ix0 = ix0; tm.ix1 = ix1; tm.iy0 = iy0; tm.iy1 = iy1;
    tm.nfast = (long long)(ix1 - ix0 + 1) * (iy1 - iy0 + 1);
    tm.nborder = tx * ty - tm.nfast;
    const long long nbf = tm.nfast * nch * nwin, nbb = tm.nborder * nch * nwin, nb = nbf + nbb;
    if (nb > nmax) return fail(TV_E_ARG, "internal: partials exceed the workspace");
    const long long ngrid = (nbb + 7) / 8 * 8 + (nbf + 7) / 8 * 8;
    if (ngrid > 0x7fffffffll) return fail(TV_E_ARG, "volume too large for the one-pass sub-gradient grid");
    const dim3 block(64, NW, 1);
    double* w0 = (double*)ws;
    double* w1 = w0 + nmax + kStage + 16;
    SgArgs2<T> sa{(const T*)so.x0, (T*)so.x_out, (T)so.step, (T)so.lambda, w1, (T*)so.norms};
    int rc = dispatch_sg(g->scheme, d.m > SG2_TWN ? 0 : d.m, [&]<int S, int M>() -> int {
        constexpr int MM = (M == 0) ? SG2_TWN : M;
        constexpr bool TW = (M == 0);
        hipLaunchKernelGGL((k_subgrad_pair<S, MM, MODE, TW>), dim3((unsigned)ngrid), block, 0, st, d, make_w<T>(g),
                           (const T*)x, (const T*)x_prev, (const T*)x_next, (T*)G, zc, (int)nch, w0, sa, tm);
        HIP_TRY(hipGetLastError());
        return 0;
    });
    if (rc) return rc;
    if (int r2 = reduce_partials(w0, nb, nmax, tvout, st)) return r2;
    if (MODE == 1) return reduce_partials(w1, nb, nmax, fidout, st);
    return 0;
}
