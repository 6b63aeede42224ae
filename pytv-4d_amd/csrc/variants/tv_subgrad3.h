// tv_subgrad3.h -- ONE-PASS TV value + sub-gradient (pytv/tv_GPU.py:47-375 of the reference): the round-4 EXPERIMENT on the geometry of the
// round-3 kernel.  OPT-IN (TV_SG_KERNEL=3), parity-green, NOT faster than k_subgrad_col -- kept as the measured answer to "build the
// tile with long row segments" (VERDICT round 3, item 5); profiles/r4_sgpattern.txt holds the numbers.
//
// Same mathematics and the same machinery as tv_subgrad2.h (scatter / flux form, z marching with all M frames in registers, raw
// buffer accesses whose offsets the hardware range-checks, LDS hand-off of the row products with one LDS-only barrier per plane,
// three block variants FAST / column-border / generic) on a different tile.  Why: tools/sgpattern.hip (the kernel's memory pattern
// without its arithmetic, geometry as a parameter) says that what the round-3 tile pays for is its STORES.  A lane of k_subgrad_col
// is one column, a wave stores 60 of its 64 columns: 240 bytes that start anywhere -- three 128-byte lines touched, none written
// whole, and a partly written line costs about what 2.7 whole ones do (the round-3 kernel with its stores moved to line-aligned
// positions, results wrong: descent loop 134 -> 155 it/s, tv_subgrad_fused_norms 2.12 -> 1.91 ms; cache-policy bits, deeper load
// rings and load alignment do not matter).  Here
//
//   a lane = 2 ROWS x 2 COLUMNS (the same four sites, the same registers), a wave = 2 rows x 128 columns, 8 waves stacked in y:
//   the 16-row x 128-column tile of before, but ONE wave owns a whole row segment -- 124 contiguous columns = 496 bytes, three
//   whole lines and one or two partial ones per row, 8-byte accesses;
//   * the ring costs one LANE on either side (its outer column only supplies x, its inner one a norm): 124 of 128 columns stored
//     instead of 120, for every scheme;
//   * column neighbours: one in the lane, one a DPP move -- half the DPP traffic per site;
//   * every row of a strip is a strip end now: the row above / below comes from memory (two more 8-byte loads per frame, L2
//     hits: the XLD form of tv_subgrad2.h -- there is no LDS left for an x hand-off), the row PRODUCTS go through LDS as before.
// Measured (64x8x1024x1024 / descent loop on 256x8x1024x1024, interleaved with the round-3 kernel): hybrid 1.54 = 1.54 ms, upwind
// 1.33 vs 1.24, loop hybrid 131 vs 143 it/s, upwind 146 - 149 vs 141 - 145.  The two remaining partial lines per row still cost 10 %
// (the same experiment on THIS kernel: loop upwind 146 -> 160), the neighbour-row loads another 10 % on hybrid -- and whole-line
// stores need a tile stride of a multiple of 32 columns, i.e. 96 of a wave's 128 columns: a quarter of the lanes, more than the
// stores cost.  Within the register budget of M = 8 frames (2048 sites per CU) no geometry has both.
// fp32, even Nx, 8-byte aligned arrays (sg3_ok in tv_subgrad_host.h).
#pragma once
#include "tv_subgrad2.h"

#ifndef TV_SG3_STALIGN
#define TV_SG3_STALIGN 0       // 1: EXPERIMENT -- stores at line-aligned positions (results wrong)
#endif
#ifndef TV_SG3_NOXLD
#define TV_SG3_NOXLD 0         // 1: EXPERIMENT -- no loads of the rows above / below the strip (results wrong): what they cost
#endif

namespace tv {

typedef float sg3_v2f __attribute__((ext_vector_type(2)));
template <int R> struct Pair { float v[R][2]; };

__device__ __forceinline__ sg3_v2f sg3_ld2(Rsrc r, unsigned off) {
    const sg2_v2i a = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    sg3_v2f o;
    o.x = __int_as_float(a.x);
    o.y = __int_as_float(a.y);
    return o;
}
__device__ __forceinline__ void sg3_st2(Rsrc r, unsigned off, float a, float b) {
    sg2_v2i w;
    w.x = __float_as_int(a);
    w.y = __float_as_int(b);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)off, 0, 0);
}
__device__ __forceinline__ sg3_v2f sg3_mk(float a, float b) {
    sg3_v2f o;
    o.x = a;
    o.y = b;
    return o;
}

// MODE as in tv_subgrad2.h: 0 G stored; 1 the descent step (README.md:118-124); 2 G and the per-voxel norms.
template <int S, int M, int MODE, bool TWIN>
struct SgPair {
    using T = float;
    static constexpr int R = 2, LC = 2, NW = 8;
    static constexpr bool CEN = (S == CENTRAL);
    static constexpr bool UP = (S == UPWIND || S == HYBRID), DN = (S == DOWNWIND || S == HYBRID);
    static constexpr bool HALO = (S == HYBRID || CEN);     // the norm of a ring row looks at the row outside the tile
    // x is valid on every column of the wave, 1 / |Dx| on all but the outermost one (hybrid / central; one-sided schemes: on one
    // side all), G from the third column on: the tile owns lanes 1 .. 62
    static constexpr int RB = R * NW, UR = RB - 2, UC = 64 * LC - 2 * LC;
    static constexpr int H = (SG2_D < M) ? SG2_D : M;
    static constexpr bool HEADS = (M % SG2_D != 0);
    static constexpr int ROWS = 2 * NW + 1, ZR = 2 * NW;
    using C = Pair<R>;

    struct Shared {
        // [parity][frame][row][lane] (8 bytes: the lane's two columns); row 2 w = product handed UP by wave w's first row, 2 w + 1 =
        // handed DOWN by its last row, row ZR = zeros (what the block's first / last wave reads: no select)
        sg3_v2f ye[2][M][ROWS][64];
        double sm[16];
    };

    template <int BK>
    static __device__ __forceinline__ void run(const DG& g, const WT<T>& w, const T* __restrict__ x, const T* __restrict__ xp,
                                               const T* __restrict__ xn, T* __restrict__ G, int zchunk, int chunk, int tile_x, int tile_y,
                                               int win, long long lid, double* __restrict__ partials, const SgArgs2<T>& sa, Shared& sh) {
        constexpr bool FAST = (BK == 0);         // no border multipliers at all
        constexpr bool ROWS_IN = (BK <= 1);      // every row this thread touches exists and has both neighbours
        constexpr bool TFAST = (BK <= 1);        // uniform time factor: the weighted time difference is carried from frame to frame
        const int lane = (int)threadIdx.x, wv = __builtin_amdgcn_readfirstlane((int)threadIdx.y);
        auto& ye = sh.ye;
        const int Mg = TWIN ? g.m : M;
        const int t0 = TWIN ? win * SG2_TWU - 1 : 0;
        auto fvalid = [&](int t) { return !TWIN || (t0 + t >= 0 && t0 + t < Mg); };
        auto fstore = [&](int t) { return !TWIN || (t0 + t >= win * SG2_TWU && t0 + t < win * SG2_TWU + SG2_TWU && t0 + t < Mg); };
        auto foff_t = [&](int t) { return (long long)(t0 + t) * g.s_t; };     // uniform
        const int fbytes = (int)(g.s_t * (long long)sizeof(T));
        const int cx = tile_x * UC - LC + LC * lane;           // the lane's first column (even: Nx is even, a lane is inside the frame or outside)
        const int yb = tile_y * UR - 1 + wv * R;               // first row of this wave's strip
        const bool in_x = FAST || (cx >= 0 && cx + 1 < g.nx);
        unsigned roff[R], soff[R];
        C mi, mfr, mbr, mfc, mbc, mft, cm;
        const bool lane_ok = in_x && lane >= 1 && lane <= 62;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int y = yb + i;
            const bool in = in_x && (ROWS_IN || (y >= 0 && y < g.ny));
            const bool own = in && lane_ok && !(wv == 0 && i == 0) && !(wv == NW - 1 && i == R - 1);
            const unsigned off = (unsigned)(((long long)y * g.rp + cx) * (long long)sizeof(T));
            roff[i] = in ? off : SG2_OOB;
            soff[i] = own ? off : SG2_OOB;
#if TV_SG3_STALIGN        // EXPERIMENT (wrong results): every lane stores, at 128-column-aligned positions
            {
                const int ca = tile_x * 128 + LC * lane;
                soff[i] = (y >= 0 && y < g.ny && ca + 1 < g.rp) ? (unsigned)(((long long)y * g.rp + ca) * (long long)sizeof(T)) : SG2_OOB;
            }
#endif
            const bool hn = in && (ROWS_IN || y + 1 < g.ny), hp = in && (ROWS_IN || y > 0);
#pragma unroll
            for (int j = 0; j < LC; ++j) {
                const int cj = cx + j;
                const bool cn = in && (cj + 1 < g.nx), cp = in && (cj > 0);
                cm.v[i][j] = own ? T(1) : T(0);
                mi.v[i][j] = in ? T(1) : T(0);
                mfr.v[i][j] = (CEN ? (hn && hp) : hn) ? T(1) : T(0);
                mbr.v[i][j] = (CEN ? (hn && hp) : hp) ? T(1) : T(0);
                mfc.v[i][j] = (CEN ? (cn && cp) : cn) ? T(1) : T(0);
                mbc.v[i][j] = (CEN ? (cn && cp) : cp) ? T(1) : T(0);
                mft.v[i][j] = T(0);
                if (!TFAST && g.ta && in) mft.v[i][j] = w.wt * mask_factor1<T>(g, w.sf, y, cj);
            }
        }
        const int zs = chunk * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        const T s = (S == HYBRID) ? Consts<T>::inv_sqrt2() : (CEN ? T(0.5) : T(1));
        const T thr = tiny_sumsq<T>() / (s * s);               // zero-gradient rule on the UNSCALED sum of squares: s^2 ss < tiny
        const T a_step = (MODE == 1) ? sa.step * sa.lambda * s : T(0);
        const T kap = (MODE == 1) ? -(T(1) - sa.step) / a_step : T(0);
        const T wt_u = g.ta ? w.wt : T(0);                     // FAST: uniform time weight
        const T wz_u = g.za ? w.wz : T(0);
        // the rows just above / below the strip, from memory (where the scheme looks that way and the row exists; the block's
        // first / last wave only for the schemes whose ring norm looks outwards)
        unsigned uoff = SG2_OOB, doff = SG2_OOB;
        if (in_x && !TV_SG3_NOXLD) {
            if ((DN || CEN) && yb > 0 && (wv > 0 || HALO)) uoff = (unsigned)(((long long)(yb - 1) * g.rp + cx) * (long long)sizeof(T));
            if ((UP || CEN) && yb + R < g.ny && (wv < NW - 1 || HALO)) doff = (unsigned)(((long long)(yb + R) * g.rp + cx) * (long long)sizeof(T));
        }
        const int r_own = 2 * wv, r_up = (wv > 0) ? 2 * (wv - 1) + 1 : ZR, r_dn = (wv < NW - 1) ? 2 * (wv + 1) : ZR;
        const sg3_v2f zero2 = sg3_mk(T(0), T(0));
        if (wv == 0) {
#pragma unroll
            for (int t = 0; t < M; ++t) ye[0][t][ZR][lane] = ye[1][t][ZR][lane] = zero2;
        }
#pragma unroll
        for (int t = 0; t < M; ++t)        // the first step reads the "previous step's" products for a store that is dropped: keep them finite
            ye[0][t][r_own][lane] = ye[0][t][r_own + 1][lane] = ye[1][t][r_own][lane] = ye[1][t][r_own + 1][lane] = zero2;
        double acc = 0.0, acc_fid = 0.0;

        auto frame = [&](const T* plane, int t) { return sg2_rsrc<T>(plane + foff_t(t), plane != nullptr && fvalid(t), fbytes); };
        auto load_rows = [&](Rsrc r, C& o) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const sg3_v2f v = sg3_ld2(r, roff[i]);
                o.v[i][0] = v.x;
                o.v[i][1] = v.y;
            }
        };

        // per-frame state carried from plane to plane (tv_subgrad2.h): x(z); the accumulators of G(z-1) and G(z); central: x(z-1);
        // downwind / hybrid: the weighted forward z difference of plane z-1 = the backward difference of plane z
        constexpr bool PZ = DN || CEN;
        C Cc[M], Pz[PZ ? M : 1], Gp[M], Gc[M], Nq[SG2_D], Nh[HEADS ? H : 1];
        const int z_lo = g.za ? zs - 1 : zs;
        {
            const T* pp = g.za ? zplane<T>(g, x, xp, xn, 2, z_lo - 1) : nullptr;
            const T* pc = zplane<T>(g, x, xp, xn, 2, z_lo);
            const int gz0 = g.z0 + z_lo;
            const bool zp0 = (gz0 > 0) && (gz0 < g.nzg) && g.za;       // planes z_lo - 1 and z_lo both exist
#pragma unroll
            for (int t = 0; t < M; ++t) {
                load_rows(frame(pc, t), Cc[t]);
                if (PZ) {
                    C pv;
                    load_rows(frame(pp, t), pv);
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int j = 0; j < LC; ++j) {
                            if (CEN) Pz[t].v[i][j] = pv.v[i][j];
                            else Pz[t].v[i][j] = (zp0 ? wz_u : T(0)) * (Cc[t].v[i][j] - pv.v[i][j]) * (FAST ? T(1) : mi.v[i][j]);
                        }
                }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < LC; ++j) Gp[t].v[i][j] = Gc[t].v[i][j] = T(0);
            }
        }
        auto next_plane = [&](int zl) -> const T* {                 // plane zl+1 if this chunk needs it
            const T* pn = zplane<T>(g, x, xp, xn, 2, zl + 1);
            return (pn != nullptr && (g.za || zl + 1 < ze)) ? pn : nullptr;
        };
        {
            const T* pn = next_plane(z_lo);
#pragma unroll
            for (int d = 0; d < H; ++d) load_rows(frame(pn, d), HEADS ? Nh[HEADS ? d : 0] : Nq[d]);
        }
        __syncthreads();
        const T* pc_c = zplane<T>(g, x, xp, xn, 2, z_lo);
        const T* pn_c = next_plane(z_lo);

        for (int zl = z_lo; zl <= ze; ++zl) {
            const int gz = g.z0 + zl, par = zl & 1;
            const bool plane_in = (gz >= 0) && (gz < g.nzg) && (zl < ze || g.za);
            const T* pc = pc_c;                                            // carried: one zplane() per step instead of three
            const T* pn = pn_c;
            const T* pn2 = (zl + 1 <= ze) ? next_plane(zl + 1) : nullptr;
            pc_c = zplane<T>(g, x, xp, xn, 2, zl + 1);
            pn_c = pn2;
            const bool z_prev = plane_in && (gz > 0), z_next = plane_in && (gz + 1 < g.nzg);
            const T wzn = (CEN ? (z_prev && z_next) : z_next) ? wz_u : T(0);
            const T m_pl = plane_in ? T(1) : T(0);
            const bool count = plane_in && (zl >= zs) && (zl < ze);
            const bool store = (zl - 1 >= zs) && (zl - 1 < ze);
            sg3_v2f hu[2] = {zero2, zero2}, hd[2] = {zero2, zero2};
#pragma unroll
            for (int k = 0; k < 2 && k < M; ++k) {
                if (DN || CEN) hu[k] = sg3_ld2(frame(pc, k), uoff);
                if (UP || CEN) hd[k] = sg3_ld2(frame(pc, k), doff);
            }
            sg3_v2f x0q[R];                           // MODE 1: x0 of the frame about to be stored, requested a frame ahead
            if (MODE == 1) {
                const Rsrc r0 = sg2_rsrc<T>(sa.x0 + (long long)(zl - 1) * g.s_z + foff_t(0), store && fstore(0), fbytes);
#pragma unroll
                for (int i = 0; i < R; ++i) x0q[i] = sg3_ld2(r0, soff[i]);
            }
            if (HEADS) {
#pragma unroll
                for (int d = 0; d < H; ++d) Nq[d] = Nh[HEADS ? d : 0];
            }
            sg3_v2f yu_n = zero2, yd_n = zero2;
            if (UP || CEN) yu_n = ye[par ^ 1][0][r_up][lane];
            if (DN || CEN) yd_n = ye[par ^ 1][0][r_dn][lane];
            C pf_t_prev, f_t_prev, c_old_prev;      // time-axis carries: product / forward difference / x of frame t-1
#pragma unroll
            for (int i = 0; i < R; ++i)
#pragma unroll
                for (int j = 0; j < LC; ++j) pf_t_prev.v[i][j] = f_t_prev.v[i][j] = c_old_prev.v[i][j] = T(0);
#pragma unroll
            for (int t = 0; t < M; ++t) {
                const C c = Cc[t];
                const C nx = Nq[t % SG2_D];          // x(zl+1, t)
                // ---- the rows above / below the strip, requested two frames ahead ---------------------------------------------
                const T xu[LC] = {hu[t & 1].x, hu[t & 1].y}, xd[LC] = {hd[t & 1].x, hd[t & 1].y};
                if (t + 2 < M) {
                    if (DN || CEN) hu[t & 1] = sg3_ld2(frame(pc, t + 2), uoff);
                    if (UP || CEN) hd[t & 1] = sg3_ld2(frame(pc, t + 2), doff);
                }
                sg3_v2f x0n[R];
                if (MODE == 1 && t + 1 < M) {
                    const Rsrc r0 = sg2_rsrc<T>(sa.x0 + (long long)(zl - 1) * g.s_z + foff_t(t + 1), store && fstore(t + 1), fbytes);
#pragma unroll
                    for (int i = 0; i < R; ++i) x0n[i] = sg3_ld2(r0, soff[i]);
                }
                // ---- raw differences --------------------------------------------------------------------------------------
                // dr[k][j] = x(row k) - x(row k-1), k = 0 .. R (row -1 = the row above the strip, row R the row below)
                T dr[R + 1][LC];
                C xr, xl;
#pragma unroll
                for (int j = 0; j < LC; ++j) {
                    dr[0][j] = c.v[0][j] - xu[j];
#pragma unroll
                    for (int i = 0; i + 1 < R; ++i) dr[i + 1][j] = c.v[i + 1][j] - c.v[i][j];
                    dr[R][j] = xd[j] - c.v[R - 1][j];
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    xl.v[i][0] = from_left(c.v[i][LC - 1]);
                    xr.v[i][0] = c.v[i][1];
                    xl.v[i][1] = c.v[i][0];
                    xr.v[i][1] = from_right(c.v[i][0]);
                }
                C f_r, b_r, f_c, b_c, f_z, b_z, f_t, b_t;     // weighted channels (central: f_* only)
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < LC; ++j) {
                        if (CEN) {
                            f_r.v[i][j] = ((i + 1 < R) ? c.v[(i + 1 < R) ? i + 1 : i][j] : xd[j]) - ((i > 0) ? c.v[(i > 0) ? i - 1 : 0][j] : xu[j]);
                            f_c.v[i][j] = xr.v[i][j] - xl.v[i][j];
                            f_z.v[i][j] = wzn * (nx.v[i][j] - Pz[t].v[i][j]);
                            if (!FAST) { f_r.v[i][j] *= mfr.v[i][j] * m_pl; f_c.v[i][j] *= mfc.v[i][j] * m_pl; f_z.v[i][j] *= mi.v[i][j]; }
                            b_r.v[i][j] = b_c.v[i][j] = b_z.v[i][j] = T(0);
                        } else {
                            f_r.v[i][j] = dr[i + 1][j];
                            b_r.v[i][j] = dr[i][j];
                            f_c.v[i][j] = xr.v[i][j] - c.v[i][j];
                            b_c.v[i][j] = c.v[i][j] - xl.v[i][j];
                            f_z.v[i][j] = wzn * (nx.v[i][j] - c.v[i][j]);
                            b_z.v[i][j] = DN ? Pz[DN ? t : 0].v[i][j] : T(0);    // = the forward difference of plane zl-1, weights and masks included
                            if (!FAST) {
                                f_r.v[i][j] *= mfr.v[i][j] * m_pl; b_r.v[i][j] *= mbr.v[i][j] * m_pl;
                                f_c.v[i][j] *= mfc.v[i][j] * m_pl; b_c.v[i][j] *= mbc.v[i][j] * m_pl;
                                f_z.v[i][j] *= mi.v[i][j];
                            }
                        }
                    }
                // time axis: forward difference to frame t+1 (of the OLD plane zl: Cc[t+1] has not been rotated yet)
                bool has_tn, has_tp;        // frame t+1 / t-1 exists in the volume
                if (!TWIN) { has_tn = (t + 1 < M); has_tp = (t > 0); }
                else { has_tn = fvalid(t) && (t0 + t + 1 < Mg); has_tp = fvalid(t) && (t0 + t > 0); }
                C xtn, xtp = c_old_prev;    // x(zl, t+1), x(zl, t-1)
                if (t + 1 < M) xtn = Cc[(t + 1 < M) ? t + 1 : t];
                else if (TWIN) load_rows(sg2_rsrc<T>(pc + foff_t(t + 1), pc != nullptr && has_tn, fbytes), xtn);
                else xtn = c;
                if (TWIN && t == 0 && (CEN || DN)) load_rows(sg2_rsrc<T>(pc + foff_t(-1), pc != nullptr && has_tp, fbytes), xtp);
                // per-VOXEL weight of the time channels (tv_geom::time_weight_vol; generic variant only)
                C wv_t;
                if (!TFAST) {
                    const T* wpl = vol_plane<T>(g, zl, t0 + t);          // uniform: the weight frame of (zl, t), ghost planes included
                    const Rsrc rw = sg2_rsrc<T>(wpl, wpl != nullptr, fbytes);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const sg3_v2f v = sg3_ld2(rw, roff[i]);
                        wv_t.v[i][0] = (wpl != nullptr) ? v.x : T(1);
                        wv_t.v[i][1] = (wpl != nullptr) ? v.y : T(1);
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < LC; ++j) {
                        const T wti = FAST ? wt_u : (TFAST ? wt_u * (mi.v[i][j] * m_pl) : (mft.v[i][j] * m_pl) * wv_t.v[i][j]);
                        if (CEN) {
                            f_t.v[i][j] = (has_tn && has_tp) ? wti * (xtn.v[i][j] - xtp.v[i][j]) : T(0);
                            b_t.v[i][j] = T(0);
                        } else if (TFAST) {
                            f_t.v[i][j] = has_tn ? wti * (xtn.v[i][j] - c.v[i][j]) : T(0);
                            b_t.v[i][j] = f_t_prev.v[i][j];
                            if (TWIN && t == 0 && DN) b_t.v[i][j] = has_tp ? wti * (c.v[i][j] - xtp.v[i][j]) : T(0);   // backward difference of the window's first frame
                        } else {
                            const T dt = has_tn ? (xtn.v[i][j] - c.v[i][j]) : T(0);
                            f_t.v[i][j] = wti * dt;
                            b_t.v[i][j] = wti * f_t_prev.v[i][j];                   // f_t_prev carries the RAW difference here
                            if (TWIN && t == 0 && DN) b_t.v[i][j] = has_tp ? wti * (c.v[i][j] - xtp.v[i][j]) : T(0);
                            f_t_prev.v[i][j] = dt;
                        }
                    }
                if (TFAST || CEN) f_t_prev = f_t;
                c_old_prev = c;
                // ---- 1 / |Dx| -------------------------------------------------------------------------------------------
                C ss, n;
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < LC; ++j) {
                        T a = f_r.v[i][j] * f_r.v[i][j] + f_c.v[i][j] * f_c.v[i][j];
                        a = a + f_z.v[i][j] * f_z.v[i][j];
                        a = a + f_t.v[i][j] * f_t.v[i][j];
                        if (S == HYBRID || S == DOWNWIND) {
                            T b = b_r.v[i][j] * b_r.v[i][j] + b_c.v[i][j] * b_c.v[i][j];
                            b = b + b_z.v[i][j] * b_z.v[i][j];
                            b = b + b_t.v[i][j] * b_t.v[i][j];
                            a = (S == HYBRID) ? a + b : b;
                        }
                        ss.v[i][j] = a;
                    }
                {
                    T mn = __builtin_fminf(__builtin_fminf(ss.v[0][0], ss.v[0][1]), __builtin_fminf(ss.v[1][0], ss.v[1][1]));
                    static_assert(R == 2 && LC == 2, "the minimum above is written for four sites");
                    // a wave that holds no vanishing gradient (the usual case) skips the selects
                    if (__builtin_expect(__any(!(mn >= thr)), 0)) {
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int j = 0; j < LC; ++j) n.v[i][j] = (ss.v[i][j] >= thr) ? rsq_fast(ss.v[i][j]) : T(0);
                    } else {
#pragma unroll
                        for (int i = 0; i < R; ++i)
#pragma unroll
                            for (int j = 0; j < LC; ++j) n.v[i][j] = rsq_fast(ss.v[i][j]);
                    }
                }
                // TV: |Dx| = s * ss * (1 / sqrt(ss)); sites the tile owns, planes this chunk owns (uniform multiplier, no branch)
                {
                    const bool cnt = count && fstore(t);
                    T sum = T(0);
                    Rsrc rn_rs;
                    if (MODE == 2) rn_rs = sg2_rsrc<T>(sa.norms + (long long)zl * g.s_z + foff_t(t), cnt, fbytes);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        T rn[LC];
#pragma unroll
                        for (int j = 0; j < LC; ++j) {
                            rn[j] = ss.v[i][j] * n.v[i][j];
                            sum += rn[j] * cm.v[i][j];
                        }
                        if (MODE == 2) sg3_st2(rn_rs, soff[i], (n.v[i][0] > T(0)) ? s * rn[0] : (T)__builtin_inff(), (n.v[i][1] > T(0)) ? s * rn[1] : (T)__builtin_inff());
                    }
                    acc += (double)(sum * (cnt ? s : T(0)));
                    pin1<double>(acc);        // or LLVM sinks the sums of all M frames below the frame loop and keeps their operands alive
                }
                // ---- scatter the products -------------------------------------------------------------------------------
                C gc = Gc[t], gn;
                T to_up[LC], to_dn[LC];                // products handed to the wave above / below
#pragma unroll
                for (int j = 0; j < LC; ++j) to_up[j] = to_dn[j] = T(0);
                if (CEN) {
                    // one product per axis, +1/2 to the site after, -1/2 to the site before
                    C p_c;
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int j = 0; j < LC; ++j) p_c.v[i][j] = f_c.v[i][j] * n.v[i][j];
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int j = 0; j < LC; ++j) {
                            const T p_r = f_r.v[i][j] * n.v[i][j];
                            if (i + 1 < R) gc.v[(i + 1 < R) ? i + 1 : i][j] += p_r; else to_dn[j] = p_r;
                            if (i > 0) gc.v[(i > 0) ? i - 1 : 0][j] -= p_r; else to_up[j] = p_r;
                            const T pl = (j > 0) ? p_c.v[i][(j > 0) ? j - 1 : 0] : from_left(p_c.v[i][LC - 1]);
                            const T pr = (j + 1 < LC) ? p_c.v[i][(j + 1 < LC) ? j + 1 : j] : from_right(p_c.v[i][0]);
                            gc.v[i][j] += pl - pr;
                            const T p_z = f_z.v[i][j] * n.v[i][j];
                            gn.v[i][j] = p_z;
                            Gp[t].v[i][j] -= p_z;
                            const T p_t = f_t.v[i][j] * n.v[i][j];
                            gc.v[i][j] += pf_t_prev.v[i][j];
                            pf_t_prev.v[i][j] = p_t;
                            if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i][j] -= p_t; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i][j]); }   // G(zl, t-1): rotated into Gp at the end of frame t-1
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int j = 0; j < LC; ++j) gn.v[i][j] = T(0);
                    if constexpr (S == HYBRID) {
                        // hybrid, FLUX form (tv_subgrad2.h): an edge (v, v + e) with difference d hands d (n(v) + n(v + e)) to v + e and
                        // takes the same from v.  Rows inside the strip, the columns (1 / |Dx| of the right neighbour: in the lane or
                        // one DPP move) and -- FAST / column-border variants -- the frames; the strip's end rows and the planes stay scatters.
#pragma unroll
                        for (int j = 0; j < LC; ++j) {
#pragma unroll
                            for (int i = 0; i + 1 < R; ++i) {
                                const T ph = f_r.v[i][j] * (n.v[i][j] + n.v[(i + 1 < R) ? i + 1 : i][j]);
                                gc.v[i][j] -= ph;
                                gc.v[(i + 1 < R) ? i + 1 : i][j] += ph;
                            }
                            to_up[j] = b_r.v[0][j] * n.v[0][j];
                            gc.v[0][j] += to_up[j];
                            to_dn[j] = f_r.v[R - 1][j] * n.v[R - 1][j];
                            gc.v[R - 1][j] -= to_dn[j];
                        }
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            // column fluxes of the row: pc[j] = the edge (col j -> col j + 1)
                            T pcx[LC];
                            pcx[0] = f_c.v[i][0] * (n.v[i][0] + n.v[i][1]);
                            pcx[1] = f_c.v[i][1] * (n.v[i][1] + from_right(n.v[i][0]));
                            gc.v[i][0] += from_left(pcx[1]) - pcx[0];
                            gc.v[i][1] += pcx[0] - pcx[1];
#pragma unroll
                            for (int j = 0; j < LC; ++j) {
                                const T pz_f = f_z.v[i][j] * n.v[i][j], pz_b = b_z.v[i][j] * n.v[i][j];
                                gc.v[i][j] += pz_b - pz_f;
                                gn.v[i][j] = pz_f;
                                Gp[t].v[i][j] -= pz_b;
                                if (TFAST) {
                                    T pt_;
                                    if (TWIN && t == 0) pt_ = b_t.v[i][j] * n.v[i][j];          // the edge into the window: its other end is not ours
                                    else pt_ = b_t.v[i][j] * (pf_t_prev.v[i][j] + n.v[i][j]);      // pf_t_prev carries 1 / |Dx| of frame t-1 here
                                    gc.v[i][j] += pt_;
                                    if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i][j] -= pt_; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i][j]); }
                                    pf_t_prev.v[i][j] = n.v[i][j];
                                } else {
                                    const T p_t = f_t.v[i][j] * n.v[i][j], q_t = b_t.v[i][j] * n.v[i][j];
                                    gc.v[i][j] += (pf_t_prev.v[i][j] - p_t) + q_t;
                                    pf_t_prev.v[i][j] = p_t;
                                    if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i][j] -= q_t; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i][j]); }
                                }
                            }
                        }
                    } else
                    if (UP) {       // forward channels: - to the site itself, + to the next site
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            T p_c[LC];
#pragma unroll
                            for (int j = 0; j < LC; ++j) p_c[j] = f_c.v[i][j] * n.v[i][j];
                            gc.v[i][0] += from_left(p_c[LC - 1]) - p_c[0];
                            gc.v[i][1] += p_c[0] - p_c[1];
#pragma unroll
                            for (int j = 0; j < LC; ++j) {
                                const T p_r = f_r.v[i][j] * n.v[i][j];
                                gc.v[i][j] -= p_r;
                                if (i + 1 < R) gc.v[(i + 1 < R) ? i + 1 : i][j] += p_r; else to_dn[j] = p_r;
                                const T p_z = f_z.v[i][j] * n.v[i][j];
                                gc.v[i][j] -= p_z;
                                gn.v[i][j] = p_z;
                                const T p_t = f_t.v[i][j] * n.v[i][j];
                                gc.v[i][j] += pf_t_prev.v[i][j] - p_t;
                                pf_t_prev.v[i][j] = p_t;
                            }
                        }
                    }
                    if (DN && S != HYBRID) {       // backward channels: + to the site itself, - to the previous site
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            T p_c[LC];
#pragma unroll
                            for (int j = 0; j < LC; ++j) p_c[j] = b_c.v[i][j] * n.v[i][j];
                            gc.v[i][0] += p_c[0] - p_c[1];
                            gc.v[i][1] += p_c[1] - from_right(p_c[0]);
#pragma unroll
                            for (int j = 0; j < LC; ++j) {
                                const T p_r = b_r.v[i][j] * n.v[i][j];
                                gc.v[i][j] += p_r;
                                if (i > 0) gc.v[(i > 0) ? i - 1 : 0][j] -= p_r; else to_up[j] = p_r;
                                const T p_z = b_z.v[i][j] * n.v[i][j];
                                gc.v[i][j] += p_z;
                                Gp[t].v[i][j] -= p_z;
                                const T p_t = b_t.v[i][j] * n.v[i][j];
                                gc.v[i][j] += p_t;
                                if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i][j] -= p_t; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i][j]); }   // G(zl, t-1): rotated into Gp at the end of frame t-1
                            }
                        }
                    }
                }
                if (DN || CEN) ye[par][t][r_own][lane] = sg3_mk(to_up[0], to_up[1]);
                if (UP || CEN) ye[par][t][r_own + 1][lane] = sg3_mk(to_dn[0], to_dn[1]);
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < LC; ++j) pin1<T>(gc.v[i][j]);
                if (MODE == 1) {
                    // the descent step's share of x rides in the accumulator (tv_subgrad2.h): x_out = step x0 - a (G + kap x)
#pragma unroll
                    for (int i = 0; i < R; ++i)
#pragma unroll
                        for (int j = 0; j < LC; ++j) gc.v[i][j] += kap * c.v[i][j];
                }
                // ---- plane zl-1 is complete: store (sites the tile owns; everything else has an out-of-range offset) -----------
                {
                    const bool st = store && fstore(t);
                    const long long foff = (long long)(zl - 1) * g.s_z + foff_t(t);      // uniform
                    // the row products the neighbouring waves published in the previous step (zero row at the block's ends)
                    if (UP || CEN) { Gp[t].v[0][0] += yu_n.x; Gp[t].v[0][1] += yu_n.y; }
                    if (DN || CEN) { Gp[t].v[R - 1][0] -= yd_n.x; Gp[t].v[R - 1][1] -= yd_n.y; }
                    if (t + 1 < M) {                 // next frame's, a frame ahead
                        if (UP || CEN) yu_n = ye[par ^ 1][(t + 1 < M) ? t + 1 : t][r_up][lane];
                        if (DN || CEN) yd_n = ye[par ^ 1][(t + 1 < M) ? t + 1 : t][r_dn][lane];
                    }
                    if (MODE != 1) {
                        const Rsrc rg = sg2_rsrc<T>(G + foff, st, fbytes);
#pragma unroll
                        for (int i = 0; i < R; ++i) sg3_st2(rg, soff[i], s * Gp[t].v[i][0], s * Gp[t].v[i][1]);
                    } else {
                        const Rsrc ro = sg2_rsrc<T>(sa.x_out + foff, st, fbytes);
                        T e2 = T(0);
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const T x0a = x0q[i].x, x0b = x0q[i].y;
                            const T xa = sa.step * x0a - a_step * Gp[t].v[i][0], xb = sa.step * x0b - a_step * Gp[t].v[i][1];
                            const T ea = xa - x0a, eb = xb - x0b;
                            e2 += (ea * ea) * cm.v[i][0] + (eb * eb) * cm.v[i][1];
                            sg3_st2(ro, soff[i], xa, xb);
                        }
                        acc_fid += (double)(e2 * (st ? T(0.5) : T(0)));
                        pin1<double>(acc_fid);
                        if (t + 1 < M) {
#pragma unroll
                            for (int i = 0; i < R; ++i) x0q[i] = x0n[i];
                        }
                    }
                }
                // rotate this frame: the finished slot carries the start of G(zl+1); x(zl) -> x(zl-1), x(zl+1) -> x(zl)
#pragma unroll
                for (int i = 0; i < R; ++i)
#pragma unroll
                    for (int j = 0; j < LC; ++j) pin1<T>(gn.v[i][j]);
                Gp[t] = gc;          // G(zl, t): still misses its z+1 term, the time term of frame t+1 and the cross-wave row terms
                Gc[t] = gn;          // the start of G(zl+1, t)
                if (CEN) Pz[CEN ? t : 0] = c;
                else if (DN) Pz[DN ? t : 0] = f_z;
                Cc[t] = nx;
                // next load of the ring: frame t + D of plane zl+1, or -- at the end of the step -- frame t + D - M of plane zl+2
                if (t + SG2_D < M) load_rows(frame(pn, t + SG2_D), Nq[t % SG2_D]);
                if (t >= M - H) load_rows(frame(pn2, t - (M - H)), HEADS ? Nh[(HEADS && t >= M - H) ? t - (M - H) : 0] : Nq[t % SG2_D]);
                __builtin_amdgcn_sched_barrier(0);   // keep the frames apart: interleaving them costs registers (scratch) for nothing
            }
            // LDS-only barrier: the hand-off rows must be visible, nothing else (tv_subgrad2.h)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        acc = block_sum(acc, sh.sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) partials[lid] = acc;
        if (MODE == 1) {
            acc_fid = block_sum(acc_fid, sh.sm);
            if (threadIdx.x == 0 && threadIdx.y == 0) sa.part_fid[lid] = acc_fid;
        }
    }
};

// grid layout, tile order and the three block variants exactly as k_subgrad_col (tv_subgrad2.h)
template <int S, int M, int MODE, bool TWIN>
__global__ __launch_bounds__(512, 2) void k_subgrad_pair(DG g, WT<float> w, const float* __restrict__ x, const float* __restrict__ xp,
                                                          const float* __restrict__ xn, float* __restrict__ G, int zchunk, int nchunks,
                                                          double* __restrict__ partials, SgArgs2<float> sa, SgTiles tm) {
    using K = SgPair<S, M, MODE, TWIN>;
    __shared__ typename K::Shared sh;
    const int Mg = TWIN ? g.m : M;
    const int nwin = TWIN ? (Mg + SG2_TWU - 1) / SG2_TWU : 1;
    const long long per_tile = (long long)nchunks * nwin;
    const long long nb_border = tm.nborder * per_tile, nb_fast = tm.nfast * per_tile;
    const long long pb = (nb_border + 7) / 8 * 8;                  // grid: [border range padded to 8][interior range padded to 8]
    const bool fast = (long long)blockIdx.x >= pb;
    const long long id = fast ? (long long)blockIdx.x - pb : (long long)blockIdx.x;
    const long long total = fast ? nb_fast : nb_border, ntiles = fast ? tm.nfast : tm.nborder, per_xcd = (total + 7) / 8;
    const long long lid = (id % 8) * per_xcd + id / 8;
    if (lid >= total) return;
    const int win = (int)(lid / (ntiles * nchunks));
    const int chunk = (int)((lid / ntiles) % nchunks);
    int bx, by;
    sg2_tile(tm, fast, lid % ntiles, bx, by);
    const long long slot = (fast ? nb_border : 0) + lid;
    const bool colb = !fast && by >= tm.iy0 && by <= tm.iy1;
    if (fast) K::template run<0>(g, w, x, xp, xn, G, zchunk, chunk, bx, by, win, slot, partials, sa, sh);
    else if (colb) K::template run<1>(g, w, x, xp, xn, G, zchunk, chunk, bx, by, win, slot, partials, sa, sh);
    else K::template run<2>(g, w, x, xp, xn, G, zchunk, chunk, bx, by, win, slot, partials, sa, sh);
}

}  // namespace tv
