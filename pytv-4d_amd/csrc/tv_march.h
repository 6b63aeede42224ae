// tv_march.h -- plane-marching fast path of the forward (D + epilogue) and transposed
// (D^T + epilogue) kernels for fp32, 16-byte lanes.
//
// Why: the one-voxel-per-thread kernels in tv_kernels.hip re-read every z / t / row-halo neighbour
// from beyond L2 (rocprofv3 FETCH_SIZE on MI355X, 256x8x1024x1024: 116 GB read for 77 GB
// algorithmic in the dual kernel) because the 16 gradient streams flush the 4 MiB XCD L2 between
// the two touches of an image line.  Here every thread owns one (row, 4-column) site, marches
// along z inside a z-chunk and keeps the M time frames of the current (and previous) plane in
// registers (M is a template parameter: the frame loop is fully unrolled, no runtime-indexed
// register arrays).  Neighbours then come from:
//     z-1 / z+1   registers carried from the previous step / the one new load of the step
//     t-1 / t+1   registers (unrolled frame array)
//     col -1 / +1 the adjacent lane via a wave shuffle (DPP/bpermute); strip edges: one 2-lane load
//     row -1 / +1 forward kernel: the block's 4-row LDS tile (halo rows: one global load per frame)
//                 transposed kernel: plain loads (same-block rows hit L1/L2; measured)
// so each image / gradient word is fetched from HBM once (plus 2 of every 4 rows as halo).
#pragma once
#include "tv_device.h"
#include "tv_stencil.h"

namespace tv {

using V4 = Vec<float, 4>;

struct MarchCoord {
    int lane, ty, col0, y, zs, ze;
    bool ok;
    long long inpl;
};

__device__ __forceinline__ MarchCoord march_coord(const DG& g, int zchunk, int z_first = 0, int z_end = -1) {
    MarchCoord c;
    c.lane = (int)threadIdx.x;
    c.ty = (int)threadIdx.y;
    const int nxv = g.nx / 4;
    const int tiles_x = (nxv + 63) / 64;
    const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
    c.col0 = (bx * 64 + c.lane) * 4;
    c.y = by * 4 + c.ty;
    c.ok = (c.col0 < g.nx) && (c.y < g.ny);
    if (z_end < 0) z_end = g.nz;
    c.zs = z_first + (int)blockIdx.y * zchunk;          // z_first < 0 / z_end > nz: ghost planes of a slab
    c.ze = (c.zs + zchunk < z_end) ? c.zs + zchunk : z_end;
    c.inpl = (long long)c.y * g.nx + c.col0;
    return c;
}

// value of the element just left of this lane's vector / just right of it, for a vector `v` that
// every lane of the wave holds for the same row: neighbours inside the strip come from the adjacent
// lane, the two strip-edge lanes read one scalar each from memory (p points at this lane's vector).
template <bool LEFT, bool RIGHT>
__device__ __forceinline__ void col_neighbours(const V4& v, const float* p, bool ok, int lane, int col0, int nx,
                                               float& left, float& right) {
    left = 0.f;
    right = 0.f;
    float edge = 0.f;
    const bool le = LEFT && (lane == 0) && ok && (col0 > 0);
    const bool re = RIGHT && (lane == 63) && ok && (col0 + 4 < nx);
    if (le || re) edge = le ? p[-1] : p[4];
    if (LEFT) {
        const float s = __shfl_up(v.v[3], 1, 64);
        left = (lane == 0) ? edge : s;
    }
    if (RIGHT) {
        const float s = __shfl_down(v.v[0], 1, 64);
        right = (lane == 63) ? edge : s;
    }
}

// =============================================================================================
// forward operator, marching.  Epi is one of the forward epilogues of tv_kernels.hip
// (StoreD, CpDual, AdmmZU): called once per (z, t) with the gradient channels of the site.
// =============================================================================================
// hp: planes per halo buffer (1, or 2 for the radius-2 entry points); [z_first, z_end): local planes to
// visit (ghost planes outside [0, nz) come from the halo buffers)
// LIGHT: for epilogues with little traffic per site (StoreD, NormEpi): single-buffered tile (two barriers per
// plane, 32 KiB at M = 8 -> up to 5 blocks/CU) and a register cap of 168 so that 3 waves/SIMD hide the latency
// PF: 1 = all M frames of plane z+1 are requested at the top of the z step (M independent loads in flight per wave
// instead of one exposed latency per frame: the epilogue's stores may alias x as far as the compiler knows, so it never
// hoists the next frame's load above them); 2 = the halo row of the first / last tile row as well
// SINGLE: single-buffered tile with two barriers per plane even without the LIGHT register cap (M = 16: the
// double-buffered tile would take 128 KiB of LDS and leave one block = one wave per SIMD on the CU)
template <int S, int M, typename Epi, bool LIGHT = false, int PF = 0, bool SINGLE = (LIGHT || M > 8)>
__global__ __launch_bounds__(256, LIGHT ? 3 : ((PF || M > 8) ? 2 : 1)) void k_D_march(DG g, WT<float> w, const float* __restrict__ x, const float* __restrict__ xp,
                                                 const float* __restrict__ xn, int zchunk, Epi epi, int hp = 1, int z_first = 0,
                                                 int z_end = -1) {
    __shared__ V4 tile[SINGLE ? 1 : 2][M][4][64];   // double-buffered: one barrier per z step
    __shared__ double sm[16];
    const MarchCoord c = march_coord(g, zchunk, z_first, z_end);
    constexpr bool NEXT = (S != DOWNWIND), PREV = (S != UPWIND);
    const V4 zero = vsplat<float, 4>(0.f);
    const V4 mf = g.ta ? mask_factor<float, 4>(g, w.sf, c.ok ? c.y : 0, c.ok ? c.col0 : 0) : vsplat<float, 4>(1.f);
    double acc = 0.0;

    V4 C[M], P[M];
    {   // prologue: planes zs (centre) and zs-1
        const float* pc = zplane<float>(g, x, xp, xn, hp, c.zs);
        const float* pp = (PREV && g.za) ? zplane<float>(g, x, xp, xn, hp, c.zs - 1) : nullptr;
#pragma unroll
        for (int t = 0; t < M; ++t) {
            C[t] = c.ok ? vload<float, 4>(pc + (long long)t * g.s_t + c.inpl) : zero;
            P[t] = (c.ok && pp != nullptr) ? vload<float, 4>(pp + (long long)t * g.s_t + c.inpl) : zero;
        }
    }
    for (int z = c.zs; z < c.ze; ++z) {
        const float* pc = zplane<float>(g, x, xp, xn, hp, z);
        const float* pn = zplane<float>(g, x, xp, xn, hp, z + 1);
        const bool has_pz = PREV && g.za && (g.z0 + z > 0);
        const bool has_nz = NEXT && g.za && (pn != nullptr);
        // the next plane is needed as a z neighbour and/or as the next centre
        const bool load_next = (pn != nullptr) && c.ok && (has_nz || (z + 1 < c.ze));
        V4 Nn[PF ? M : 1], H[PF == 2 ? M : 1];
        const bool halo_up = PREV && (c.ty == 0) && c.ok && (c.y > 0);
        const bool halo_dn = NEXT && (c.ty == 3) && c.ok && (c.y + 1 < g.ny);
        if (PF) {
#pragma unroll
            for (int t = 0; t < M; ++t) Nn[t] = load_next ? vload<float, 4>(pn + (long long)t * g.s_t + c.inpl) : zero;
        }
        if (PF == 2) {
#pragma unroll
            for (int t = 0; t < M; ++t)
                H[t] = (halo_up || halo_dn) ? vload<float, 4>(pc + (long long)t * g.s_t + c.inpl + (halo_up ? -(long long)g.nx : (long long)g.nx)) : zero;
        }
        // publish plane z for the row neighbours
        const int buf = SINGLE ? 0 : ((z - c.zs) & 1);
#pragma unroll
        for (int t = 0; t < M; ++t) tile[buf][t][c.ty][c.lane] = C[t];
        __syncthreads();
        V4 cold = zero;                   // plane-z value of frame t-1 (C[t-1] is already overwritten)
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const long long off = (long long)t * g.s_t + c.inpl;
            V4 N;
            if constexpr (PF != 0) N = Nn[t];
            else N = load_next ? vload<float, 4>(pn + off) : zero;
            XN<float, 4> n;
            n.c = C[t];
            n.col0 = c.col0;
            n.nr = n.pr = n.nc = n.pc = n.nz = n.pz = n.nt = n.pt = zero;
            n.h_nr = n.h_pr = n.h_nz = n.h_pz = n.h_nt = n.h_pt = false;
            if (NEXT) {
                n.h_nr = c.ok && (c.y + 1 < g.ny);
                if (c.ty < 3) n.nr = tile[buf][t][c.ty + 1][c.lane];
                else if (PF == 2) n.nr = H[t];
                else if (n.h_nr) n.nr = vload<float, 4>(pc + off + g.nx);
                n.h_nz = has_nz;
                n.nz = N;
                if (t + 1 < M) { n.h_nt = (g.ta != 0); n.nt = C[(t + 1 < M) ? t + 1 : t]; }
            }
            if (PREV) {
                n.h_pr = c.ok && (c.y > 0);
                if (c.ty > 0) n.pr = tile[buf][t][c.ty - 1][c.lane];
                else if (PF == 2) n.pr = H[t];
                else if (n.h_pr) n.pr = vload<float, 4>(pc + off - g.nx);
                n.h_pz = has_pz;
                n.pz = P[t];
                if (t > 0) { n.h_pt = (g.ta != 0); n.pt = cold; }
            }
            float left, right;
            col_neighbours<PREV, NEXT>(C[t], pc + off, c.ok, c.lane, c.col0, g.nx, left, right);
            if (NEXT) n.nc = shift_left<float, 4>(C[t], right);
            if (PREV) n.pc = shift_right<float, 4>(C[t], left);
            if (c.ok) {
                V4 o[8];
                d_slots<S, float, 4>(g, w, n, mf, o);
                Coord cc;
                cc.zl = z; cc.t = t; cc.y = c.y; cc.col0 = c.col0; cc.ok = true;
                if constexpr (requires { epi(g, cc, o, n.c); }) acc += epi(g, cc, o, n.c);
                else acc += epi(g, cc, o);
            }
            cold = C[t];
            P[t] = C[t];
            C[t] = N;
        }
        // double-buffered: no second barrier (the other buffer is written next, and it was last read before
        // this step's barrier); single-buffered: everyone must be done reading before the tile is rewritten
        if (SINGLE) __syncthreads();
    }
    if (Epi::REDUCES) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) epi.partials[linear_block_id()] = acc;
    }
}

// =============================================================================================
// transposed operator, marching.  Epi is one of the D^T epilogues (StoreDT, AxpyDT, CpPrimal).
// zmode / tmode: 0 = adjoint of a forward difference, 1 = of a backward difference, 2 = central.
// State per frame: A = backward-looking z channel one plane below (y^(z-1)), B = the forward-looking
// channel's centre plane (loaded as "z+1" one step earlier); central keeps (z-1, z) of one channel.
// =============================================================================================
#ifndef TV_DT_NT
#define TV_DT_NT 0               // 1: EXPERIMENT the read-once streams of k_DT_march non-temporal: much SLOWER (tv_DT hybrid 3.6 -> 6.3 ms: the border scalars and row taps of neighbouring threads want those lines in the cache)
#endif
template <typename T, int V> __device__ __forceinline__ Vec<T, V> DTLD(const T* p) {
#if TV_DT_NT
    return vload_s<T, V>(p);
#else
    return vload<T, V>(p);
#endif
}
// the same for one 16-byte lane of T (fp64, round 4: two columns per lane, 128 columns per wave row)
template <int V> __device__ __forceinline__ MarchCoord march_coord_v(const DG& g, int zchunk) {
    MarchCoord c;
    c.lane = (int)threadIdx.x;
    c.ty = (int)threadIdx.y;
    const int nxv = g.nx / V;
    const int tiles_x = (nxv + 63) / 64;
    const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
    c.col0 = (bx * 64 + c.lane) * V;
    c.y = by * 4 + c.ty;
    c.ok = (c.col0 < g.nx) && (c.y < g.ny);
    c.zs = (int)blockIdx.y * zchunk;
    c.ze = (c.zs + zchunk < g.nz) ? c.zs + zchunk : g.nz;
    c.inpl = (long long)c.y * g.nx + c.col0;
    return c;
}
template <bool LEFT, bool RIGHT, typename T, int V>
__device__ __forceinline__ void col_neighbours_v(const Vec<T, V>& v, const T* p, bool ok, int lane, int col0, int nx, T& left, T& right) {
    left = T(0);
    right = T(0);
    T edge = T(0);
    const bool le = LEFT && (lane == 0) && ok && (col0 > 0);
    const bool re = RIGHT && (lane == 63) && ok && (col0 + V < nx);
    if (le || re) edge = le ? p[-1] : p[V];
    if (LEFT) {
        const T s = __shfl_up(v.v[V - 1], 1, 64);
        left = (lane == 0) ? edge : s;
    }
    if (RIGHT) {
        const T s = __shfl_down(v.v[0], 1, 64);
        right = (lane == 63) ? edge : s;
    }
}
// T: float (16-byte lanes of 4 columns) or -- round 4 -- double (2 columns): the same registers, the same 16-byte accesses.  (double is
// instantiated with the AxpyDT epilogue only: with StoreDT the compiler gives hybrid M = 8 256 VGPRs and one wave per SIMD -- 7.0 ms against
// 5.0 one-site -- or, capped at 168, 484 bytes of scratch; the plain store runs as AxpyDT with no base and alpha = 1: 135 VGPRs.)
template <int S, int M, typename Epi, typename T = float>
__global__ __launch_bounds__(256) void k_DT_march(DG g, WT<T> w, const T* __restrict__ q, const T* __restrict__ qp,
                                                  const T* __restrict__ qn, int zchunk, Epi epi) {
    constexpr int V = 16 / (int)sizeof(T);
    using V4 = Vec<T, V>;                  // (the name of the fp32 original: one 16-byte lane)
    __shared__ double sm[16];
    const MarchCoord c = march_coord_v<V>(g, zchunk);
    const V4 zero = vsplat<T, V>(T(0));
    const V4 mf = g.ta ? mask_factor<T, V>(g, w.sf, c.ok ? c.y : 0, c.ok ? c.col0 : 0) : vsplat<T, V>(T(1));
    double acc = 0.0;
    constexpr bool HY = (S == HYBRID);
    // run-time central fallbacks (two-point axes use the forward stencil, i.e. mode 0)
    const bool z_cen = (S == CENTRAL) && !g.z_two;
    const bool t_cen = (S == CENTRAL) && !g.t_two;
    // channel whose z adjoint looks backwards (needs plane z-1) / forwards (needs plane z+1)
    const int ch_zb = g.ch_z, ch_zf = HY ? g.ch_z + 1 : g.ch_z;
    const int ch_tb = g.ch_t, ch_tf = HY ? g.ch_t + 1 : g.ch_t;
    constexpr bool ZB = (S == UPWIND || S == HYBRID || S == CENTRAL);    // some channel looks backwards in z
    constexpr bool ZF = (S == DOWNWIND || S == HYBRID || S == CENTRAL);  // some channel looks forwards in z

    auto ldq = [&](int zl, int ch, long long off) -> V4 {   // q[zl, ch] at in-plane offset off
        return DTLD<T, V>(q + (long long)zl * g.s_dz + (long long)ch * g.s_z + off);          // z / time channels: each sample is read once
    };

    V4 A[M], B[M];
#pragma unroll
    for (int t = 0; t < M; ++t) { A[t] = zero; B[t] = zero; }
    if (g.za && c.ok) {
        const int gz = g.z0 + c.zs;
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const long long off = (long long)t * g.s_t + c.inpl;
            if (ZB && gz >= 1) A[t] = (c.zs >= 1) ? ldq(c.zs - 1, ch_zb, off) : vload<T, V>(qp + off);
            if (S == DOWNWIND || HY) B[t] = ldq(c.zs, ch_zf, off);          // centre of the forward-looking channel
            if (S == CENTRAL) B[t] = ldq(c.zs, ch_zb, off);                // central: plane z of the same channel
        }
    }
    for (int z = c.zs; z < c.ze; ++z) {
        const int gz = g.z0 + z;
        V4 t_lo = zero;      // backward-looking time channel at t-1
        V4 t_ce = zero;      // forward-looking time channel at t (loaded as t+1 one iteration earlier)
        V4 t_c1 = zero;      // central: time channel at t (becomes t-1)
        if (g.ta && c.ok) {
            if (S == DOWNWIND || HY) t_ce = ldq(z, ch_tf, c.inpl);
            if (S == CENTRAL) t_c1 = ldq(z, ch_tb, c.inpl);
        }
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const long long inpl_t = (long long)t * g.s_t + c.inpl;
            const long long off0 = (long long)z * g.s_dz + inpl_t;              // channel 0 at this site
            V4 r = zero, rt = zero;
            if (c.ok) {
                // ---------------- rows ----------------------------------------------------------
                auto rows = [&](auto mode, int ch) {
                    constexpr int MD = decltype(mode)::value;
                    const T* pch = q + off0 + (long long)ch * g.s_z;
                    const V4 lo = (c.y >= 1 && MD != 1) ? vload<T, V>(pch - g.nx) : zero;
                    const V4 ce = (MD != 2) ? vload<T, V>(pch) : zero;
                    const V4 hi = (c.y + 1 < g.ny && MD != 0) ? vload<T, V>(pch + g.nx) : zero;
                    r = r + adj_axis<MD, T, V>(c.y, g.ny, lo, ce, hi);
                };
                if (S == UPWIND) rows(IC<0>{}, 0);
                if (S == DOWNWIND) rows(IC<1>{}, 0);
                if (S == CENTRAL) rows(IC<2>{}, 0);
                if (HY) { rows(IC<0>{}, 0); rows(IC<1>{}, 2); }
            }
            // ---------------- columns (shuffles are wave-wide: outside the ok branch) --------------
            auto cols = [&](auto mode, int ch) {
                constexpr int MD = decltype(mode)::value;
                const T* pch = q + off0 + (long long)ch * g.s_z;
                const V4 ce = c.ok ? DTLD<T, V>(pch) : zero;                 // column channels: the row neighbours of the ROW channels come from
                T left, right;                                     // the cache (plain loads), everything else is read once
                col_neighbours_v<(MD != 1), (MD != 0), T, V>(ce, pch, c.ok, c.lane, c.col0, g.nx, left, right);
                const V4 lo = shift_right<T, V>(ce, left);
                const V4 hi = shift_left<T, V>(ce, right);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const int col = c.col0 + i;
                    T a, b;
                    if (MD == 0) { a = (col >= 1) ? lo.v[i] : T(0); b = (col <= g.nx - 2) ? ce.v[i] : T(0); }
                    else if (MD == 1) { a = (col >= 1) ? ce.v[i] : T(0); b = (col <= g.nx - 2) ? hi.v[i] : T(0); }
                    else { a = (col >= 2) ? lo.v[i] : T(0); b = (col <= g.nx - 3) ? hi.v[i] : T(0); }
                    r.v[i] += a - b;
                }
            };
            if (S == UPWIND) cols(IC<0>{}, 1);
            if (S == DOWNWIND) cols(IC<1>{}, 1);
            if (S == CENTRAL) cols(IC<2>{}, 1);
            if (HY) { cols(IC<0>{}, 1); cols(IC<1>{}, 3); }
            if (c.ok) {
                // ---------------- z ---------------------------------------------------------------
                if (g.za) {
                    V4 hi = zero;       // plane z+1 of the forward-looking (or central) channel
                    if (ZF && (S != CENTRAL || z_cen) && gz + 1 < g.nzg)
                        hi = (z + 1 < g.nz) ? ldq(z + 1, ch_zf, inpl_t) : vload<T, V>(qn + inpl_t);
                    if (S == UPWIND || (S == CENTRAL && !z_cen)) {
                        const V4 ce = (S == CENTRAL) ? B[t] : ldq(z, ch_zb, inpl_t);
                        r = r + w.wz * adj_axis<0, T, V>(gz, g.nzg, A[t], ce, zero);
                        A[t] = ce;
                        if (S == CENTRAL) B[t] = (z + 1 < c.ze) ? ldq(z + 1, ch_zb, inpl_t) : zero;
                    } else if (S == DOWNWIND) {
                        r = r + w.wz * adj_axis<1, T, V>(gz, g.nzg, zero, B[t], hi);
                        B[t] = hi;
                    } else if (S == CENTRAL) {
                        r = r + w.wz * adj_axis<2, T, V>(gz, g.nzg, A[t], zero, hi);
                        A[t] = B[t];
                        B[t] = hi;
                    } else {
                        const V4 ce = ldq(z, ch_zb, inpl_t);
                        r = r + w.wz * adj_axis<0, T, V>(gz, g.nzg, A[t], ce, zero);
                        r = r + w.wz * adj_axis<1, T, V>(gz, g.nzg, zero, B[t], hi);
                        A[t] = ce;
                        B[t] = hi;
                    }
                }
                // ---------------- time ------------------------------------------------------------
                if (g.ta) {
                    V4 hi = zero;       // frame t+1 of the forward-looking (or central) channel
                    if ((S == DOWNWIND || HY || t_cen) && t + 1 < M) hi = ldq(z, ch_tf, inpl_t + g.s_t);
                    if (S == UPWIND || (S == CENTRAL && !t_cen)) {
                        const V4 ce = (S == CENTRAL) ? t_c1 : ldq(z, ch_tb, inpl_t);
                        rt = rt + w.wt * adj_axis<0, T, V>(t, M, t_lo, ce, zero);
                        t_lo = ce;
                        if (S == CENTRAL) t_c1 = (t + 1 < M) ? ldq(z, ch_tb, inpl_t + g.s_t) : zero;
                    } else if (S == DOWNWIND) {
                        rt = rt + w.wt * adj_axis<1, T, V>(t, M, zero, t_ce, hi);
                        t_ce = hi;
                    } else if (S == CENTRAL) {
                        rt = rt + w.wt * adj_axis<2, T, V>(t, M, t_lo, zero, hi);
                        t_lo = t_c1;
                        t_c1 = hi;
                    } else {
                        const V4 ce = ldq(z, ch_tb, inpl_t);
                        rt = rt + w.wt * adj_axis<0, T, V>(t, M, t_lo, ce, zero);
                        rt = rt + w.wt * adj_axis<1, T, V>(t, M, zero, t_ce, hi);
                        t_lo = ce;
                        t_ce = hi;
                    }
                    r = r + rt * mf;
                }
                if (S == HYBRID) r = Consts<T>::inv_sqrt2() * r;
                if (S == CENTRAL) r = T(0.5) * r;
                acc += epi((long long)z * g.s_z + inpl_t, r);
            }
        }
    }
    if (Epi::REDUCES) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) epi.partials[linear_block_id()] = acc;
    }
}

}  // namespace tv
