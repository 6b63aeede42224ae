// tv_dstream.h -- d = D x as a pure STREAMING kernel (tv_D: pytv/tv_operators_GPU.py:134-581 of the reference), fp32,
// 16-byte lanes, M <= 8 frames (more frames: windows of 8).
//
// tv_D moves 1 word in and Nd words out per voxel: it is a store kernel.  What limited the earlier kernels was not
// bandwidth but the ORDER of memory operations inside a wave.  gfx950 has ONE in-order counter (vmcnt) for vector
// loads and stores: a load whose result is needed while older stores are still in flight can only be waited for with
// a count that also covers those stores.  The one-site-per-thread kernel re-reads x from beyond L2 (11.9 GB for a
// 2.1 GB image); the marching kernel k_D_march reads x once but issues every load BEHIND the previous frame's Nd stores
// and consumes it at once (s_waitcnt vmcnt(0): one exposed store round trip per frame) and drains the queue again at its
// per-plane __syncthreads().  Here
//   * no LDS tile, no barrier: a wave covers 4 rows x 16 lanes (64 columns), row neighbours are 16-lane shuffles, column
//     neighbours one-lane DPP shifts; the first / last row of the wave tile reads one halo row, the first / last lane of
//     a row one edge element -- both predicated loads;
//   * every load is issued ONE PLANE ahead of its use, and before the stores of the current frame in program order: the
//     data is in flight for M frames of work and the wait for it never covers a younger store;
//   * the z difference is shared: s wz (x(z) - x(z-1)) is the backward channel of plane z AND the forward channel of
//     plane z-1, so step z stores it to both places and x(z+1) is never needed inside the step: state = 2 M vectors.
// central (radius 2 in z) keeps three planes and writes its z channel one plane behind.
#pragma once
#include "tv_device.h"
#include "tv_stencil.h"
#include "tv_fused.h"

#ifndef TV_DSTREAM_NT
#define TV_DSTREAM_NT 1          // the gradient array is stored non-temporally (hybrid tv_D 3.98 -> 3.87 ms, central 2.44 - 2.66 -> 2.21 - 2.41; 0: plain stores)
#endif
namespace tv {
template <typename T, int V> __device__ __forceinline__ void DSTU(T* ubase, unsigned voff, const Vec<T, V>& v) {
#if TV_DSTREAM_NT
    stu_s_t<T, V>(ubase, voff, v);
#else
    stu_t<T, V>(ubase, voff, v);
#endif
}


__device__ __forceinline__ float dpp_from_left(float v) {      // lane-1 inside the 16-lane row (lane 0 of a row: 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_from_right(float v) {     // lane+1 inside the 16-lane row (lane 15 of a row: 0)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101 /* row_shl:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ double dpp_from_left(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x111, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x111, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double dpp_from_right(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x101, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x101, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// the 16-lane row shuffles for lanes of V = 16 / sizeof(T) columns (ldu_t / ldu1_t / stu_t: tv_fused.h)
template <typename T, int V> __device__ __forceinline__ Vec<T, V> shfl_up16_t(const Vec<T, V>& v) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = __shfl_up(v.v[i], 16, 64);
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> shfl_down16_t(const Vec<T, V>& v) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = __shfl_down(v.v[i], 16, 64);
    return r;
}

#ifndef TV_TWN
#define TV_TWN 8                 // frames per time window of the streaming / one-sweep kernels.  EXPERIMENT -DTV_TWN=4 -DTV_WAVES=4: M = 8 volumes as two windows
#endif                           // of 4 frames, half the per-thread state, 16 waves per CU (round 4, verdict item 6; profiles/r4_window4_ab.txt)
#ifndef TV_WAVES
#define TV_WAVES 0               // 0: the launch bounds below; 4: at least 4 waves per SIMD (<= 128 VGPRs) for every instantiation
#endif
constexpr int DS_TWN = TV_TWN;      // frames per time window (M > DS_TWN)

// Block composition of the streaming kernels (k_D_stream, k_normal_stream*): ST_NWX waves side by side times ST_NWY waves
// stacked, each a 4-row x 16-lane wave tile -> block tile 4 ST_NWY rows x 256 ST_NWX bytes.  Waves of one block request
// their halo rows / border elements in the same frame as the owner of those lines requests them, so inside a block they
// cost no extra fetch; only the block's outline does: 2 rows per 4 ST_NWY, 2 x 64 B per 256 ST_NWX bytes of row.
#ifndef TV_STREAM_NWX
#define TV_STREAM_NWX 4
#endif
#ifndef TV_STREAM_NWY
#define TV_STREAM_NWY 1
#endif
// TV_STREAM_SYNC: 1 = the waves of a block meet at a bare s_barrier (no fence, no vmcnt drain) at the top of every plane,
// 2 = of every frame -- keeps them close enough in time for the shared lines to still be in L2 when the second wave asks
#ifndef TV_STREAM_SYNC
#define TV_STREAM_SYNC 2
#endif
__device__ __forceinline__ void st_sync_plane() { if (TV_STREAM_SYNC == 1) __builtin_amdgcn_s_barrier(); }
__device__ __forceinline__ void st_sync_frame() { if (TV_STREAM_SYNC >= 2) __builtin_amdgcn_s_barrier(); }
constexpr int ST_NWX = TV_STREAM_NWX, ST_NWY = TV_STREAM_NWY, ST_THREADS = 64 * ST_NWX * ST_NWY;
constexpr int ST_BCV = 16 * ST_NWX;      // 4-column vectors per block-tile row
constexpr int ST_BR = 4 * ST_NWY;        // rows per block tile

// block (64, 4): wave w covers columns [64 w, 64 w + 64) of a 4-row x 256-column block tile; grid.x = XCD-ordered
// (tile, chunk, window) ids.  TWIN: the block works on frames [t0, t0 + M) of a volume with more than 8 frames and reads
// the x frame on either side of its window for the time differences (M == DS_TWN).
template <int S, int M, bool TWIN, typename T = float>
__global__ __launch_bounds__(ST_THREADS, TV_WAVES ? TV_WAVES : ((S == CENTRAL || M >= 8) ? 2 : 3)) void k_D_stream(DG g, WT<T> w, const T* __restrict__ x,
                                                                       const T* __restrict__ xp, const T* __restrict__ xn,
                                                                       T* __restrict__ d, int zchunk, int nchunks) {
    constexpr int V = 16 / (int)sizeof(T);          // columns per lane (16-byte lanes): 4 floats / 2 doubles
    using VT = Vec<T, V>;
    constexpr bool UP = (S == UPWIND || S == HYBRID), DN = (S == DOWNWIND || S == HYBRID), CEN = (S == CENTRAL);
    constexpr bool NEXT = (S != DOWNWIND), PREV = (S != UPWIND);
    const int lane = (int)threadIdx.x, wave = (int)threadIdx.y;
    const int row = lane >> 4, lx = lane & 15;
    const int nxv = (g.nx + V - 1) / V;         // (a last lane with pad columns: ragged PITCHED rows only, tv_geom::row_pitch)
    const int tiles_x = (nxv + ST_BCV - 1) / ST_BCV, tiles_y = (g.ny + ST_BR - 1) / ST_BR;
    const int Mg = TWIN ? g.m : M;
    const int nwin = TWIN ? (Mg + DS_TWN - 1) / DS_TWN : 1;
    // XCD-aware order: consecutive workgroup ids go round-robin to the 8 XCDs; give every XCD a contiguous run of
    // (chunk, tile) ids so that vertically adjacent tiles (they share their halo rows) meet in the same L2
    const long long ntiles = (long long)tiles_x * tiles_y, total = ntiles * nchunks * nwin, per_xcd = (total + 7) / 8;
    const long long lid = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (lid >= total) return;
    const int win = (int)(lid / (ntiles * nchunks));
    const int chunk = (int)((lid / ntiles) % nchunks), tile = (int)(lid % ntiles);
    const int t0 = TWIN ? win * DS_TWN : 0;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int col0 = (bx * ST_BCV + (wave % ST_NWX) * 16 + lx) * V, y = (by * ST_NWY + wave / ST_NWX) * 4 + row;
    const bool ok = (col0 < g.nx) && (y < g.ny);
    const unsigned voff = ok ? (unsigned)(((long long)y * g.rp + col0) * (long long)sizeof(T)) : 0u;      // byte offset inside a frame
    const unsigned row_bytes = (unsigned)g.rp * (unsigned)sizeof(T);
    const int zs = chunk * zchunk;
    const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
    const VT zero = vsplat<T, V>(T(0));
    const VT mf = g.ta ? mask_factor<T, V>(g, w.sf, ok ? y : 0, ok ? col0 : 0) : vsplat<T, V>(T(1));
    const bool z_fwd = CEN && g.z_two;
    // halo row / edge element of this lane (one predicated load each per frame)
    const bool want_up = PREV && (row == 0) && ok && (y > 0);
    const bool want_dn = NEXT && (row == 3) && ok && (y + 1 < g.ny);
    const bool want_halo = want_up || want_dn;
    const unsigned hoff = want_up ? voff - row_bytes : voff + row_bytes;
    const bool want_le = PREV && (lx == 0) && ok && (col0 > 0);
    const bool want_re = NEXT && (lx == 15) && ok && (col0 + V < g.nx);
    const bool want_edge = want_le || want_re;
    const unsigned eoff = want_le ? voff - (unsigned)sizeof(T) : voff + 16u;

    auto fvalid = [&](int t) { return !TWIN || (t0 + t < Mg); };
    auto foff = [&](int t) { return (long long)(t0 + t) * g.s_t; };             // uniform
    // state: plane z (C), plane z-1 (P), central: plane z-2 (P2); halo rows and edge elements of plane z
    VT C[M], P[M], P2[CEN ? M : 1], H[M];
    T E[M];
    auto load_plane = [&](const T* pl, int t, bool with_halo, VT& c, VT& h, T& e) {
        const bool v = (pl != nullptr) && fvalid(t);
        c = (v && ok) ? ldu_t<T, V>(pl + foff(t), voff) : zero;
        h = (v && with_halo && want_halo) ? ldu_t<T, V>(pl + foff(t), hoff) : zero;
        e = (v && with_halo && want_edge) ? ldu1_t<T>(pl + foff(t), eoff) : T(0);
    };
    {
        const T* pc = zplane<T>(g, x, xp, xn, 1, zs);
        const T* pp = (g.za && (PREV || UP)) ? zplane<T>(g, x, xp, xn, 1, zs - 1) : nullptr;
#pragma unroll
        for (int t = 0; t < M; ++t) {
            load_plane(pc, t, true, C[t], H[t], E[t]);
            P[t] = (pp != nullptr && ok && fvalid(t)) ? ldu_t<T, V>(pp + foff(t), voff) : zero;
            if (CEN) P2[t] = zero;       // x(zs - 2) is never needed: the z channel of plane zs - 1 is the previous chunk's
        }
    }
    // the z channel of the chunk's last plane needs plane ze: one trailing step (forward and central z differences)
    const bool trail = g.za && (UP || CEN);
    const int z_last = trail ? ze : ze - 1;
    for (int z = zs; z <= z_last; ++z) {
        st_sync_plane();
        const int gz = g.z0 + z;
        const bool in_chunk = (z < ze);
        const bool plane_here = (gz < g.nzg);                                 // plane z exists in the volume
        const bool plane_prev = (gz > 0);
        // plane to request now (consumed at step z + 1)
        const bool need_next = (z + 1 <= z_last);
        const T* pn = need_next ? zplane<T>(g, x, xp, xn, 1, z + 1) : nullptr;
        T* dz_cur = d + (long long)z * g.s_dz;                             // gradient plane z (uniform)
        T* dz_prv = d + (long long)(z - 1) * g.s_dz;
        VT cold = zero;          // x(z, t-1)
        if (TWIN && in_chunk && g.ta && t0 > 0 && PREV) {                     // frame left of the window (backward / central in t)
            const T* pc = zplane<T>(g, x, xp, xn, 1, z);
            cold = ok ? ldu_t<T, V>(pc + foff(-1), voff) : zero;
        }
#pragma unroll
        for (int t = 0; t < M; ++t) {
            if (TWIN && !fvalid(t)) break;                                    // ragged last window (block-uniform)
            st_sync_frame();
            const int tg = t0 + t;
            const VT c = C[t];
            // ---- z differences of this step ---------------------------------------------------------------------
            // radius-1 schemes: dzv = wz (x(z) - x(z-1)) is D_down(z) and D_up(z-1); central: wz (x(z) - x(z-2)) is 2 D(z-1)
            VT dzv = zero;
            if (g.za) {
                if (!CEN || z_fwd) { if (plane_here && plane_prev) dzv = w.wz * (c - P[t]); }
                else if (plane_here && gz >= 2) dzv = w.wz * (c - P2[t]);
            }
            // ---- in-plane and time channels of plane z -------------------------------------------------------------
            XN<T, V> n;
            n.c = c;
            n.col0 = col0;
            n.nr = n.pr = n.nc = n.pc = n.nz = n.pz = n.nt = n.pt = zero;
            n.h_nr = n.h_pr = n.h_nz = n.h_pz = n.h_nt = n.h_pt = false;
            if (in_chunk) {
                const VT h = H[t];
                if (NEXT) {
                    n.h_nr = ok && (y + 1 < g.ny);
                    const VT sdn = shfl_down16_t<T, V>(c);
                    n.nr = (row == 3) ? h : sdn;
                    const T sh = dpp_from_right(c.v[0]);
                    n.nc = shift_left<T, V>(c, (lx == 15) ? E[t] : sh);
                    if (t + 1 < M) { n.h_nt = (g.ta != 0) && (tg + 1 < Mg); n.nt = C[(t + 1 < M) ? t + 1 : t]; }
                    else if (TWIN && g.ta && tg + 1 < Mg) {                    // frame right of the window
                        const T* pc = zplane<T>(g, x, xp, xn, 1, z);
                        n.h_nt = true;
                        n.nt = ok ? ldu_t<T, V>(pc + foff(t + 1), voff) : zero;
                    }
                }
                if (PREV) {
                    n.h_pr = ok && (y > 0);
                    const VT sup = shfl_up16_t<T, V>(c);
                    n.pr = (row == 0) ? h : sup;
                    const T sh = dpp_from_left(c.v[V - 1]);
                    n.pc = shift_right<T, V>(c, (lx == 0) ? E[t] : sh);
                    if (tg > 0) { n.h_pt = (g.ta != 0); n.pt = cold; }
                }
            }
            VT o[8];
            VT mft = mf;          // per-voxel weight on the time channels (tv_geom::time_weight_vol): one more streamed read
            if (g.wv != nullptr && g.ta && in_chunk && ok) mft = mf * ldu_t<T, V>(static_cast<const T*>(g.wv) + (long long)z * g.s_z + foff(t), voff);
            d_slots<S, T, V>(g, w, n, mft, o);       // z slots come out as zero (h_nz = h_pz = false): filled below
            // ---- next plane: requested before this frame's stores -----------------------------------------------------
            cold = c;
            if (CEN) P2[t] = P[t];
            P[t] = c;
            if (need_next) load_plane(pn, t, z + 1 < ze, C[t], H[t], E[t]);      // the trailing step needs no halo
            // ---- stores -------------------------------------------------------------------------------------------
            if (!ok) continue;
            const long long fo = foff(t);
            if (S == HYBRID) {
                const VT dzs = Consts<T>::inv_sqrt2() * dzv;
                if (in_chunk) {
                    DSTU<T, V>(dz_cur + fo, voff, o[0]);
                    DSTU<T, V>(dz_cur + 1 * g.s_z + fo, voff, o[1]);
                    DSTU<T, V>(dz_cur + 2 * g.s_z + fo, voff, o[2]);
                    DSTU<T, V>(dz_cur + 3 * g.s_z + fo, voff, o[3]);
                    if (g.za) DSTU<T, V>(dz_cur + (long long)(g.ch_z + 1) * g.s_z + fo, voff, dzs);
                    if (g.ta) {
                        DSTU<T, V>(dz_cur + (long long)g.ch_t * g.s_z + fo, voff, o[6]);
                        DSTU<T, V>(dz_cur + (long long)(g.ch_t + 1) * g.s_z + fo, voff, o[7]);
                    }
                }
                if (g.za && z > zs) DSTU<T, V>(dz_prv + (long long)g.ch_z * g.s_z + fo, voff, dzs);
            } else {
                if (in_chunk) {
                    DSTU<T, V>(dz_cur + fo, voff, o[0]);
                    DSTU<T, V>(dz_cur + 1 * g.s_z + fo, voff, o[1]);
                    if (g.ta) DSTU<T, V>(dz_cur + (long long)g.ch_t * g.s_z + fo, voff, o[3]);
                    if (S == DOWNWIND && g.za) DSTU<T, V>(dz_cur + (long long)g.ch_z * g.s_z + fo, voff, dzv);
                    if (CEN && z_fwd && g.za) {
                        // two-plane volume: forward stencil, D(0) = 1/2 wz (x(1) - x(0)), D(1) = 0
                        if (gz == g.nzg - 1) DSTU<T, V>(dz_cur + (long long)g.ch_z * g.s_z + fo, voff, zero);
                    }
                }
                if (g.za && z > zs) {
                    if (S == UPWIND) DSTU<T, V>(dz_prv + (long long)g.ch_z * g.s_z + fo, voff, dzv);
                    if (CEN) {
                        // z channel of plane z-1: central 1/2 wz (x(z) - x(z-2)) on interior planes, 0 on the first / last one
                        VT cz = T(0.5) * dzv;
                        if (!z_fwd && !(plane_here && gz >= 2)) cz = zero;
                        DSTU<T, V>(dz_prv + (long long)g.ch_z * g.s_z + fo, voff, cz);
                    }
                }
            }
        }
    }
}

}  // namespace tv
