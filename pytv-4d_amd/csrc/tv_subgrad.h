// tv_subgrad.h -- ONE-PASS TV value + sub-gradient (pytv/tv_GPU.py:47-375 of the reference) for the radius-1
// schemes (upwind, downwind, hybrid), fp32, 16-byte lanes.
//
// The two-pass form (tv_subgrad: 1/|Dx| to memory, then a gather over x and 1/|Dx|) moves >= 5 words per
// voxel for 2 words of algorithmic traffic.  Here 1/|Dx| never leaves the chip.  With d_f / d_b the forward /
// backward gradient channels of a site and n = 1/|Dx| (0 where |Dx| = 0), the gather of tv_stencil.h
//     G(v) = s * sum_a [ up:   d_b(v) n(v-e_a) - d_f(v) n(v)   ]  +  [ down: d_b(v) n(v) - d_f(v) n(v+e_a) ]
// is rewritten with d_b(v) = d_f(v - e_a) as a SCATTER of the per-site products PF = d_f n, PB = d_b n:
//     G(v) = s * sum_a [ up:   PF_a(v-e_a) - PF_a(v) ]  +  [ down: PB_a(v) - PB_a(v+e_a) ]
// so a site needs its neighbours' PRODUCTS, not their norms, and nothing is recomputed:
//
//   thread = one (row, 4-col) site, marching z inside a z-chunk, the M frames unrolled in registers;
//   step z computes n(z), PF(z), PB(z) from x(z-1), x(z), x(z+1) and adds them to three accumulators:
//     G(z-1) -= PB_z(z)   (now complete: stored),   G(z) += in-plane / time / own terms,   G(z+1) := PF_z(z)
//   in-plane neighbours: a wave covers 4 rows x 16 lanes (64 columns): columns by a one-lane shuffle, rows by a
//   16-lane shuffle; the first / last row of a wave hands its row product to the neighbouring wave of the block
//   through a small LDS buffer (two barriers per plane).
//   A block (NW waves stacked in y) computes n and the products on its whole (4 NW rows) x (16 lanes) tile but
//   stores G only for the inner (4 NW - 2) x 14 sites: the outer ring is the overlap with the neighbouring
//   blocks (its x comes from L2 mostly), which is what makes this a pure function of x -- no fix-up pass, no
//   inter-block communication.  A z-chunk recomputes one plane of n on either side for the same reason.
#pragma once
#include "tv_device.h"
#include "tv_stencil.h"
#include "tv_fused.h"

#ifndef SG_X
#define SG_X 0
#endif

namespace tv {

// Opaque use of a vector: stops LLVM from SINKING the accumulation chains of a frame below the barrier that follows the
// frame loop (it does: the sums are only consumed there, and the ~14 product vectors of every frame then stay live
// to the end of the loop -- 70 VGPRs per frame instead of 20).
__device__ __forceinline__ void pin(F4& a) { asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3])); }

template <int S, int M, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_subgrad_one(DG g, WT<float> w, const float* __restrict__ x,
                                                            const float* __restrict__ xp, const float* __restrict__ xn,
                                                            float* __restrict__ G, int zchunk, double* __restrict__ partials) {
    static_assert(S != CENTRAL, "radius-2 scheme");
    constexpr bool UP = (S == UPWIND || S == HYBRID), DN = (S == DOWNWIND || S == HYBRID);
    constexpr bool HALO = (S == HYBRID);   // only the hybrid norm of a ring row looks at the row outside the tile
    constexpr int RB = 4 * NW, UR = RB - 2, UC = 14;
    __shared__ F4 xe[M][NW][2][16];      // x of the first / last row of every wave (cross-wave row neighbours)
    __shared__ F4 ye[M][NW][2][16];      // [0]: PB_r of the first row (for the wave above), [1]: PF_r of the last row
    __shared__ F4 lds_P[M][64 * NW];     // x(z-1) of every site: per-thread slots (registers are the scarce resource)
    __shared__ double sm[16];
    const int lane = (int)threadIdx.x, wv = (int)threadIdx.y;
    const int tid = wv * 64 + lane;
    const int rr = lane >> 4, lx = lane & 15, ry = wv * 4 + rr;
    const int nxv = g.nx / 4;
    const int tiles_x = (nxv + UC - 1) / UC;
    const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
    const int cv = bx * UC - 1 + lx, y = by * UR - 1 + ry, col0 = cv * 4;
    const bool in = (cv >= 0) && (cv < nxv) && (y >= 0) && (y < g.ny);               // site inside the frame
    const bool useful = in && (ry >= 1) && (ry <= RB - 2) && (lx >= 1) && (lx <= 14);
    const unsigned voff = in ? (unsigned)(((long long)y * g.nx + col0) * 4) : 0u;    // byte offset inside a frame (< 2^32: host)
    const int zs = (int)blockIdx.y * zchunk;
    const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
    const F4 zero = vsplat<float, 4>(0.f);
    const F4 mf = (g.ta && in) ? mask_factor<float, 4>(g, w.sf, y, col0) : vsplat<float, 4>(1.f);
    const float s = (S == HYBRID) ? Consts<float>::inv_sqrt2() : 1.f;
    // frame-border masks as multipliers (straight-line code: every `if` around a vector costs registers here)
    const float m_pr = (in && y > 0) ? 1.f : 0.f, m_nr = (in && y + 1 < g.ny) ? 1.f : 0.f;
    const float m_c0 = (in && col0 > 0) ? 1.f : 0.f, m_c3 = (in && col0 + 3 < g.nx - 1) ? 1.f : 0.f;
    const float m_in = in ? 1.f : 0.f;
    const float wt = g.ta ? w.wt : 0.f;
    // ring rows of a hybrid tile read the row just outside the tile from memory
    const bool ring_up = HALO && (ry == 0) && in && (y > 0), ring_dn = HALO && (ry == RB - 1) && in && (y + 1 < g.ny);
    const unsigned hoff = ring_up ? voff - (unsigned)g.nx * 4u : voff + (unsigned)g.nx * 4u;
    // cross-wave row neighbours: slot of the wave above / below (clamped), and whether this lane takes them
    const int w_up = (wv > 0) ? wv - 1 : 0, w_dn = (wv < NW - 1) ? wv + 1 : NW - 1;
    const bool take_up = (rr == 0), take_dn = (rr == 3);
    const float m_xup = (rr == 0 && wv > 0) ? 1.f : 0.f, m_xdn = (rr == 3 && wv < NW - 1) ? 1.f : 0.f;
    const float m_iup = (rr > 0) ? 1.f : 0.f, m_idn = (rr < 3) ? 1.f : 0.f;
    double acc = 0.0;

    F4 C[M], Gp[M], Gc[M];
    {
        const float* pp = g.za ? zplane<float>(g, x, xp, xn, 2, zs - 2) : nullptr;
        const float* pc = zplane<float>(g, x, xp, xn, 2, g.za ? zs - 1 : zs);
#pragma unroll
        for (int t = 0; t < M; ++t) {
            lds_P[t][tid] = (in && pp) ? ldu(pp + (long long)t * g.s_t, voff) : zero;
            C[t] = (in && pc) ? ldu(pc + (long long)t * g.s_t, voff) : zero;
            Gp[t] = zero;
            Gc[t] = zero;
        }
    }
    // no z axis: planes are independent and the chunk needs no extra plane on either side
    const int z_lo = g.za ? zs - 1 : zs;
    for (int zl = z_lo; zl <= ze; ++zl) {
        const int gz = g.z0 + zl;
        const bool plane_in = (gz >= 0) && (gz < g.nzg) && (zl < ze || g.za);
        const float* pc = zplane<float>(g, x, xp, xn, 2, zl);
        const float* pn = zplane<float>(g, x, xp, xn, 2, zl + 1);
        const bool want_next = (pn != nullptr) && in && (g.za || zl + 1 < ze);
        F4 N[M];
#pragma unroll
        for (int t = 0; t < M; ++t) N[t] = want_next ? ldu(pn + (long long)t * g.s_t, voff) : zero;
#pragma unroll
        for (int t = 0; t < M; ++t)
            if (rr == 0 || rr == 3) xe[t][wv][rr == 3 ? 1 : 0][lx] = C[t];
        __syncthreads();
        // uniform per step: z weights (0 when the neighbour plane does not exist), validity of this plane
        const float wzp = (plane_in && g.za && gz > 0) ? w.wz : 0.f, wzn = (plane_in && g.za && gz + 1 < g.nzg) ? w.wz : 0.f;
        const float m_site = plane_in ? m_in : 0.f;                  // 1/|Dx| of a site that does not exist is 0
        const bool count = useful && plane_in && (zl >= zs) && (zl < ze);
        const bool store = useful && (zl - 1 >= zs) && (zl - 1 < ze);
        const bool halo_here = (ring_up || ring_dn) && (pc != nullptr);
        F4 pf_t_prev = zero;                 // PF_t of frame t-1 (up part: added to frame t)
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const F4 c = C[t];
            // ---- neighbourhood of x(zl, t) ------------------------------------------------------------
            F4 xu = (SG_X & 64) ? c : shfl_up16(c), xd = (SG_X & 64) ? c : shfl_down16(c);
            {
                F4 h = zero;
                if (HALO && halo_here) h = ldu(pc + (long long)t * g.s_t, hoff);
                const F4 eu = xe[t][w_up][1][lx], ed = xe[t][w_dn][0][lx];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xu.v[i] = take_up ? ((wv > 0) ? eu.v[i] : h.v[i]) : xu.v[i];
                    xd.v[i] = take_dn ? ((wv < NW - 1) ? ed.v[i] : h.v[i]) : xd.v[i];
                }
            }
            const float xl = __shfl_up(c.v[3], 1, 64), xr = __shfl_down(c.v[0], 1, 64);   // ring lanes: don't care
            // ---- gradient channels (same arithmetic as d_slots / subgrad_site) -------------------------
            F4 f_r = m_nr * (xd - c), b_r = m_pr * (c - xu), f_c, b_c;
            f_c.v[0] = c.v[1] - c.v[0]; f_c.v[1] = c.v[2] - c.v[1]; f_c.v[2] = c.v[3] - c.v[2]; f_c.v[3] = m_c3 * (xr - c.v[3]);
            b_c.v[0] = m_c0 * (c.v[0] - xl); b_c.v[1] = f_c.v[0]; b_c.v[2] = f_c.v[1]; b_c.v[3] = f_c.v[2];
            F4 f_z = wzn * (N[t] - c), b_z = wzp * (c - lds_P[t][tid]);
            F4 f_t = zero, b_t = zero;
            if (t + 1 < M) f_t = (wt * (C[(t + 1 < M) ? t + 1 : t] - c)) * mf;
            if (t > 0) b_t = (wt * (c - C[(t > 0) ? t - 1 : 0])) * mf;
            if (S == HYBRID) {
                f_r = s * f_r; f_c = s * f_c; b_r = s * b_r; b_c = s * b_c;
                f_z = s * f_z; b_z = s * b_z; f_t = s * f_t; b_t = s * b_t;
            }
            F4 ss = zero;
            if (S == HYBRID) ss = ((((((f_r * f_r + f_c * f_c) + b_r * b_r) + b_c * b_c) + f_z * f_z) + b_z * b_z) + f_t * f_t) + b_t * b_t;
            if (S == UPWIND) ss = ((f_r * f_r + f_c * f_c) + f_z * f_z) + f_t * f_t;
            if (S == DOWNWIND) ss = ((b_r * b_r + b_c * b_c) + b_z * b_z) + b_t * b_t;
            F4 n, rn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rn.v[i] = (SG_X & 32) ? __builtin_amdgcn_sqrtf(ss.v[i]) : tsqrt(ss.v[i]);
                n.v[i] = (SG_X & 32) ? m_site * __builtin_amdgcn_rcpf(rn.v[i]) : ((rn.v[i] > tiny_norm<float>()) ? m_site / rn.v[i] : 0.f);
            }
            // four norms in fp32 (each carries its own 2^-24 already), then fp64 across frames / planes / threads
            if (count && !(SG_X & 4)) acc += (double)((rn.v[0] + rn.v[1]) + (rn.v[2] + rn.v[3]));
            // ---- scatter the products ---------------------------------------------------------------------
            F4 gc = Gc[t], gn = zero;
            if (UP) {
                const F4 pf_r = f_r * n, pf_c = f_c * n, pf_z = f_z * n, pf_t = f_t * n;
                const F4 from_up = shfl_up16(pf_r);                       // PF_r of the row above (same wave)
                const float from_left = __shfl_up(pf_c.v[3], 1, 64);
                if (rr == 3) ye[t][wv][1][lx] = pf_r;
                gc = gc + (m_iup * from_up - pf_r);
                gc = gc + (shift_right<float, 4>(pf_c, from_left) - pf_c);
                gc = gc - pf_z;
                gn = pf_z;
                gc = gc + (pf_t_prev - pf_t);
                pf_t_prev = pf_t;
            }
            if (DN) {
                const F4 pb_r = b_r * n, pb_c = b_c * n, pb_z = b_z * n, pb_t = b_t * n;
                const F4 from_dn = shfl_down16(pb_r);                     // PB_r of the row below (same wave)
                const float from_right = __shfl_down(pb_c.v[0], 1, 64);
                if (rr == 0) ye[t][wv][0][lx] = pb_r;
                gc = gc + (pb_r - m_idn * from_dn);
                gc = gc + (pb_c - shift_left<float, 4>(pb_c, from_right));
                gc = gc + pb_z;
                Gp[t] = Gp[t] - pb_z;
                gc = gc + pb_t;
                if (t > 0) { Gc[t - 1] = Gc[t - 1] - pb_t; pin(Gc[t - 1]); }
            }
            pin(gc);
            Gc[t] = gc;
            // ---- plane zl-1 is complete ------------------------------------------------------------------
            if (store) {
                F4 o = Gp[t];
                if (S == HYBRID) o = s * o;
                stu(G + (long long)(zl - 1) * g.s_z + (long long)t * g.s_t, voff, o);
            }
            pin(gn);
            Gp[t] = gn;                      // the slot of the finished plane now carries the start of G(zl+1)
            if (!(SG_X & 128)) __builtin_amdgcn_sched_barrier(0);   // keep the frames apart (register pressure)
        }
        __syncthreads();
        // cross-wave row products, rotation of the planes
#pragma unroll
        for (int t = 0; t < M; ++t) {
            F4 gc = Gc[t];
            if (UP) gc = gc + m_xup * ye[t][w_up][1][lx];
            if (DN) gc = gc - m_xdn * ye[t][w_dn][0][lx];
            Gc[t] = Gp[t];
            Gp[t] = gc;
            lds_P[t][tid] = C[t];
            C[t] = N[t];
        }
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[linear_block_id()] = acc;
}

}  // namespace tv
