// tv_nstream.h -- out = (I + rho D^T D) x from x alone as a STREAMING kernel (tv_normal_op: the inner operator of the ADMM
// x-update's conjugate-gradient loop), radius-1 schemes (upwind / downwind / hybrid share D^T D = sum_a w_a^2 (bwd_a -
// fwd_a) as long as the time weight does not vary along t), fp32, 16-byte lanes, any number of frames (windows of 8).
//
// Same structure as k_D_stream (tv_dstream.h): a wave covers 4 rows x 16 lanes, row neighbours are 16-lane shuffles,
// column neighbours one-lane DPP shifts, halo rows / edge elements predicated loads; no LDS tile, no barrier; x is read
// once, every load is issued one plane ahead of its use.  State: planes z-1, z, z+1 of the M frames in registers.
//
// Modes (NormalArgs): b == nullptr:  out = A x,  dots = { <x, out>, <x, x> }
//                     b != nullptr:  out = b - A x (and out2 = out when given: r and the first search direction of CG),
//                                    dots = { <out, out>, <x, x> }
#pragma once
#include "tv_device.h"
#include "tv_stencil.h"
#include "tv_fused.h"
#include "tv_dstream.h"

namespace tv {

struct NormalArgs {
    const float* x;
    const float* xp;       // TWO planes z0-2, z0-1 (or nullptr)
    const float* xn;       // TWO planes z0+nz, z0+nz+1 (or nullptr)
    const float* b;        // or nullptr
    float* out;
    float* out2;           // or nullptr
    float rho;
    double* part0;         // per-block partials of the first / second dot product
    double* part1;
};

constexpr int NS_TWN = 8;

template <int M, bool TWIN>
__global__ __launch_bounds__(256, M >= 6 ? 2 : 3) void k_normal_stream(DG g, WT<float> w, NormalArgs a, int zchunk, int nchunks) {
    __shared__ double sm[16];
    const int lane = (int)threadIdx.x, wave = (int)threadIdx.y;
    const int row = lane >> 4, lx = lane & 15;
    const int nxv = g.nx / 4;
    const int tiles_x = (nxv + 63) / 64, tiles_y = (g.ny + 3) / 4;
    const int Mg = TWIN ? g.m : M;
    const int nwin = TWIN ? (Mg + NS_TWN - 1) / NS_TWN : 1;
    const long long ntiles = (long long)tiles_x * tiles_y, total = ntiles * nchunks * nwin, per_xcd = (total + 7) / 8;
    const long long lid = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;       // XCD-aware order (tv_dstream.h)
    double acc0 = 0.0, acc1 = 0.0;
    if (lid < total) {
        const int win = (int)(lid / (ntiles * nchunks));
        const int chunk = (int)((lid / ntiles) % nchunks), tile = (int)(lid % ntiles);
        const int t0 = TWIN ? win * NS_TWN : 0;
        const int bx = tile % tiles_x, by = tile / tiles_x;
        const int col0 = (bx * 64 + wave * 16 + lx) * 4, y = by * 4 + row;
        const bool ok = (col0 < g.nx) && (y < g.ny);
        const unsigned voff = ok ? (unsigned)(((long long)y * g.nx + col0) * 4) : 0u;
        const unsigned row_bytes = (unsigned)g.nx * 4u;
        const int zs = chunk * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        const F4 zero = vsplat<float, 4>(0.f);
        F4 mf2 = vsplat<float, 4>(1.f);
        if (g.ta) {
            const F4 mf = mask_factor<float, 4>(g, w.sf, ok ? y : 0, ok ? col0 : 0);
            mf2 = (w.wt * w.wt) * (mf * mf);
        }
        const float wz2 = g.za ? w.wz * w.wz : 0.f;
        // existence of the in-plane neighbours as multipliers (no branches around vectors in the frame loop)
        const float m_pr = (ok && y > 0) ? 1.f : 0.f, m_nr = (ok && y + 1 < g.ny) ? 1.f : 0.f;
        const float m_c0 = (ok && col0 > 0) ? 1.f : 0.f, m_c3 = (ok && col0 + 4 < g.nx) ? 1.f : 0.f;
        const bool want_up = (row == 0) && ok && (y > 0), want_dn = (row == 3) && ok && (y + 1 < g.ny);
        const unsigned hoff = want_up ? voff - row_bytes : voff + row_bytes;
        // a tile row needs BOTH halo rows when the wave tile is a single row high at the frame border: rows 0 and 3 differ,
        // so one predicated load per lane is enough (row 0 reads y-1, row 3 reads y+1)
        const bool want_le = (lx == 0) && ok && (col0 > 0), want_re = (lx == 15) && ok && (col0 + 4 < g.nx);
        const unsigned eoff = want_le ? voff - 4u : voff + 16u;
        auto fvalid = [&](int t) { return !TWIN || (t0 + t < Mg); };
        auto foff = [&](int t) { return (long long)(t0 + t) * g.s_t; };
        F4 C[M], P[M], N[M], H[M];
        float E[M];
        auto plane = [&](int zl) { return g.za ? zplane<float>(g, a.x, a.xp, a.xn, 2, zl) : ((zl >= 0 && zl < g.nz) ? a.x + (long long)zl * g.s_z : nullptr); };
        auto load_c = [&](const float* pl, int t) { return (pl != nullptr && ok && fvalid(t)) ? ldu(pl + foff(t), voff) : zero; };
        {
            const float* pp = g.za ? plane(zs - 1) : nullptr;
            const float* pc = plane(zs);
            const float* pn = (g.za || zs + 1 < ze) ? plane(zs + 1) : nullptr;        // the next centre plane even without a z axis
#pragma unroll
            for (int t = 0; t < M; ++t) {
                P[t] = load_c(pp, t);
                C[t] = load_c(pc, t);
                N[t] = load_c(pn, t);
                H[t] = (pc != nullptr && (want_up || want_dn) && fvalid(t)) ? ldu(pc + foff(t), hoff) : zero;
                E[t] = (pc != nullptr && (want_le || want_re) && fvalid(t)) ? ldu1(pc + foff(t), eoff) : 0.f;
            }
        }
        for (int z = zs; z < ze; ++z) {
            const int gz = g.z0 + z;
            const float m_pz = (g.za && gz > 0) ? wz2 : 0.f, m_nz = (g.za && gz + 1 < g.nzg) ? wz2 : 0.f;
            const float* pc = plane(z);
            const float* pc1 = (z + 1 < ze) ? plane(z + 1) : nullptr;             // centre plane of the next step (halo rows, edges)
            const float* pn2 = (z + 1 < ze && (g.za || z + 2 < ze)) ? plane(z + 2) : nullptr;      // its next plane
            F4 cold = zero;
            if (TWIN && g.ta && t0 > 0) cold = ok ? ldu(pc + foff(-1), voff) : zero;
#pragma unroll
            for (int t = 0; t < M; ++t) {
                if (TWIN && !fvalid(t)) break;
                const int tg = t0 + t;
                const F4 c = C[t], h = H[t];
                // ---- - Laplacian-like sum: (c - prev) - (next - c) per axis, missing neighbours drop their term ----------
                const F4 sdn = shfl_down16(c), sup = shfl_up16(c);
                const F4 nr = (row == 3) ? h : sdn, pr = (row == 0) ? h : sup;
                F4 r = m_pr * (c - pr) - m_nr * (nr - c);
                {
                    // the cross-lane moves are executed by EVERY lane (a DPP read from a lane that a branch has switched
                    // off returns 0), the select comes afterwards
                    const float from_l = dpp_from_left(c.v[3]), from_r = dpp_from_right(c.v[0]);
                    const float left = (lx == 0) ? E[t] : from_l;
                    const float right = (lx == 15) ? E[t] : from_r;
                    const float e0 = c.v[1] - c.v[0], e1 = c.v[2] - c.v[1], e2 = c.v[3] - c.v[2];
                    // interior elements of the vector always have both column neighbours inside the frame (nx % 4 == 0)
                    r.v[0] += m_c0 * (c.v[0] - left) - e0;
                    r.v[1] += e0 - e1;
                    r.v[2] += e1 - e2;
                    r.v[3] += e2 - m_c3 * (right - c.v[3]);
                }
                r = r + (m_pz * (c - P[t]) - m_nz * (N[t] - c));
                if (g.ta) {
                    F4 tt = zero;
                    if (tg > 0) tt = tt + (c - cold);
                    if (t + 1 < M) { if (tg + 1 < Mg) tt = tt - (C[(t + 1 < M) ? t + 1 : t] - c); }
                    else if (TWIN && tg + 1 < Mg) tt = tt - ((ok ? ldu(pc + foff(t + 1), voff) : zero) - c);
                    r = r + mf2 * tt;
                }
                // ---- rotate the planes, request the next ones (before the stores of this frame) ----------------------------
                cold = c;
                P[t] = c;
                C[t] = N[t];
                N[t] = load_c(pn2, t);
                H[t] = (pc1 != nullptr && (want_up || want_dn) && fvalid(t)) ? ldu(pc1 + foff(t), hoff) : zero;
                E[t] = (pc1 != nullptr && (want_le || want_re) && fvalid(t)) ? ldu1(pc1 + foff(t), eoff) : 0.f;
                // ---- epilogue ---------------------------------------------------------------------------------------------
                if (!ok) continue;
                const long long fo = (long long)z * g.s_z + foff(t);
                F4 o;
                if (a.b == nullptr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o.v[i] = c.v[i] + a.rho * r.v[i];
                        acc0 += (double)c.v[i] * (double)o.v[i];
                        acc1 += (double)c.v[i] * (double)c.v[i];
                    }
                } else {
                    const F4 bv = ldu(a.b + fo, voff);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        o.v[i] = bv.v[i] - (c.v[i] + a.rho * r.v[i]);
                        acc0 += (double)o.v[i] * (double)o.v[i];
                        acc1 += (double)c.v[i] * (double)c.v[i];
                    }
                    if (a.out2 != nullptr) stu(a.out2 + fo, voff, o);
                }
                stu(a.out + fo, voff, o);
            }
        }
    }
    acc0 = block_sum(acc0, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part0[blockIdx.x] = acc0;
    acc1 = block_sum(acc1, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part1[blockIdx.x] = acc1;
}

}  // namespace tv
