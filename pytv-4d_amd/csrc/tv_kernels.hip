// tv_kernels.hip -- one-site-per-thread HIP kernels (gfx950 / CDNA4) for the PyTV-4D hot path and most of the
// C-ABI declared in include/pytv4d.h.  (Plane-marching kernels: tv_march.h + tv_march_D/DT.hip; the one-sweep
// Chambolle-Pock iteration and the marching sub-gradient gather: tv_fused.h + tv_fused.hip.)
//
//   k_D      : forward operator D (all four schemes) with a fused epilogue (tv_stencil.h)
//                StoreD   -> materialise the gradient            (tv_D)
//                NormEpi  -> 1/|Dx|_2 per voxel + TV partials    (tv_subgrad pass 1)
//                CpDual   -> q <- proj(q + sigma D x)            (tv_cp_dual)
//                AdmmZU   -> z/u update of ADMM                  (tv_admm_zu)
//   k_DT     : transposed operator as a GATHER (no atomics, no scratch time buffer) with
//                StoreDT / AxpyDT / CpPrimal epilogues           (tv_DT, tv_DT_axpy, tv_cp_primal)
//   k_subgrad_vec / k_subgrad_central_vec : sub-gradient from x and 1/|Dx| (tv_subgrad pass 2)
//   k_normal_vec  / k_normal_central_vec  : x + rho D^T D x from x alone   (tv_normal_op)
//   k_gather : scalar reference evaluation of the last two (TV_SCALAR_GATHER=1)
//   k_l21, k_dot, k_sub, k_cg1, k_cg2, k_sgstep, k_reduce        small streaming / reduction kernels
//
// Per-voxel definitions follow SURVEY 8a-1 / 8a-2, i.e. pytv/tv_operators_CPU.py:117-154,198-218,
// 264-284,330-358 (D) and :398-448,487-516,554-583,622-658 (D^T); the sub-gradient follows
// pytv/tv_CPU.py:91-126,176-190,239-253,302-330.
#include <hip/hip_runtime.h>

#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_site.h"

namespace tv {

// =============================================================================================
// forward kernel
// =============================================================================================
template <int S, typename T, int V, typename Epi>
__global__ __launch_bounds__(256) void k_D(DG g, WT<T> w, const T* x, const T* xp, const T* xn, int hp, int z_first, Epi epi) {
    __shared__ double sm[16];
    Coord c = thread_coord<V>(g, z_first);
    double acc = d_site<S, T, V>(g, w, x, xp, xn, hp, c, epi, PlainMem());
    if (Epi::REDUCES) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) epi.partials[linear_block_id()] = acc;
    }
}

template <int S, typename T, int V, typename Src, typename Epi>
__global__ __launch_bounds__(256) void k_DT(DG g, WT<T> w, Src src, Epi epi) {
    __shared__ double sm[16];
    const Coord c = thread_coord<V>(g, 0);
    double acc = 0.0;
    if (c.ok) {
        long long inpl;
        const Vec<T, V> r = dt_site<S, T, V>(g, w, src, c, inpl);
        acc = epi((long long)c.zl * g.s_z + inpl, r);
    }
    if (Epi::REDUCES) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) epi.partials[linear_block_id()] = acc;
    }
}

// =============================================================================================
// radius-2 stencils evaluated from x alone (scalar per voxel)
// =============================================================================================
template <typename T> struct XA {
    DG g;
    const T* x;
    const T* xp;
    const T* xn;
    int hp;
    __device__ __forceinline__ bool in(int zl, int t, int y, int c) const {
        const int gz = g.z0 + zl;
        return gz >= 0 && gz < g.nzg && t >= 0 && t < g.m && y >= 0 && y < g.ny && c >= 0 && c < g.nx;
    }
    __device__ __forceinline__ T at(int zl, int t, int y, int c) const {
        const T* p = zplane<T>(g, x, xp, xn, hp, zl);
        return p ? p[(long long)t * g.s_t + (long long)y * g.rp + c] : T(0);
    }
};

// channel value of D x at voxel q for (axis, type); 0 when q is outside the volume.
// axis: 0 rows, 1 cols, 2 z, 3 t.  type: hybrid only, 0 = up, 1 = down.
template <int S, typename T>
__device__ __forceinline__ T dval(const XA<T>& X, const WT<T>& w, int axis, int type, int zl, int t, int y, int c) {
    if (!X.in(zl, t, y, c)) return T(0);
    const int dz = (axis == 2), dt = (axis == 3), dy = (axis == 0), dc = (axis == 1);
    int mode = (S == UPWIND) ? 0 : (S == DOWNWIND) ? 1 : (S == HYBRID) ? type : 2;
    if (S == CENTRAL && ((axis == 2 && X.g.z_two) || (axis == 3 && X.g.t_two))) mode = 0;
    const bool hn = X.in(zl + dz, t + dt, y + dy, c + dc);
    const bool hq = X.in(zl - dz, t - dt, y - dy, c - dc);
    T diff = T(0);
    if (mode == 0) { if (hn) diff = X.at(zl + dz, t + dt, y + dy, c + dc) - X.at(zl, t, y, c); }
    else if (mode == 1) { if (hq) diff = X.at(zl, t, y, c) - X.at(zl - dz, t - dt, y - dy, c - dc); }
    else { if (hn && hq) diff = X.at(zl + dz, t + dt, y + dy, c + dc) - X.at(zl - dz, t - dt, y - dy, c - dc); }
    if (axis == 2) diff = w.wz * diff;
    if (axis == 3) {
        diff = w.wt * diff;
        diff *= mask_factor1<T>(X.g, w.sf, y, c);
        diff *= vol_factor1<T>(X.g, zl, t, y, c);
    }
    if (S == HYBRID) diff *= Consts<T>::inv_sqrt2();
    if (S == CENTRAL) diff *= T(0.5);
    return diff;
}

// MODE 0: sub-gradient G = unit-weight adjoint of (Dx / |Dx|)   (pytv/tv_CPU.py:91-126 etc.)
// MODE 1: out = x + rho * D^T D x, partial of <x, out>
template <int S, typename T, int MODE>
__global__ __launch_bounds__(256) void k_gather(XA<T> X, WT<T> w, const T* norms_ext, T rho, T* out, double* partials) {
    __shared__ double sm[16];
    const DG& g = X.g;
    const Coord c = thread_coord<1>(g, 0);
    double acc = 0.0;
    if (c.ok) {
        const int zl = c.zl, t = c.t, y = c.y, col = c.col0;
        auto F = [&](int axis, int type, int qz, int qt, int qy, int qc) -> T {
            if (!X.in(qz, qt, qy, qc)) return T(0);
            const T d = dval<S, T>(X, w, axis, type, qz, qt, qy, qc);
            if (MODE == 0) {
                const T n = norms_ext[(long long)(qz + 1) * g.s_z + (long long)qt * g.s_t + (long long)qy * g.rp + qc];
                return d * n;           // n = 1/|Dx| from pass 1 (0 where |Dx| == 0)
            }
            // D^T D: the adjoint scales the time sample of voxel q by q's own weight-volume factor
            return (axis == 3) ? d * vol_factor1<T>(g, qz, qt, qy, qc) : d;
        };
        T r = T(0), rt = T(0);
        const int naxes = 4;
        for (int axis = 0; axis < naxes; ++axis) {
            if (axis == 2 && !g.za) continue;
            if (axis == 3 && !g.ta) continue;
            const int dz = (axis == 2), dt = (axis == 3), dy = (axis == 0), dc = (axis == 1);
            const int ntypes = (S == HYBRID) ? 2 : 1;
            for (int type = 0; type < ntypes; ++type) {
                int mode = (S == UPWIND) ? 0 : (S == DOWNWIND) ? 1 : (S == HYBRID) ? type : 2;
                if (S == CENTRAL && ((axis == 2 && g.z_two) || (axis == 3 && g.t_two))) mode = 0;
                T term;
                if (mode == 0) term = F(axis, type, zl - dz, t - dt, y - dy, col - dc) - F(axis, type, zl, t, y, col);
                else if (mode == 1) term = F(axis, type, zl, t, y, col) - F(axis, type, zl + dz, t + dt, y + dy, col + dc);
                else term = F(axis, type, zl - dz, t - dt, y - dy, col - dc) - F(axis, type, zl + dz, t + dt, y + dy, col + dc);
                if (MODE == 1) {
                    if (axis == 2) term = w.wz * term;
                    if (axis == 3) term = w.wt * term;
                }
                if (axis == 3) rt += term; else r += term;
            }
        }
        if (MODE == 1 && g.ta) rt *= mask_factor1<T>(g, w.sf, y, col);
        r += rt;
        if (S == HYBRID) r *= Consts<T>::inv_sqrt2();
        if (S == CENTRAL) r *= T(0.5);
        const long long off = (long long)zl * g.s_z + (long long)t * g.s_t + (long long)y * g.rp + col;
        if (MODE == 0) {
            out[off] = r;
        } else {
            const T xv = X.x[off];
            const T o = xv + rho * r;
            out[off] = o;
            acc = (double)xv * (double)o;
        }
    }
    if (MODE == 1) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) partials[linear_block_id()] = acc;
    }
}


// =============================================================================================
// vectorised forms of the two gather kernels for the schemes whose stencil has radius 1 in x
// (upwind, downwind, hybrid).  With f = forward and b = backward difference at the site:
//   D up-channel at p-e equals (w b)(p), D down-channel at p+e equals (w f)(p), so
//   G(p)      = s * sum_a [ up: d_b/n(p-e) - d_f/n(p) ]  +  [ down: d_b/n(p) - d_f/n(p+e) ]
//   D^T D x   = sum_a w_a^2 (b_a - f_a)           (identical for the three schemes)
// where d_f = ((w f) mf) s, d_b = ((w b) mf) s reproduce the arithmetic of D exactly.
// =============================================================================================
template <int S, typename T, int V>
__global__ __launch_bounds__(256) void k_subgrad_vec(DG g, WT<T> w, const T* x, const T* xp, const T* xn, const T* norms_ext, T* G) {
    const Coord c = thread_coord<V>(g, 0);
    if (!c.ok) return;
    const Vec<T, V> r = sg_site<S, T, V>(g, w, x, xp, xn, norms_ext, c, PlainMem(), PlainMem());
    vstore<T, V>(G + (long long)c.zl * g.s_z + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0, r);
}

template <int S, typename T, int V>
__global__ __launch_bounds__(256) void k_normal_vec(DG g, WT<T> w, const T* x, const T* xp, const T* xn, T rho, T* out, double* partials) {
    static_assert(S != CENTRAL, "radius-2 scheme: use k_gather");
    __shared__ double sm[16];
    const Coord c = thread_coord<V>(g, 0);
    double acc = 0.0;
    if (c.ok) {
        const T* pc = zplane<T>(g, x, xp, xn, 2, c.zl);
        const T* pp = g.za ? zplane<T>(g, x, xp, xn, 2, c.zl - 1) : nullptr;
        const T* pn = g.za ? zplane<T>(g, x, xp, xn, 2, c.zl + 1) : nullptr;
        XN<T, V> xs;
        load_xn<T, V, true, true>(g, pc, pp, pn, c, xs);
        const Vec<T, V> zero = vsplat<T, V>(T(0));
        Vec<T, V> r = zero;
        r = r + ((xs.h_pr ? xs.c - xs.pr : zero) - (xs.h_nr ? xs.nr - xs.c : zero));
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int col = c.col0 + i;
            r.v[i] += ((col > 0) ? xs.c.v[i] - xs.pc.v[i] : T(0)) - ((col < g.nx - 1) ? xs.nc.v[i] - xs.c.v[i] : T(0));
        }
        if (g.za) r = r + (w.wz * w.wz) * ((xs.h_pz ? xs.c - xs.pz : zero) - (xs.h_nz ? xs.nz - xs.c : zero));
        if (g.ta) {
            const Vec<T, V> mf = mask_factor<T, V>(g, w.sf, c.y, c.col0);
            if (g.wv == nullptr) {
                r = r + ((w.wt * w.wt) * ((xs.h_pt ? xs.c - xs.pt : zero) - (xs.h_nt ? xs.nt - xs.c : zero))) * (mf * mf);
            } else {
                // weight volume f(z,t,y,x): up channels weigh the backward term with f(t-1)^2 and the forward one with f(t)^2,
                // down channels with f(t)^2 and f(t+1)^2; hybrid is the mean of the two
                const Vec<T, V> f0 = vol_factor<T, V>(g, c.zl, c.t, c.y, c.col0), fp = vol_factor<T, V>(g, c.zl, c.t - 1, c.y, c.col0),
                                fn = vol_factor<T, V>(g, c.zl, c.t + 1, c.y, c.col0);
                const Vec<T, V> b = xs.h_pt ? xs.c - xs.pt : zero, f = xs.h_nt ? xs.nt - xs.c : zero;
                Vec<T, V> tt;
                if (S == UPWIND) tt = b * (fp * fp) - f * (f0 * f0);
                else if (S == DOWNWIND) tt = b * (f0 * f0) - f * (fn * fn);
                else tt = T(0.5) * ((b * (fp * fp) - f * (f0 * f0)) + (b * (f0 * f0) - f * (fn * fn)));
                r = r + ((w.wt * w.wt) * tt) * (mf * mf);
            }
        }
        const long long off = (long long)c.zl * g.s_z + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
        Vec<T, V> o;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            o.v[i] = xs.c.v[i] + rho * r.v[i];
            acc += (double)xs.c.v[i] * (double)o.v[i];
        }
        zero_pad_cols<T, V>(g, c.col0, o);
        vstore<T, V>(out + off, o);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[linear_block_id()] = acc;
}

// central scheme: D^T D x = 1/4 sum_a w_a^2 [ v(p-e) (x(p) - x(p-2e)) - v(p+e) (x(p+2e) - x(p)) ], v(q) = 1 iff q is an
// interior point of the axis (0 < q_a < n_a - 1); two-point z / t axes use the forward stencil instead
template <typename T, int V>
__global__ __launch_bounds__(256) void k_normal_central_vec(DG g, WT<T> w, const T* x, const T* xp, const T* xn, T rho, T* out,
                                                           double* partials) {
    __shared__ double sm[16];
    const Coord c = thread_coord<V>(g, 0);
    double acc = 0.0;
    if (c.ok) {
        const Vec<T, V> zero = vsplat<T, V>(T(0));
        const long long inpl = (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
        const T* pc = zplane<T>(g, x, xp, xn, 2, c.zl) + inpl;
        const Vec<T, V> xc = vload<T, V>(pc);
        Vec<T, V> r = zero;
        // generic axis term given the vectors two steps before / after (zero where they do not exist)
        auto axis2 = [&](int pos, int n, const Vec<T, V>& m2, const Vec<T, V>& p2) -> Vec<T, V> {
            Vec<T, V> a = zero;
            if (pos - 1 > 0 && pos - 1 < n - 1) a = a + (xc - m2);       // q = p-e interior
            if (pos + 1 > 0 && pos + 1 < n - 1) a = a - (p2 - xc);       // q = p+e interior
            return a;
        };
        auto axis1 = [&](int pos, int n, const Vec<T, V>& m1, const Vec<T, V>& p1) -> Vec<T, V> {   // forward-stencil fallback
            Vec<T, V> a = zero;
            if (pos >= 1) a = a + (xc - m1);
            if (pos <= n - 2) a = a - (p1 - xc);
            return a;
        };
        // rows
        r = r + axis2(c.y, g.ny, (c.y >= 2) ? vload<T, V>(pc - 2 * (long long)g.rp) : zero,
                      (c.y + 2 < g.ny) ? vload<T, V>(pc + 2 * (long long)g.rp) : zero);
        // columns (per element)
        {
            const Vec<T, V> lv = (c.col0 >= V) ? vload<T, V>(pc - V) : zero, rv = (c.col0 + 2 * V <= g.nx) ? vload<T, V>(pc + V) : zero;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int col = c.col0 + i;
                T m2, p2;
                if (V >= 2) {
                    m2 = (i >= 2) ? xc.v[i >= 2 ? i - 2 : 0] : lv.v[(V - 2 + i) % V];
                    p2 = (i + 2 < V) ? xc.v[(i + 2 < V) ? i + 2 : 0] : rv.v[(i + 2) % V];
                } else {
                    m2 = (col >= 2) ? pc[i - 2] : T(0);
                    p2 = (col + 2 < g.nx) ? pc[i + 2] : T(0);
                }
                T a = T(0);
                if (col - 1 > 0 && col - 1 < g.nx - 1) a += xc.v[i] - m2;
                if (col + 1 > 0 && col + 1 < g.nx - 1) a -= p2 - xc.v[i];
                r.v[i] += a;
            }
        }
        if (g.za) {
            const int gz = g.z0 + c.zl;
            if (g.z_two) {
                const T* pm = zplane<T>(g, x, xp, xn, 2, c.zl - 1);
                const T* pp = zplane<T>(g, x, xp, xn, 2, c.zl + 1);
                r = r + (w.wz * w.wz) * axis1(gz, g.nzg, pm ? vload<T, V>(pm + inpl) : zero, pp ? vload<T, V>(pp + inpl) : zero);
            } else {
                const T* pm = zplane<T>(g, x, xp, xn, 2, c.zl - 2);
                const T* pp = zplane<T>(g, x, xp, xn, 2, c.zl + 2);
                r = r + (w.wz * w.wz) * axis2(gz, g.nzg, pm ? vload<T, V>(pm + inpl) : zero, pp ? vload<T, V>(pp + inpl) : zero);
            }
        }
        if (g.ta) {
            Vec<T, V> rt;
            if (g.wv == nullptr) {
                if (g.t_two) rt = (w.wt * w.wt) * axis1(c.t, g.m, (c.t >= 1) ? vload<T, V>(pc - g.s_t) : zero,
                                                                 (c.t + 1 < g.m) ? vload<T, V>(pc + g.s_t) : zero);
                else rt = (w.wt * w.wt) * axis2(c.t, g.m, (c.t >= 2) ? vload<T, V>(pc - 2 * g.s_t) : zero,
                                                (c.t + 2 < g.m) ? vload<T, V>(pc + 2 * g.s_t) : zero);
            } else {
                // weight volume: the term that comes from the channel at voxel q carries f(q)^2
                const Vec<T, V> f0 = vol_factor<T, V>(g, c.zl, c.t, c.y, c.col0), fp = vol_factor<T, V>(g, c.zl, c.t - 1, c.y, c.col0),
                                fn = vol_factor<T, V>(g, c.zl, c.t + 1, c.y, c.col0);
                Vec<T, V> a = zero;
                if (g.t_two) {                       // forward stencil: channel at p-e and at p
                    if (c.t >= 1) a = a + (xc - vload<T, V>(pc - g.s_t)) * (fp * fp);
                    if (c.t <= g.m - 2) a = a - (vload<T, V>(pc + g.s_t) - xc) * (f0 * f0);
                } else {                             // central: channel at p-e and at p+e (interior frames only)
                    if (c.t - 1 > 0 && c.t - 1 < g.m - 1) a = a + (xc - vload<T, V>(pc - 2 * g.s_t)) * (fp * fp);
                    if (c.t + 1 > 0 && c.t + 1 < g.m - 1) a = a - (vload<T, V>(pc + 2 * g.s_t) - xc) * (fn * fn);
                }
                rt = (w.wt * w.wt) * a;
            }
            const Vec<T, V> mf = mask_factor<T, V>(g, w.sf, c.y, c.col0);
            r = r + rt * (mf * mf);
        }
        Vec<T, V> o;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            o.v[i] = xc.v[i] + rho * (T(0.25) * r.v[i]);
            acc += (double)xc.v[i] * (double)o.v[i];
        }
        zero_pad_cols<T, V>(g, c.col0, o);
        vstore<T, V>(out + (long long)c.zl * g.s_z + inpl, o);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[linear_block_id()] = acc;
}

// central scheme sub-gradient, vectorised:  G(p) = 1/2 sum_a [ g_a(p-e) - g_a(p+e) ],  g_a(q) = d_a(q) / |D x|(q),
// (norms_ext holds 1/|Dx|)  d_a(q) = 1/2 w_a (x(q+e) - x(q-e)) at interior q (mask factor on the time channel); two-point z / t axes use
// the forward stencil: G += g_a(p-e) - g_a(p), d_a(q) = 1/2 w_a (x(q+e) - x(q)).   (pytv/tv_CPU.py:302-330)
template <typename T, int V>
__global__ __launch_bounds__(256) void k_subgrad_central_vec(DG g, WT<T> w, const T* x, const T* xp, const T* xn, const T* norms_ext,
                                                            T* G) {
    const Coord c = thread_coord<V>(g, 0);
    if (!c.ok) return;
    const Vec<T, V> hr = sg_site_central<T, V>(g, w, x, xp, xn, norms_ext, c, PlainMem(), PlainMem());
    const long long inpl = (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
    vstore<T, V>(G + (long long)c.zl * g.s_z + inpl, hr);
}

// =============================================================================================
// l2,1 norm of a materialised gradient (pytv/tv_operators_GPU.py:75-81 as ONE pass)
// =============================================================================================
template <typename T, int V>
__global__ __launch_bounds__(256) void k_l21(DG g, const T* d, T* norms, double* partials) {
    __shared__ double sm[16];
    const Coord c = thread_coord<V>(g, 0);
    double acc = 0.0;
    if (c.ok) {
        const long long inpl = (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
        const T* base = d + (long long)c.zl * g.s_dz + inpl;
        Vec<T, V> s = vsplat<T, V>(T(0));
        for (int ch = 0; ch < g.nd; ++ch) {
            const Vec<T, V> v = vload_s<T, V>(base + (long long)ch * g.s_z);       // read once: non-temporal (tv_device.h)
            s = s + v * v;
        }
        Vec<T, V> n;
#pragma unroll
        for (int i = 0; i < V; ++i) { n.v[i] = tsqrt(s.v[i]); acc += (double)n.v[i]; }
        if (norms != nullptr) vstore<T, V>(norms + (long long)c.zl * g.s_z + inpl, n);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[linear_block_id()] = acc;
}

#ifndef TV_CG_NT
#define TV_CG_NT 1               // the streams of the flat kernels (every array read / written once) non-temporal: k_cgcg -2 % on the CG outer iteration; read-only streams gain most (tv_l21 0.71 -> 0.81 - 0.88 of 8 TB/s with nt loads); 0: plain
#endif
template <typename T, int V> __device__ __forceinline__ Vec<T, V> CGLD(const T* p) {
#if TV_CG_NT
    return vload_s<T, V>(p);
#else
    return vload<T, V>(p);
#endif
}
template <typename T, int V> __device__ __forceinline__ void CGST(T* p, const Vec<T, V>& a) {
#if TV_CG_NT
    vstore_s<T, V>(p, a);
#else
    vstore<T, V>(p, a);
#endif
}
// =============================================================================================
// flat streaming kernels (grid-stride, one V-wide 16-byte vector per lane per trip; nv = n / V vectors)
// =============================================================================================
template <typename T, int V> __global__ __launch_bounds__(256) void k_sub(long long nv, const T* a, const T* b, T* out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x)
        CGST<T, V>(out + i * V, CGLD<T, V>(a + i * V) - CGLD<T, V>(b + i * V));
}
template <typename T, int V> __global__ __launch_bounds__(256) void k_dot(long long nv, const T* a, const T* b, double* partials) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const Vec<T, V> av = CGLD<T, V>(a + i * V), bv = CGLD<T, V>(b + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) acc += (double)av.v[k] * (double)bv.v[k];
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// alpha = rs/dAd;  x += alpha d;  r -= alpha Ad;  partial <r, r>
template <typename T, int V>
__global__ __launch_bounds__(256) void k_cg1(long long nv, T* x, T* r, const T* d, const T* Ad, const double* rs,
                                              const double* dAd, double* partials) {
    __shared__ double sm[16];
    const double den = *dAd;
    const T alpha = (den > 0.0) ? (T)(*rs / den) : T(0);
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        CGST<T, V>(x + i * V, CGLD<T, V>(x + i * V) + alpha * CGLD<T, V>(d + i * V));
        const Vec<T, V> rn = CGLD<T, V>(r + i * V) - alpha * CGLD<T, V>(Ad + i * V);
        CGST<T, V>(r + i * V, rn);
#pragma unroll
        for (int k = 0; k < V; ++k) acc += (double)rn.v[k] * (double)rn.v[k];
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// beta = rs_new/rs;  d = r + beta d
template <typename T, int V>
__global__ __launch_bounds__(256) void k_cg2(long long nv, T* d, const T* r, const double* rs_new, const double* rs) {
    const double den = *rs;
    const T beta = (den > 0.0) ? (T)(*rs_new / den) : T(0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x)
        CGST<T, V>(d + i * V, CGLD<T, V>(r + i * V) + beta * CGLD<T, V>(d + i * V));
}
// single-reduction CG (Chronopoulos-Gear): sc = {gamma = <r,r>, delta = <r,w>, gamma_old, alpha_old}; alpha_old == 0 marks
// the first step of a solve.  d = r + beta d; s = w + beta s; x += alpha d; r -= alpha s; optional partial 1/2 |x - x0|^2
__device__ __forceinline__ void cgcg_scalars(const double* sc, double& alpha, double& beta, bool& first) {
    const double gamma = sc[0], delta = sc[1], gamma_old = sc[2], alpha_old = sc[3];
    first = (alpha_old == 0.0);
    beta = (!first && gamma_old > 0.0) ? gamma / gamma_old : 0.0;
    const double den = first ? delta : delta - beta * gamma / alpha_old;
    alpha = (den > 0.0) ? gamma / den : 0.0;
}
template <typename T, int V>
__global__ __launch_bounds__(256) void k_cgcg(long long nv, T* x, T* r, T* d, T* s, const T* w, const double* sc, const T* x0,
                                               double* partials) {
    __shared__ double sm[16];
    double al, be;
    bool first;
    cgcg_scalars(sc, al, be, first);
    const T alpha = (T)al, beta = (T)be;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const Vec<T, V> rv = CGLD<T, V>(r + i * V), wv = CGLD<T, V>(w + i * V);
        Vec<T, V> dn = rv, sn = wv;
        if (!first) {
            dn = rv + beta * CGLD<T, V>(d + i * V);
            sn = wv + beta * CGLD<T, V>(s + i * V);
        }
        const Vec<T, V> xn = CGLD<T, V>(x + i * V) + alpha * dn;
        CGST<T, V>(d + i * V, dn);
        CGST<T, V>(s + i * V, sn);
        CGST<T, V>(x + i * V, xn);
        CGST<T, V>(r + i * V, rv - alpha * sn);
        if (x0 != nullptr) {
            const Vec<T, V> x0v = CGLD<T, V>(x0 + i * V);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const double e = (double)xn.v[k] - (double)x0v.v[k];
                acc += 0.5 * e * e;
            }
        }
    }
    if (x0 != nullptr) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) partials[blockIdx.x] = acc;
    }
}
__global__ void k_cgcg_advance(double* sc) {
    double al, be;
    bool first;
    cgcg_scalars(sc, al, be, first);
    sc[2] = sc[0];
    sc[3] = (al != 0.0) ? al : 1e-300;       // never 0 again inside a solve: 0 means "first step"
}
// out = a - b; partial <out, out>; optional second copy
template <typename T, int V>
__global__ __launch_bounds__(256) void k_sub_dot(long long nv, const T* a, const T* b, T* out, T* out2, double* partials) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const Vec<T, V> o = CGLD<T, V>(a + i * V) - CGLD<T, V>(b + i * V);
        CGST<T, V>(out + i * V, o);
        if (out2 != nullptr) CGST<T, V>(out2 + i * V, o);
#pragma unroll
        for (int k = 0; k < V; ++k) acc += (double)o.v[k] * (double)o.v[k];
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// Chebyshev step, composed form (tv_cheb_step where the streaming normal operator does not apply): ax = A x is in `out`
//   out = [add +] x + alpha (b - ax) + beta (x - y);  partials: |b - ax|^2 and |out - ref|^2 (or |x|^2)
template <typename T, int V>
__global__ __launch_bounds__(256) void k_cheb_combine(long long nv, const T* x, const T* b, const T* y, const T* add, const T* ref, T* out, T alpha,
                                                      T beta, T yscale, double* part0, double* part1) {
    __shared__ double sm[16];
    double acc0 = 0.0, acc1 = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const Vec<T, V> xv = CGLD<T, V>(x + i * V), ax = CGLD<T, V>(out + i * V), bv = CGLD<T, V>(b + i * V);
        const Vec<T, V> yv = (y != nullptr) ? CGLD<T, V>(y + i * V) : yscale * bv;
        Vec<T, V> o;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const T res = bv.v[k] - ax.v[k];
            o.v[k] = (xv.v[k] + alpha * res) + beta * (xv.v[k] - yv.v[k]);
            acc0 += (double)res * (double)res;
        }
        if (add != nullptr) o = CGLD<T, V>(add + i * V) + o;
        Vec<T, V> rv = xv;
        if (ref != nullptr) rv = CGLD<T, V>(ref + i * V);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const double e = (ref != nullptr) ? (double)o.v[k] - (double)rv.v[k] : (double)xv.v[k];
            acc1 += e * e;
        }
        CGST<T, V>(out + i * V, o);
    }
    acc0 = block_sum(acc0, sm);
    if (threadIdx.x == 0) part0[blockIdx.x] = acc0;
    acc1 = block_sum(acc1, sm);
    if (threadIdx.x == 0) part1[blockIdx.x] = acc1;
}
// out = a x + b y (y may be nullptr: out = a x); partial |out - ref|^2 when ref is given
template <typename T, int V>
__global__ __launch_bounds__(256) void k_axpby(long long nv, T a, const T* x, T b, const T* y, const T* ref, T* out, double* partials) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        Vec<T, V> o = a * CGLD<T, V>(x + i * V);
        if (y != nullptr) o = o + b * CGLD<T, V>(y + i * V);
        if (ref != nullptr) {
            const Vec<T, V> rv = CGLD<T, V>(ref + i * V);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const double e = (double)o.v[k] - (double)rv.v[k];
                acc += e * e;
            }
        }
        if (out != nullptr) CGST<T, V>(out + i * V, o);          // out == nullptr: the distance only (a pure reduction: one word less)
    }
    if (ref != nullptr) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) partials[blockIdx.x] = acc;
    }
}
// Chambolle-Pock with a data-fidelity operator (README.md:148 with A != I), data space, any length:
//   k_cpop_p  : p <- (p + sigma r) / (1 + sigma)              (r = A x - b carried from the previous iteration)
//   k_cpop_res: r <- Ax - b,  partial 1/2 |r|^2
template <typename T, int V> __global__ __launch_bounds__(256) void k_cpop_p(long long nv, T* p, const T* r, T sigma, T inv) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x)
        CGST<T, V>(p + i * V, inv * (CGLD<T, V>(p + i * V) + sigma * CGLD<T, V>(r + i * V)));
}
template <typename T, int V> __global__ __launch_bounds__(256) void k_cpop_res(long long nv, const T* ax, const T* b, T* r, double* partials) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const Vec<T, V> o = CGLD<T, V>(ax + i * V) - CGLD<T, V>(b + i * V);
        CGST<T, V>(r + i * V, o);
#pragma unroll
        for (int k = 0; k < V; ++k) acc += 0.5 * (double)o.v[k] * (double)o.v[k];
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// README.md:122-123: x <- x - step * ((x - x0) + lambda G); partial 1/2 |x - x0|^2
template <typename T, int V>
__global__ __launch_bounds__(256) void k_sgstep(long long nv, T* x, const T* x0, const T* G, T step, T lambda, double* partials) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const Vec<T, V> xv = CGLD<T, V>(x + i * V), x0v = CGLD<T, V>(x0 + i * V), gv = CGLD<T, V>(G + i * V);
        Vec<T, V> xn;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            xn.v[k] = xv.v[k] - step * ((xv.v[k] - x0v.v[k]) + lambda * gv.v[k]);
            const double e = (double)xn.v[k] - (double)x0v.v[k];
            acc += 0.5 * e * e;
        }
        CGST<T, V>(x + i * V, xn);
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// deterministic tree over the per-block partials: out[b] = sum of chunk b
__global__ __launch_bounds__(256) void k_reduce(const double* in, long long n, double* out) {
    __shared__ double sm[16];
    const long long chunk = (n + gridDim.x - 1) / gridDim.x;
    const long long lo = (long long)blockIdx.x * chunk;
    const long long hi = (lo + chunk < n) ? lo + chunk : n;
    double acc = 0.0;
    for (long long i = lo + threadIdx.x; i < hi; i += blockDim.x) acc += in[i];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

}  // namespace tv

extern "C" {

const char* tv_last_error(void) { return g_err.c_str(); }
int tv_version(void) { return 500; }
int tv_abi_version(void) { return TV_ABI_VERSION; }

int tv_set_option(const char* name, int value) {
    TvOption* o = (name != nullptr) ? find_option(name) : nullptr;
    if (o == nullptr) return fail(TV_E_ARG, "unknown option");
    o->value = value;
    o->has = 1;
    return 0;
}
int tv_unset_option(const char* name) {
    TvOption* o = (name != nullptr) ? find_option(name) : nullptr;
    if (o == nullptr) return fail(TV_E_ARG, "unknown option");
    o->has = 0;
    return 0;
}
int tv_get_option(const char* name, int dflt) { return (name != nullptr) ? env_int(name, dflt) : dflt; }

int tv_num_channels(const tv_geom* g) {
    DG d;
    int rc = make_dg(g, d, true);
    return rc ? rc : d.nd;
}

size_t tv_workspace_bytes(const tv_geom* g) {
    DG d;
    if (make_dg(g, d, true)) return 0;
    return (size_t)(3 * (max_partials(d) + kStage + 16)) * sizeof(double);    // three independent partial arrays (the third: tv_cp_sweep with TV_CP_FID_BOTH, round 5)
}

// ---------------------------------------------------------------------------------------------
int tv_D(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* dout, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || dout == nullptr) return fail(TV_E_ARG, "NULL array");
    if (int rc = check_x_halos(g, d, x_prev, x_next)) return rc;
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_prev, x_next, dout, d.wv, d.wvp, d.wvn});
    hipStream_t st = (hipStream_t)stream;
    // TV_D_KERNEL: 0 = one site per thread (k_D), 1 = plane-marching with an LDS tile (k_D_march), 2 = streaming
    // (k_D_stream: no tile, no barrier, x read once).  Default: streaming for planes of at least TV_MARCH_MIN_PLANE_KB
    // (64x8x1024x1024, ms hybrid / upwind / downwind / central: k_D 4.90 / 2.39 / 2.37 / 3.04, k_D_march 5.76 / 2.71 /
    // 2.47 / 2.99, k_D_stream 4.31 / 2.16 / 2.17 / 2.31); on small planes the z / t neighbours of the one-site kernel
    // stay in L2 and it wins (256x1x512x512: 0.39 / 0.19 ms against 0.40 / 0.23) -- profiles/r2_d_kernels.txt
    const bool big_plane = (long long)d.s_z * (g->dtype == TV_F32 ? 4 : 8) >= (long long)env_int("TV_MARCH_MIN_PLANE_KB", 4096) * 1024;
    const int kern = env_int("TV_D_KERNEL", env_int("TV_MARCH_D", 0) ? 1 : (big_plane ? 2 : 0));
    if (kern == 2 && tvm::D_stream_ok(g, d, vec) && !env_int("TV_NO_MARCH", 0))
        return tvm::D_stream(g, d, x, x_prev, x_next, st, dout);
    if (kern == 1 && march_ok(g, d, vec)) {
        long long nb;
        return tvm::D_store(g, d, x, x_prev, x_next, st, &nb, (float*)dout);
    }
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        StoreD<S, T, V> epi{(T*)dout, nullptr};
        hipLaunchKernelGGL((k_D<S, T, V, StoreD<S, T, V>>), lc.grid, lc.block, 0, st, d, make_w<T>(g), (const T*)x,
                           (const T*)x_prev, (const T*)x_next, 1, 0, epi);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int tv_DT(const tv_geom* g, const void* y, const void* y_prev, const void* y_next, void* out, void* stream) {
    return tv_DT_axpy(g, y, nullptr, y_prev, y_next, nullptr, 1.0, out, stream);
}

int tv_DT_axpy(const tv_geom* g, const void* a, const void* b, const void* ab_prev, const void* ab_next,
               const void* base, double alpha, void* out, void* stream) {
    return tv_DT_axpy2(g, a, b, ab_prev, ab_next, base, nullptr, 0.0, alpha, out, stream);
}

int tv_DT_axpy2(const tv_geom* g, const void* a, const void* b, const void* ab_prev, const void* ab_next,
                const void* base, const void* base2, double beta, double alpha, void* out, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (a == nullptr || out == nullptr) return fail(TV_E_ARG, "NULL array");
    if (int rc = check_y_halos(g, d, ab_prev, ab_next)) return rc;
    const bool vec = rows_vectorisable(g, d) && aligned16({a, b, ab_prev, ab_next, base, base2, out, d.wv});
    hipStream_t st = (hipStream_t)stream;
    const bool plain_store = (base == nullptr && base2 == nullptr && alpha == 1.0);
    if (b == nullptr && march_dt_ok(g, d, vec, plain_store)) {          // plane-marching adjoint (tv_march.h): fp32, and since round 4 fp64 hybrid tv_DT
        long long nb;
        if (plain_store) return tvm::DT_store(g, d, a, ab_prev, ab_next, st, &nb, out);
        return tvm::DT_axpy(g, d, a, ab_prev, ab_next, st, &nb, out, base, alpha, base2, beta);
    }
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        WT<T> w = make_w<T>(g);
        if (b == nullptr) {
            SrcPlain<T, V> src{(const T*)a, (const T*)ab_prev, (const T*)ab_next};
            if (plain_store) {
                StoreDT<T, V> epi{(T*)out, nullptr};
                hipLaunchKernelGGL((k_DT<S, T, V, SrcPlain<T, V>, StoreDT<T, V>>), lc.grid, lc.block, 0, st, d, w, src, epi);
            } else {
                AxpyDT<T, V> epi{(T*)out, (const T*)base, (T)alpha, nullptr, (const T*)base2, (T)beta};
                hipLaunchKernelGGL((k_DT<S, T, V, SrcPlain<T, V>, AxpyDT<T, V>>), lc.grid, lc.block, 0, st, d, w, src, epi);
            }
        } else {
            SrcDiff<T, V> src{(const T*)a, (const T*)b, (const T*)ab_prev, (const T*)ab_next};
            AxpyDT<T, V> epi{(T*)out, (const T*)base, (T)alpha, nullptr, (const T*)base2, (T)beta};
            hipLaunchKernelGGL((k_DT<S, T, V, SrcDiff<T, V>, AxpyDT<T, V>>), lc.grid, lc.block, 0, st, d, w, src, epi);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int tv_l21(const tv_geom* g, const void* dimg, int32_t nd, void* norms, double* result, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (dimg == nullptr || result == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (nd < 1) return fail(TV_E_CHANNELS, "nd must be >= 1");
    // the l2,1 norm does not care which scheme produced the channels: honour the caller's nd
    d.nd = nd;
    d.s_dz = d.s_z * nd;
    const bool vec = rows_vectorisable(g, d) && aligned16({dimg, norms});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    return dispatch(0, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        hipLaunchKernelGGL((k_l21<T, V>), lc.grid, lc.block, 0, st, d, (const T*)dimg, (T*)norms, (double*)ws);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lc.nblocks, nmax, result, st);
    });
}

// ---------------------------------------------------------------------------------------------
int tv_subgrad(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* G, void* norms_ext,
               double* tvout, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || G == nullptr || norms_ext == nullptr || tvout == nullptr || ws == nullptr)
        return fail(TV_E_ARG, "NULL array");
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr)))
        return fail(TV_E_HALO, "tv_subgrad on a slab needs two halo planes on each interior side");
    if (d.za && d.ta && d.wv != nullptr && ((e_lo && d.wvp == nullptr) || (e_hi && d.wvn == nullptr)))
        return fail(TV_E_HALO, "tv_subgrad on a slab with a weight volume needs time_weight_prev / time_weight_next");
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_prev, x_next, norms_ext, d.wv, d.wvp, d.wvn});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    if (march_ok(g, d, vec && aligned16({G})) && !env_int("TV_NO_MARCH_SUBGRAD", 0)) {
        // plane-marching passes: every field is fetched once (tv_march.h / tv_fused.h)
        const int glo = d.za ? e_lo : 0, ghi = d.za ? e_hi : 0;
        long long nb;
        if (int rc = tvm::D_norms(g, d, x, x_prev, x_next, st, &nb, (float*)norms_ext, (double*)ws, glo, ghi)) return rc;
        if (int rc = reduce_partials((double*)ws, nb, nmax, tvout, st)) return rc;
        if (tvm::subgrad_pass2_ok(g, d))
            return tvm::subgrad_pass2(g, d, x, x_prev, x_next, st, (const float*)norms_ext, (float*)G);
        // no marching gather for this case (central: radius-2 stencil; M > 8): one site per thread
        LC lg = launch_cfg(d, 4, d.nz);
        switch (g->scheme) {
            case TV_UPWIND:
                hipLaunchKernelGGL((k_subgrad_vec<UPWIND, float, 4>), lg.grid, lg.block, 0, st, d, make_w<float>(g), (const float*)x,
                                   (const float*)x_prev, (const float*)x_next, (const float*)norms_ext, (float*)G);
                break;
            case TV_DOWNWIND:
                hipLaunchKernelGGL((k_subgrad_vec<DOWNWIND, float, 4>), lg.grid, lg.block, 0, st, d, make_w<float>(g), (const float*)x,
                                   (const float*)x_prev, (const float*)x_next, (const float*)norms_ext, (float*)G);
                break;
            case TV_HYBRID:
                hipLaunchKernelGGL((k_subgrad_vec<HYBRID, float, 4>), lg.grid, lg.block, 0, st, d, make_w<float>(g), (const float*)x,
                                   (const float*)x_prev, (const float*)x_next, (const float*)norms_ext, (float*)G);
                break;
            default:
                hipLaunchKernelGGL((k_subgrad_central_vec<float, 4>), lg.grid, lg.block, 0, st, d, make_w<float>(g), (const float*)x,
                                   (const float*)x_prev, (const float*)x_next, (const float*)norms_ext, (float*)G);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        WT<T> w = make_w<T>(g);
        // pass 1: norms on the local planes plus one ghost plane per interior side
        const int ghosts_lo = d.za ? e_lo : 0, ghosts_hi = d.za ? e_hi : 0;
        LC lc = launch_cfg(d, V, d.nz + ghosts_lo + ghosts_hi);
        NormEpi<S, T, V> epi{(T*)norms_ext, (double*)ws};
        hipLaunchKernelGGL((k_D<S, T, V, NormEpi<S, T, V>>), lc.grid, lc.block, 0, st, d, w, (const T*)x, (const T*)x_prev,
                           (const T*)x_next, 2, -ghosts_lo, epi);
        HIP_TRY(hipGetLastError());
        if (int rc = reduce_partials((double*)ws, lc.nblocks, nmax, tvout, st)) return rc;
        // pass 2: gather (vectorised for the radius-1 schemes, scalar radius-2 kernel for central)
        if constexpr (S != CENTRAL) {
            const bool v2 = vec && aligned16({G});
            if (v2 && V > 1) {
                LC lg = launch_cfg(d, V, d.nz);
                hipLaunchKernelGGL((k_subgrad_vec<S, T, V>), lg.grid, lg.block, 0, st, d, w, (const T*)x, (const T*)x_prev,
                                   (const T*)x_next, (const T*)norms_ext, (T*)G);
            } else {
                LC lg = launch_cfg(d, 1, d.nz);
                hipLaunchKernelGGL((k_subgrad_vec<S, T, 1>), lg.grid, lg.block, 0, st, d, w, (const T*)x, (const T*)x_prev,
                                   (const T*)x_next, (const T*)norms_ext, (T*)G);
            }
        } else if (env_int("TV_SCALAR_GATHER", 0)) {
            // reference-style scalar evaluation of the radius-2 stencil (kept as an in-library cross-check)
            LC lg = launch_cfg(d, 1, d.nz);
            XA<T> X{d, (const T*)x, (const T*)x_prev, (const T*)x_next, 2};
            hipLaunchKernelGGL((k_gather<S, T, 0>), lg.grid, lg.block, 0, st, X, w, (const T*)norms_ext, T(0), (T*)G, (double*)nullptr);
        } else {
            const bool v2 = vec && aligned16({G});
            if (v2 && V > 1) {
                LC lg = launch_cfg(d, V, d.nz);
                hipLaunchKernelGGL((k_subgrad_central_vec<T, V>), lg.grid, lg.block, 0, st, d, w, (const T*)x, (const T*)x_prev,
                                   (const T*)x_next, (const T*)norms_ext, (T*)G);
            } else {
                LC lg = launch_cfg(d, 1, d.nz);
                hipLaunchKernelGGL((k_subgrad_central_vec<T, 1>), lg.grid, lg.block, 0, st, d, w, (const T*)x, (const T*)x_prev,
                                   (const T*)x_next, (const T*)norms_ext, (T*)G);
            }
        }
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int tv_normal_op(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, double rho, void* out,
                 double* dot, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || out == nullptr || dot == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr)))
        return fail(TV_E_HALO, "tv_normal_op on a slab needs two halo planes on each interior side");
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_prev, x_next, out, d.wv});
    // TV_NORMAL_KERNEL: 2 = streaming (k_normal_stream: default for fp32 planes >= TV_MARCH_MIN_PLANE_KB, radius-1 schemes,
    // any M), 1 = the marching LIGHT kernel of round 1 (M <= 8), 0 = one site per thread
    const int nkern = env_int("TV_NORMAL_KERNEL", 2);
    if (nkern == 2 && tvm::N_stream_ok(g, d, vec)) {
        long long nb;
        double* w0 = (double*)ws;
        double* w1 = w0 + nmax + kStage + 16;
        if (int rc = tvm::N_stream(g, d, x, x_prev, x_next, nullptr, out, nullptr, rho, st, &nb, w0, w1)) return rc;
        return reduce_partials(w0, nb, nmax, dot, st);
    }
    if (nkern >= 1 && g->scheme != TV_CENTRAL && d.m <= 8 && march_ok(g, d, vec) && !env_int("TV_NO_MARCH_NORMAL", 0)) {
        long long nb;
        if (int rc = tvm::D_normal_op(g, d, x, x_prev, x_next, st, &nb, (float*)out, (float)rho, (double*)ws)) return rc;
        return reduce_partials((double*)ws, nb, nmax, dot, st);
    }
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lg = launch_cfg(d, V, d.nz);
        if constexpr (S != CENTRAL) {
            hipLaunchKernelGGL((k_normal_vec<S, T, V>), lg.grid, lg.block, 0, st, d, make_w<T>(g), (const T*)x, (const T*)x_prev,
                               (const T*)x_next, (T)rho, (T*)out, (double*)ws);
        } else if (env_int("TV_SCALAR_GATHER", 0)) {
            LC l1 = launch_cfg(d, 1, d.nz);
            XA<T> X{d, (const T*)x, (const T*)x_prev, (const T*)x_next, 2};
            hipLaunchKernelGGL((k_gather<S, T, 1>), l1.grid, l1.block, 0, st, X, make_w<T>(g), (const T*)nullptr, (T)rho, (T*)out,
                               (double*)ws);
            HIP_TRY(hipGetLastError());
            return reduce_partials((double*)ws, l1.nblocks, nmax, dot, st);
        } else {
            hipLaunchKernelGGL((k_normal_central_vec<T, V>), lg.grid, lg.block, 0, st, d, make_w<T>(g), (const T*)x, (const T*)x_prev,
                               (const T*)x_next, (T)rho, (T*)out, (double*)ws);
        }
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lg.nblocks, nmax, dot, st);
    });
}

// ---------------------------------------------------------------------------------------------
int tv_cp_dual(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* q, double sigma_D,
               double lambda, double* tvout, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || q == nullptr || tvout == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (!(lambda > 0.0)) return fail(TV_E_ARG, "lambda must be > 0");
    if (int rc = check_x_halos(g, d, x_prev, x_next)) return rc;
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_prev, x_next, q, d.wv});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    if (march_ok(g, d, vec)) {
        long long nb;
        if (int rc = tvm::D_cp_dual(g, d, x, x_prev, x_next, st, &nb, (float*)q, (float)sigma_D, (float)(1.0 / lambda),
                                            (double*)ws)) return rc;
        return reduce_partials((double*)ws, nb, nmax, tvout, st);
    }
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        CpDual<S, T, V> epi{(T*)q, (T)sigma_D, (T)(1.0 / lambda), (double*)ws};
        hipLaunchKernelGGL((k_D<S, T, V, CpDual<S, T, V>>), lc.grid, lc.block, 0, st, d, make_w<T>(g), (const T*)x,
                           (const T*)x_prev, (const T*)x_next, 1, 0, epi);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lc.nblocks, nmax, tvout, st);
    });
}

int tv_cp_primal(const tv_geom* g, const void* q, const void* q_prev, const void* q_next, void* x, const void* x0,
                 void* p, double tau, double sigma_A, double* fid, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (q == nullptr || x == nullptr || x0 == nullptr || p == nullptr || fid == nullptr || ws == nullptr)
        return fail(TV_E_ARG, "NULL array");
    if (int rc = check_y_halos(g, d, q_prev, q_next)) return rc;
    const bool vec = rows_vectorisable(g, d) && aligned16({q, q_prev, q_next, x, x0, p, d.wv});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    if (march_ok(g, d, vec)) {
        long long nb;
        if (int rc = tvm::DT_cp_primal(g, d, q, q_prev, q_next, st, &nb, (float*)x, (const float*)x0, (float*)p, (float)tau,
                                               (float)sigma_A, (float)(1.0 / (1.0 + sigma_A)), (double*)ws)) return rc;
        return reduce_partials((double*)ws, nb, nmax, fid, st);
    }
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        SrcPlain<T, V> src{(const T*)q, (const T*)q_prev, (const T*)q_next};
        CpPrimal<T, V> epi{(T*)x, (const T*)x0, (T*)p, (T)tau, (T)sigma_A, (T)(1.0 / (1.0 + sigma_A)), (double*)ws};
        hipLaunchKernelGGL((k_DT<S, T, V, SrcPlain<T, V>, CpPrimal<T, V>>), lc.grid, lc.block, 0, st, d, make_w<T>(g), src, epi);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lc.nblocks, nmax, fid, st);
    });
}

static int admm_zu_impl(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* z, void* u,
                        double thresh, double* tvout, void* ws, void* stream, int tform);
int tv_admm_zu(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* z, void* u,
               double thresh, double* tvout, void* ws, void* stream) {
    return admm_zu_impl(g, x, x_prev, x_next, z, u, thresh, tvout, ws, stream, 0);
}
int tv_admm_tu(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* t, void* u,
               double thresh, double* tvout, void* ws, void* stream) {
    return admm_zu_impl(g, x, x_prev, x_next, t, u, thresh, tvout, ws, stream, 1);
}
static int admm_zu_impl(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* z, void* u,
                        double thresh, double* tvout, void* ws, void* stream, int tform) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || z == nullptr || u == nullptr || tvout == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (int rc = check_x_halos(g, d, x_prev, x_next)) return rc;
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_prev, x_next, z, u, d.wv});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    if (march_ok(g, d, vec)) {
        long long nb;
        if (int rc = tvm::D_admm_zu(g, d, x, x_prev, x_next, st, &nb, (float*)z, (float*)u, (float)thresh, (double*)ws, tform))
            return rc;
        return reduce_partials((double*)ws, nb, nmax, tvout, st);
    }
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        AdmmZU<S, T, V> epi{(T*)z, (T*)u, (T)thresh, (double*)ws, tform};
        hipLaunchKernelGGL((k_D<S, T, V, AdmmZU<S, T, V>>), lc.grid, lc.block, 0, st, d, make_w<T>(g), (const T*)x,
                           (const T*)x_prev, (const T*)x_next, 1, 0, epi);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lc.nblocks, nmax, tvout, st);
    });
}

// ---------------------------------------------------------------------------------------------
static long long nvox(const DG& d) { return d.s_z * d.nz; }

// launch a flat kernel template KERN<T, V> with the widest 16-byte vector that n and the pointers allow
#define TV_FLAT_LAUNCH(KERN, dtype, n, ptrs, ...)                                                                             \
    do {                                                                                                                      \
        const bool al__ = aligned16 ptrs;                                                                                     \
        if ((dtype) == TV_F32) {                                                                                              \
            using T = float;                                                                                                  \
            if (al__ && (n) % 4 == 0) hipLaunchKernelGGL((KERN<T, 4>), dim3(kFlatBlocks), dim3(256), 0, st, (long long)(n) / 4, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERN<T, 1>), dim3(kFlatBlocks), dim3(256), 0, st, (long long)(n), __VA_ARGS__);            \
        } else {                                                                                                              \
            using T = double;                                                                                                 \
            if (al__ && (n) % 2 == 0) hipLaunchKernelGGL((KERN<T, 2>), dim3(kFlatBlocks), dim3(256), 0, st, (long long)(n) / 2, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERN<T, 1>), dim3(kFlatBlocks), dim3(256), 0, st, (long long)(n), __VA_ARGS__);            \
        }                                                                                                                     \
    } while (0)

int tv_sub(int32_t dtype, int64_t n, const void* a, const void* b, void* out, void* stream) {
    if (n < 0 || a == nullptr || b == nullptr || out == nullptr) return fail(TV_E_ARG, "bad argument");
    if (dtype != TV_F32 && dtype != TV_F64) return fail(TV_E_ARG, "unknown dtype");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    TV_FLAT_LAUNCH(k_sub, dtype, n, ({a, b, out}), (const T*)a, (const T*)b, (T*)out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tv_cpop_p(int32_t dtype, int64_t n, void* p, const void* r, double sigma_A, void* stream) {
    if (n < 0 || p == nullptr || r == nullptr) return fail(TV_E_ARG, "bad argument");
    if (dtype != TV_F32 && dtype != TV_F64) return fail(TV_E_ARG, "unknown dtype");
    if (!(sigma_A >= 0.0)) return fail(TV_E_ARG, "sigma_A must be non-negative");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    TV_FLAT_LAUNCH(k_cpop_p, dtype, n, ({p, r}), (T*)p, (const T*)r, (T)sigma_A, (T)(1.0 / (1.0 + sigma_A)));
    HIP_TRY(hipGetLastError());
    return 0;
}

int tv_cpop_residual(int32_t dtype, int64_t n, const void* ax, const void* b, void* r, double* fid, void* ws, void* stream) {
    if (n < 0 || ax == nullptr || b == nullptr || r == nullptr || fid == nullptr || ws == nullptr) return fail(TV_E_ARG, "bad argument");
    if (dtype != TV_F32 && dtype != TV_F64) return fail(TV_E_ARG, "unknown dtype");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(fid, 0, sizeof(double), st));
        return 0;
    }
    TV_FLAT_LAUNCH(k_cpop_res, dtype, n, ({ax, b, r}), (const T*)ax, (const T*)b, (T*)r, (double*)ws);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, st, (const double*)ws, (long long)kFlatBlocks, fid);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tv_dot(const tv_geom* g, const void* a, const void* b, double* result, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (a == nullptr || b == nullptr || result == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    TV_FLAT_LAUNCH(k_dot, g->dtype, nvox(d), ({a, b}), (const T*)a, (const T*)b, (double*)ws);
    HIP_TRY(hipGetLastError());
    return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), result, st);
}

int tv_cg_step1(const tv_geom* g, void* x, void* r, const void* dvec, const void* Ad, const double* rs, const double* dAd,
                double* rs_new, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!x || !r || !dvec || !Ad || !rs || !dAd || !rs_new || !ws) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    TV_FLAT_LAUNCH(k_cg1, g->dtype, nvox(d), ({x, r, dvec, Ad}), (T*)x, (T*)r, (const T*)dvec, (const T*)Ad, rs, dAd, (double*)ws);
    HIP_TRY(hipGetLastError());
    return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), rs_new, st);
}

int tv_cg_step2(const tv_geom* g, void* dvec, const void* r, const double* rs_new, const double* rs, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!dvec || !r || !rs_new || !rs) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    TV_FLAT_LAUNCH(k_cg2, g->dtype, nvox(d), ({dvec, r}), (T*)dvec, (const T*)r, rs_new, rs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tv_subgrad_step(const tv_geom* g, void* x, const void* x0, const void* G, double step, double lambda, double* fid,
                    void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!x || !x0 || !G || !fid || !ws) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    TV_FLAT_LAUNCH(k_sgstep, g->dtype, nvox(d), ({x, x0, G}), (T*)x, (const T*)x0, (const T*)G, (T)step, (T)lambda, (double*)ws);
    HIP_TRY(hipGetLastError());
    return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), fid, st);
}

// out = A x (b == NULL) or out = b - A x (b != NULL; out2, when given, receives the same vector), A = I + rho D^T D;
// dots[0] = <x, A x> resp. <out, out>, dots[1] = <x, x>  (device fp64, local planes)
int tv_normal_op2(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, double rho, const void* b, void* out,
                  void* out2, double* dots, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || out == nullptr || dots == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (out2 != nullptr && b == nullptr) return fail(TV_E_ARG, "out2 is the copy of the residual: it needs b");
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr)))
        return fail(TV_E_HALO, "tv_normal_op2 on a slab needs two halo planes on each interior side");
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_prev, x_next, out, out2, b, d.wv});
    if (env_int("TV_NORMAL_KERNEL", 2) == 2 && tvm::N_stream_ok(g, d, vec)) {
        long long nb;
        double* w0 = (double*)ws;
        double* w1 = w0 + nmax + kStage + 16;
        if (int rc = tvm::N_stream(g, d, x, x_prev, x_next, b, out, out2, rho, st, &nb, w0, w1)) return rc;
        if (int rc = reduce_partials(w0, nb, nmax, dots, st)) return rc;
        return reduce_partials(w1, nb, nmax, dots + 1, st);
    }
    // composition of the existing entry points (fp64, central, weight volume, small planes)
    if (int rc = tv_normal_op(g, x, x_prev, x_next, rho, out, dots, ws, stream)) return rc;
    if (int rc = tv_dot(g, x, x, dots + 1, ws, stream)) return rc;
    if (b != nullptr) {
        TV_FLAT_LAUNCH(k_sub_dot, g->dtype, nvox(d), ({b, out, out2}), (const T*)b, (const T*)out, (T*)out, (T*)out2, (double*)ws);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, kFlatBlocks, nmax, dots, st);
    }
    return 0;
}

// One Chebyshev step on A e = b, A = I + rho D^T D (include/pytv4d.h): out = [add +] x + alpha (b - A x) + beta (x - y);
// dots[0] = |b - A x|^2, dots[1] = |out - ref|^2 (ref given) or |x|^2.  No scalar of the recurrence depends on the vectors: a
// sharded solve needs halo planes only, no all-reduce.
int tv_cheb_step(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, double rho, const void* b, const void* y,
                 double yscale, const void* add, const void* ref, double alpha, double beta, void* out, double* dots, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || b == nullptr || out == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (ref != nullptr && dots == nullptr) return fail(TV_E_ARG, "ref without a place for |out - ref|^2");
    if (out == x || out == y || out == b || out == add || out == ref) return fail(TV_E_ARG, "out must not alias an input");
    if (y != nullptr && yscale != 0.0) return fail(TV_E_ARG, "yscale is the stand-in for a missing y (y = yscale * b)");
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr)))
        return fail(TV_E_HALO, "tv_cheb_step on a slab needs two halo planes on each interior side");
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_prev, x_next, out, b, y, add, ref, d.wv});
    double* w0 = (double*)ws;
    double* w1 = w0 + nmax + kStage + 16;
    if (env_int("TV_NORMAL_KERNEL", 2) == 2 && tvm::N_stream_ok(g, d, vec)) {
        long long nb;
        const tvm::NCheb c{y, add, ref, alpha, beta, yscale};
        if (int rc = tvm::N_stream(g, d, x, x_prev, x_next, b, out, nullptr, rho, st, &nb, dots ? w0 : nullptr, dots ? w1 : nullptr, &c)) return rc;
        if (dots == nullptr) return 0;
        if (int rc = reduce_partials(w0, nb, nmax, dots, st)) return rc;
        return reduce_partials(w1, nb, nmax, dots + 1, st);
    }
    // composition: out <- A x with the operator of this geometry, then one flat pass
    double* const scratch_dots = w1 + nmax + kStage + 16;        // dots == NULL: the partial sums still exist on this path, their totals go nowhere
    if (dots == nullptr) dots = scratch_dots;
    if (int rc = tv_normal_op(g, x, x_prev, x_next, rho, out, dots, ws, stream)) return rc;
    TV_FLAT_LAUNCH(k_cheb_combine, g->dtype, nvox(d), ({x, b, y, add, ref, out}), (const T*)x, (const T*)b, (const T*)y, (const T*)add,
                   (const T*)ref, (T*)out, (T)alpha, (T)beta, (T)yscale, w0, w1);
    HIP_TRY(hipGetLastError());
    if (int rc = reduce_partials(w0, kFlatBlocks, nmax, dots, st)) return rc;
    return reduce_partials(w1, kFlatBlocks, nmax, dots + 1, st);
}

// out = a x + b y (y == NULL: out = a x); *dist2 (or NULL) = |out - ref|^2; out == NULL (with ref): the distance alone, nothing stored
int tv_axpby(const tv_geom* g, double a, const void* x, double b, const void* y, const void* ref, void* out, double* dist2, void* ws,
             void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (x == nullptr || (out == nullptr && ref == nullptr)) return fail(TV_E_ARG, "NULL array");
    if ((ref != nullptr) != (dist2 != nullptr) || (ref != nullptr && ws == nullptr)) return fail(TV_E_ARG, "ref, dist2 and ws go together");
    hipStream_t st = (hipStream_t)stream;
    TV_FLAT_LAUNCH(k_axpby, g->dtype, nvox(d), ({x, y, ref, out}), (T)a, (const T*)x, (T)b, (const T*)y, (const T*)ref, (T*)out, (double*)ws);
    HIP_TRY(hipGetLastError());
    if (ref != nullptr) return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), dist2, st);
    return 0;
}

// One step of the single-reduction conjugate gradient (Chronopoulos-Gear): see include/pytv4d.h
int tv_cg_update(const tv_geom* g, void* x, void* r, void* dvec, void* s, const void* w, double* sc, const void* x0, double* fid,
                 void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d, true)) return rc;
    if (!x || !r || !dvec || !s || !w || !sc || !ws) return fail(TV_E_ARG, "NULL array");
    if (x0 != nullptr && fid == nullptr) return fail(TV_E_ARG, "x0 without a place for the fidelity");
    hipStream_t st = (hipStream_t)stream;
    TV_FLAT_LAUNCH(k_cgcg, g->dtype, nvox(d), ({x, r, dvec, s, w, x0}), (T*)x, (T*)r, (T*)dvec, (T*)s, (const T*)w, (const double*)sc,
                   (const T*)x0, (double*)ws);
    hipLaunchKernelGGL(k_cgcg_advance, dim3(1), dim3(1), 0, st, sc);
    HIP_TRY(hipGetLastError());
    if (x0 != nullptr) return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), fid, st);
    return 0;
}

}  // extern "C"
