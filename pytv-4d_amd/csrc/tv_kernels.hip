// tv_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the PyTV-4D hot path and the
// C-ABI declared in include/pytv4d.h.
//
//   k_D      : forward operator D (all four schemes) with a fused epilogue
//                StoreD   -> materialise the gradient            (tv_D)
//                NormEpi  -> |Dx|_2 per voxel + TV partial sums  (tv_subgrad pass 1)
//                CpDual   -> q <- proj(q + sigma D x)            (tv_cp_dual)
//                AdmmZU   -> z/u update of ADMM                  (tv_admm_zu)
//   k_DT     : transposed operator as a GATHER (no atomics, no scratch time buffer) with
//                StoreDT / AxpyDT / CpPrimal epilogues           (tv_DT, tv_DT_axpy, tv_cp_primal)
//   k_gather : radius-2 stencils evaluated from x alone          (tv_subgrad pass 2, tv_normal_op)
//   k_l21, k_dot, k_sub, k_cg1, k_cg2, k_sgstep, k_reduce        small streaming / reduction kernels
//
// Per-voxel definitions follow SURVEY 8a-1 / 8a-2, i.e. pytv/tv_operators_CPU.py:117-154,198-218,
// 264-284,330-358 (D) and :398-448,487-516,554-583,622-658 (D^T); the sub-gradient follows
// pytv/tv_CPU.py:91-126,176-190,239-253,302-330.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <initializer_list>
#include <string>
#include <type_traits>

#include "../../include/pytv4d.h"
#include "tv_device.h"

namespace tv {

// =============================================================================================
// neighbourhood of x around one voxel-vector
// =============================================================================================
template <typename T, int V> struct XN {
    Vec<T, V> c;                 // centre
    Vec<T, V> nr, pr;            // next / previous row
    Vec<T, V> nc, pc;            // next / previous column (shifted vectors)
    Vec<T, V> nz, pz;            // next / previous plane
    Vec<T, V> nt, pt;            // next / previous frame
    bool h_nr, h_pr, h_nz, h_pz, h_nt, h_pt;
    int col0;
};

template <typename T, int V, bool NEXT, bool PREV>
__device__ __forceinline__ void load_xn(const DG& g, const T* plane_c, const T* plane_p, const T* plane_n,
                                        const Coord& c, XN<T, V>& o) {
    const long long off = (long long)c.t * g.s_t + (long long)c.y * g.nx + c.col0;
    const T* p = plane_c + off;
    o.c = vload<T, V>(p);
    o.col0 = c.col0;
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    o.nr = o.pr = o.nc = o.pc = o.nz = o.pz = o.nt = o.pt = zero;
    o.h_nr = o.h_pr = o.h_nz = o.h_pz = o.h_nt = o.h_pt = false;
    if (NEXT) {
        o.h_nr = (c.y + 1 < g.ny);
        if (o.h_nr) o.nr = vload<T, V>(p + g.nx);
        const T tail = (c.col0 + V < g.nx) ? p[V] : T(0);
        o.nc = shift_left<T, V>(o.c, tail);
        if (g.za) {
            o.h_nz = (plane_n != nullptr);
            if (o.h_nz) o.nz = vload<T, V>(plane_n + off);
        }
        if (g.ta) {
            o.h_nt = (c.t + 1 < g.m);
            if (o.h_nt) o.nt = vload<T, V>(p + g.s_t);
        }
    }
    if (PREV) {
        o.h_pr = (c.y > 0);
        if (o.h_pr) o.pr = vload<T, V>(p - g.nx);
        const T head = (c.col0 > 0) ? p[-1] : T(0);
        o.pc = shift_right<T, V>(o.c, head);
        if (g.za) {
            o.h_pz = (plane_p != nullptr);
            if (o.h_pz) o.pz = vload<T, V>(plane_p + off);
        }
        if (g.ta) {
            o.h_pt = (c.t > 0);
            if (o.h_pt) o.pt = vload<T, V>(p - g.s_t);
        }
    }
}

// =============================================================================================
// gradient channels of one voxel-vector, in SLOT order
//   non-hybrid: 0 rows, 1 cols, 2 z, 3 t
//   hybrid    : 0 row-up, 1 col-up, 2 row-down, 3 col-down, 4 z-up, 5 z-down, 6 t-up, 7 t-down
// =============================================================================================
template <int S, typename T, int V>
__device__ __forceinline__ void d_slots(const DG& g, const WT<T>& w, const XN<T, V>& n, const Vec<T, V>& mf,
                                        Vec<T, V> (&o)[8]) {
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    Vec<T, V> f_r = zero, f_c = zero, f_z = zero, f_t = zero;   // forward
    Vec<T, V> b_r = zero, b_c = zero, b_z = zero, b_t = zero;   // backward
    Vec<T, V> c_r = zero, c_c = zero, c_z = zero, c_t = zero;   // central
    constexpr bool FW = (S == UPWIND || S == HYBRID || S == CENTRAL);   // central needs fwd for 2-point axes
    constexpr bool BW = (S == DOWNWIND || S == HYBRID);
    if (FW) {
        if (n.h_nr) f_r = n.nr - n.c;
#pragma unroll
        for (int i = 0; i < V; ++i) f_c.v[i] = (n.col0 + i < g.nx - 1) ? n.nc.v[i] - n.c.v[i] : T(0);
        if (n.h_nz) f_z = w.wz * (n.nz - n.c);
        if (n.h_nt) f_t = (w.wt * (n.nt - n.c)) * mf;
    }
    if (BW) {
        if (n.h_pr) b_r = n.c - n.pr;
#pragma unroll
        for (int i = 0; i < V; ++i) b_c.v[i] = (n.col0 + i > 0) ? n.c.v[i] - n.pc.v[i] : T(0);
        if (n.h_pz) b_z = w.wz * (n.c - n.pz);
        if (n.h_pt) b_t = (w.wt * (n.c - n.pt)) * mf;
    }
    if (S == CENTRAL) {
        if (n.h_nr && n.h_pr) c_r = n.nr - n.pr;
#pragma unroll
        for (int i = 0; i < V; ++i)
            c_c.v[i] = (n.col0 + i > 0 && n.col0 + i < g.nx - 1) ? n.nc.v[i] - n.pc.v[i] : T(0);
        if (g.z_two) c_z = f_z;
        else if (n.h_nz && n.h_pz) c_z = w.wz * (n.nz - n.pz);
        if (g.t_two) c_t = f_t;
        else if (n.h_nt && n.h_pt) c_t = (w.wt * (n.nt - n.pt)) * mf;
    }
    if (S == UPWIND) { o[0] = f_r; o[1] = f_c; o[2] = f_z; o[3] = f_t; }
    if (S == DOWNWIND) { o[0] = b_r; o[1] = b_c; o[2] = b_z; o[3] = b_t; }
    if (S == CENTRAL) {
        const T h = T(0.5);
        o[0] = h * c_r; o[1] = h * c_c; o[2] = h * c_z; o[3] = h * c_t;
    }
    if (S == HYBRID) {
        const T s = Consts<T>::inv_sqrt2();
        o[0] = s * f_r; o[1] = s * f_c; o[2] = s * b_r; o[3] = s * b_c;
        o[4] = s * f_z; o[5] = s * b_z; o[6] = s * f_t; o[7] = s * b_t;
    }
    if (S != HYBRID) { o[4] = o[5] = o[6] = o[7] = zero; }
}

template <int I> using IC = std::integral_constant<int, I>;

// f(slot, channel) for every ACTIVE channel; slot is a compile-time constant
template <int S, typename F> __device__ __forceinline__ void for_each_channel(const DG& g, F&& f) {
    f(IC<0>{}, 0);
    f(IC<1>{}, 1);
    if (S == HYBRID) {
        f(IC<2>{}, 2);
        f(IC<3>{}, 3);
        if (g.za) { f(IC<4>{}, g.ch_z); f(IC<5>{}, g.ch_z + 1); }
        if (g.ta) { f(IC<6>{}, g.ch_t); f(IC<7>{}, g.ch_t + 1); }
    } else {
        if (g.za) f(IC<2>{}, g.ch_z);
        if (g.ta) f(IC<3>{}, g.ch_t);
    }
}

template <typename T, int V> __device__ __forceinline__ Vec<T, V> sumsq_slots(const Vec<T, V> (&o)[8]) {
    Vec<T, V> s = vsplat<T, V>(T(0));
#pragma unroll
    for (int k = 0; k < 8; ++k) s = s + o[k] * o[k];   // inactive slots are exactly zero
    return s;
}

// =============================================================================================
// epilogues of the forward kernel
// =============================================================================================
template <int S, typename T, int V> struct StoreD {
    static constexpr bool REDUCES = false;
    T* d;
    double* partials;
    __device__ __forceinline__ double operator()(const DG& g, const Coord& c, const Vec<T, V> (&o)[8]) const {
        T* base = d + (long long)c.zl * g.s_dz + (long long)c.t * g.s_t + (long long)c.y * g.nx + c.col0;
        for_each_channel<S>(g, [&](auto slot, int ch) { vstore<T, V>(base + (long long)ch * g.s_z, o[decltype(slot)::value]); });
        return 0.0;
    }
};

// |Dx| per voxel (0 -> +inf, pytv/tv_GPU.py:88) into an array with one extra plane in front
template <int S, typename T, int V> struct NormEpi {
    static constexpr bool REDUCES = true;
    T* norms_ext;
    double* partials;
    __device__ __forceinline__ double operator()(const DG& g, const Coord& c, const Vec<T, V> (&o)[8]) const {
        const Vec<T, V> s = sumsq_slots<T, V>(o);
        Vec<T, V> n;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const T r = tsqrt(s.v[i]);
            acc += (double)r;
            n.v[i] = (r == T(0)) ? (T)INFINITY : r;
        }
        vstore<T, V>(norms_ext + (long long)(c.zl + 1) * g.s_z + (long long)c.t * g.s_t + (long long)c.y * g.nx + c.col0, n);
        return (c.zl >= 0 && c.zl < g.nz) ? acc : 0.0;
    }
};

// Chambolle-Pock dual update, README.md:149-151 (with keepdims over the channel axis)
template <int S, typename T, int V> struct CpDual {
    static constexpr bool REDUCES = true;
    T* q;
    T sigma, inv_lambda;
    double* partials;
    __device__ __forceinline__ double operator()(const DG& g, const Coord& c, const Vec<T, V> (&o)[8]) const {
        T* base = q + (long long)c.zl * g.s_dz + (long long)c.t * g.s_t + (long long)c.y * g.nx + c.col0;
        Vec<T, V> v[8];
        Vec<T, V> vs = vsplat<T, V>(T(0));
        for_each_channel<S>(g, [&](auto slot, int ch) {
            constexpr int k = decltype(slot)::value;
            v[k] = vload<T, V>(base + (long long)ch * g.s_z) + sigma * o[k];
            vs = vs + v[k] * v[k];
        });
        const Vec<T, V> ds = sumsq_slots<T, V>(o);
        Vec<T, V> scale;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            acc += (double)tsqrt(ds.v[i]);
            scale.v[i] = T(1) / tmax(T(1), tsqrt(vs.v[i]) * inv_lambda);
        }
        for_each_channel<S>(g, [&](auto slot, int ch) {
            constexpr int k = decltype(slot)::value;
            vstore<T, V>(base + (long long)ch * g.s_z, v[k] * scale);
        });
        return acc;
    }
};

// ADMM: v = Dx + u;  z = v * max(0, 1 - thresh/|v|);  u = v - z
template <int S, typename T, int V> struct AdmmZU {
    static constexpr bool REDUCES = true;
    T* z;
    T* u;
    T thresh;
    double* partials;
    __device__ __forceinline__ double operator()(const DG& g, const Coord& c, const Vec<T, V> (&o)[8]) const {
        const long long off = (long long)c.zl * g.s_dz + (long long)c.t * g.s_t + (long long)c.y * g.nx + c.col0;
        Vec<T, V> v[8];
        Vec<T, V> vs = vsplat<T, V>(T(0));
        for_each_channel<S>(g, [&](auto slot, int ch) {
            constexpr int k = decltype(slot)::value;
            v[k] = o[k] + vload<T, V>(u + off + (long long)ch * g.s_z);
            vs = vs + v[k] * v[k];
        });
        const Vec<T, V> ds = sumsq_slots<T, V>(o);
        Vec<T, V> scale;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            acc += (double)tsqrt(ds.v[i]);
            const T nv = tsqrt(vs.v[i]);
            scale.v[i] = (nv > T(0)) ? tmax(T(0), T(1) - thresh / nv) : T(0);
        }
        for_each_channel<S>(g, [&](auto slot, int ch) {
            constexpr int k = decltype(slot)::value;
            const Vec<T, V> zz = v[k] * scale;
            vstore<T, V>(z + off + (long long)ch * g.s_z, zz);
            vstore<T, V>(u + off + (long long)ch * g.s_z, v[k] - zz);
        });
        return acc;
    }
};

// =============================================================================================
// forward kernel
// =============================================================================================
template <int S, typename T, int V, typename Epi>
__global__ __launch_bounds__(256) void k_D(DG g, WT<T> w, const T* x, const T* xp, const T* xn, int hp, int z_first, Epi epi) {
    __shared__ double sm[16];
    Coord c = thread_coord<V>(g, z_first);
    double acc = 0.0;
    const T* pc = zplane<T>(g, x, xp, xn, hp, c.zl);
    if (c.ok && pc != nullptr) {
        constexpr bool NEXT = (S != DOWNWIND), PREV = (S != UPWIND);
        const T* pp = nullptr;
        const T* pn = nullptr;
        if (g.za) {
            if (PREV) pp = zplane<T>(g, x, xp, xn, hp, c.zl - 1);
            if (NEXT) pn = zplane<T>(g, x, xp, xn, hp, c.zl + 1);
        }
        XN<T, V> n;
        load_xn<T, V, NEXT, PREV>(g, pc, pp, pn, c, n);
        const Vec<T, V> mf = g.ta ? mask_factor<T, V>(g, w.sf, c.y, c.col0) : vsplat<T, V>(T(1));
        Vec<T, V> o[8];
        d_slots<S, T, V>(g, w, n, mf, o);
        acc = epi(g, c, o);
    }
    if (Epi::REDUCES) {
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) epi.partials[linear_block_id()] = acc;
    }
}

// =============================================================================================
// transposed operator: sources and epilogues
// =============================================================================================
template <typename T, int V> struct SrcPlain {
    const T* y;
    const T* yp;     // halo plane z0-1 of the backward-looking z channel
    const T* yn;     // halo plane z0+nz of the forward-looking z channel
    __device__ __forceinline__ Vec<T, V> ld(long long off) const { return vload<T, V>(y + off); }
    __device__ __forceinline__ T lds(long long off) const { return y[off]; }
    __device__ __forceinline__ Vec<T, V> ldp(long long off) const { return vload<T, V>(yp + off); }
    __device__ __forceinline__ Vec<T, V> ldn(long long off) const { return vload<T, V>(yn + off); }
};
template <typename T, int V> struct SrcDiff {    // a - b, halos already differenced
    const T* a;
    const T* b;
    const T* yp;
    const T* yn;
    __device__ __forceinline__ Vec<T, V> ld(long long off) const { return vload<T, V>(a + off) - vload<T, V>(b + off); }
    __device__ __forceinline__ T lds(long long off) const { return a[off] - b[off]; }
    __device__ __forceinline__ Vec<T, V> ldp(long long off) const { return vload<T, V>(yp + off); }
    __device__ __forceinline__ Vec<T, V> ldn(long long off) const { return vload<T, V>(yn + off); }
};

template <typename T, int V> struct StoreDT {
    static constexpr bool REDUCES = false;
    T* out;
    double* partials;
    __device__ __forceinline__ double operator()(long long off, const Vec<T, V>& r) const {
        vstore<T, V>(out + off, r);
        return 0.0;
    }
};
template <typename T, int V> struct AxpyDT {     // out = base + alpha * r
    static constexpr bool REDUCES = false;
    T* out;
    const T* base;
    T alpha;
    double* partials;
    __device__ __forceinline__ double operator()(long long off, const Vec<T, V>& r) const {
        Vec<T, V> b = (base != nullptr) ? vload<T, V>(base + off) : vsplat<T, V>(T(0));
        vstore<T, V>(out + off, b + alpha * r);
        return 0.0;
    }
};
// Chambolle-Pock primal step with the fidelity-dual update folded in, README.md:148,154,157
template <typename T, int V> struct CpPrimal {
    static constexpr bool REDUCES = true;
    T* x;
    const T* x0;
    T* p;
    T tau, sigma_a, inv_1p_sigma_a;
    double* partials;
    __device__ __forceinline__ double operator()(long long off, const Vec<T, V>& r) const {
        const Vec<T, V> xv = vload<T, V>(x + off), x0v = vload<T, V>(x0 + off), pv = vload<T, V>(p + off);
        Vec<T, V> pn, xn;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            pn.v[i] = (pv.v[i] + sigma_a * (xv.v[i] - x0v.v[i])) * inv_1p_sigma_a;
            xn.v[i] = (xv.v[i] - tau * pn.v[i]) - tau * r.v[i];
            const double e = (double)xn.v[i] - (double)x0v.v[i];
            acc += 0.5 * e * e;
        }
        vstore<T, V>(p + off, pn);
        vstore<T, V>(x + off, xn);
        return acc;
    }
};

// One axis of the gather.  MODE 0: y^(p-e) - y^(p)   (adjoint of a forward difference)
//                          MODE 1: y^(p) - y^(p+e)   (adjoint of a backward difference)
//                          MODE 2: y^(p-e) - y^(p+e) (adjoint of a central difference)
// y^ = y with the samples the forward operator never writes forced to zero.
// lo/hi are the loaded neighbour vectors; pos/n the coordinate along the axis and its extent.
template <int MODE, typename T, int V>
__device__ __forceinline__ Vec<T, V> adj_axis(int pos, int n, const Vec<T, V>& lo, const Vec<T, V>& ce, const Vec<T, V>& hi) {
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    if (MODE == 0) return ((pos >= 1) ? lo : zero) - ((pos <= n - 2) ? ce : zero);
    if (MODE == 1) return ((pos >= 1) ? ce : zero) - ((pos <= n - 2) ? hi : zero);
    return ((pos >= 2) ? lo : zero) - ((pos <= n - 3) ? hi : zero);
}

template <int S, typename T, int V, typename Src, typename Epi>
__global__ __launch_bounds__(256) void k_DT(DG g, WT<T> w, Src src, Epi epi) {
    __shared__ double sm[16];
    const Coord c = thread_coord<V>(g, 0);
    double acc = 0.0;
    if (c.ok) {
        const Vec<T, V> zero = vsplat<T, V>(T(0));
        const long long inpl = (long long)c.t * g.s_t + (long long)c.y * g.nx + c.col0;   // offset inside a plane
        const long long offd = (long long)c.zl * g.s_dz + inpl;                          // channel 0 of this voxel
        const int gz = g.z0 + c.zl;
        Vec<T, V> r = zero, rt = zero;

        // ---- one (channel, mode) at a time; MODE as in adj_axis -------------------------------
        auto rows = [&](auto mode, int ch) {
            constexpr int M = decltype(mode)::value;
            const long long o = offd + (long long)ch * g.s_z;
            const Vec<T, V> lo = (c.y >= 1 && M != 1) ? src.ld(o - g.nx) : zero;
            const Vec<T, V> ce = (M != 2) ? src.ld(o) : zero;
            const Vec<T, V> hi = (c.y + 1 < g.ny && M != 0) ? src.ld(o + g.nx) : zero;
            r = r + adj_axis<M, T, V>(c.y, g.ny, lo, ce, hi);
        };
        auto cols = [&](auto mode, int ch) {
            constexpr int M = decltype(mode)::value;
            const long long o = offd + (long long)ch * g.s_z;
            const Vec<T, V> ce = src.ld(o);
            const T head = (c.col0 > 0) ? src.lds(o - 1) : T(0);
            const T tail = (c.col0 + V < g.nx) ? src.lds(o + V) : T(0);
            // neighbours one column away; for the central adjoint they are what is needed directly
            const Vec<T, V> lo = shift_right<T, V>(ce, head);
            const Vec<T, V> hi = shift_left<T, V>(ce, tail);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int col = c.col0 + i;
                T a, b;
                if (M == 0) { a = (col >= 1) ? lo.v[i] : T(0); b = (col <= g.nx - 2) ? ce.v[i] : T(0); }
                else if (M == 1) { a = (col >= 1) ? ce.v[i] : T(0); b = (col <= g.nx - 2) ? hi.v[i] : T(0); }
                else { a = (col >= 2) ? lo.v[i] : T(0); b = (col <= g.nx - 3) ? hi.v[i] : T(0); }
                r.v[i] += a - b;
            }
        };
        auto zax = [&](auto mode, int ch) {
            constexpr int M = decltype(mode)::value;
            const long long o = offd + (long long)ch * g.s_z;
            Vec<T, V> lo = zero, hi = zero;
            if (M != 1 && gz >= 1) lo = (c.zl >= 1) ? src.ld(o - g.s_dz) : src.ldp(inpl);
            if (M != 0 && gz + 1 < g.nzg) hi = (c.zl + 1 < g.nz) ? src.ld(o + g.s_dz) : src.ldn(inpl);
            const Vec<T, V> ce = (M != 2) ? src.ld(o) : zero;
            r = r + w.wz * adj_axis<M, T, V>(gz, g.nzg, lo, ce, hi);
        };
        auto tax = [&](auto mode, int ch) {
            constexpr int M = decltype(mode)::value;
            const long long o = offd + (long long)ch * g.s_z;
            const Vec<T, V> lo = (c.t >= 1 && M != 1) ? src.ld(o - g.s_t) : zero;
            const Vec<T, V> ce = (M != 2) ? src.ld(o) : zero;
            const Vec<T, V> hi = (c.t + 1 < g.m && M != 0) ? src.ld(o + g.s_t) : zero;
            rt = rt + w.wt * adj_axis<M, T, V>(c.t, g.m, lo, ce, hi);
        };

        if (S == UPWIND) {
            rows(IC<0>{}, 0); cols(IC<0>{}, 1);
            if (g.za) zax(IC<0>{}, g.ch_z);
            if (g.ta) tax(IC<0>{}, g.ch_t);
        } else if (S == DOWNWIND) {
            rows(IC<1>{}, 0); cols(IC<1>{}, 1);
            if (g.za) zax(IC<1>{}, g.ch_z);
            if (g.ta) tax(IC<1>{}, g.ch_t);
        } else if (S == CENTRAL) {
            rows(IC<2>{}, 0); cols(IC<2>{}, 1);
            if (g.za) { if (g.z_two) zax(IC<0>{}, g.ch_z); else zax(IC<2>{}, g.ch_z); }
            if (g.ta) { if (g.t_two) tax(IC<0>{}, g.ch_t); else tax(IC<2>{}, g.ch_t); }
        } else {
            rows(IC<0>{}, 0); cols(IC<0>{}, 1); rows(IC<1>{}, 2); cols(IC<1>{}, 3);
            if (g.za) { zax(IC<0>{}, g.ch_z); zax(IC<1>{}, g.ch_z + 1); }
            if (g.ta) { tax(IC<0>{}, g.ch_t); tax(IC<1>{}, g.ch_t + 1); }
        }
        if (g.ta) {
            // mask_static scales only the time contribution, at the output voxel
            // (pytv/tv_operators_CPU.py:442-446)
            const Vec<T, V> mf = mask_factor<T, V>(g, w.sf, c.y, c.col0);
            r = r + rt * mf;
        }
        if (S == HYBRID) r = Consts<T>::inv_sqrt2() * r;
        if (S == CENTRAL) r = T(0.5) * r;
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
        return p ? p[(long long)t * g.s_t + (long long)y * g.nx + c] : T(0);
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
        if (X.g.mask != nullptr && X.g.mask[(long long)y * X.g.nx + c]) diff *= w.sf;
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
                const T n = norms_ext[(long long)(qz + 1) * g.s_z + (long long)qt * g.s_t + (long long)qy * g.nx + qc];
                return d / n;           // n == +inf where |Dx| == 0  ->  exactly 0
            }
            return d;
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
        if (MODE == 1 && g.ta && g.mask != nullptr && g.mask[(long long)y * g.nx + col]) rt *= w.sf;
        r += rt;
        if (S == HYBRID) r *= Consts<T>::inv_sqrt2();
        if (S == CENTRAL) r *= T(0.5);
        const long long off = (long long)zl * g.s_z + (long long)t * g.s_t + (long long)y * g.nx + col;
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
// l2,1 norm of a materialised gradient (pytv/tv_operators_GPU.py:75-81 as ONE pass)
// =============================================================================================
template <typename T, int V>
__global__ __launch_bounds__(256) void k_l21(DG g, const T* d, T* norms, double* partials) {
    __shared__ double sm[16];
    const Coord c = thread_coord<V>(g, 0);
    double acc = 0.0;
    if (c.ok) {
        const long long inpl = (long long)c.t * g.s_t + (long long)c.y * g.nx + c.col0;
        const T* base = d + (long long)c.zl * g.s_dz + inpl;
        Vec<T, V> s = vsplat<T, V>(T(0));
        for (int ch = 0; ch < g.nd; ++ch) {
            const Vec<T, V> v = vload<T, V>(base + (long long)ch * g.s_z);
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

// =============================================================================================
// flat streaming kernels (grid-stride, one element per lane per trip; n = nz*m*ny*nx)
// =============================================================================================
template <typename T> __global__ __launch_bounds__(256) void k_sub(long long n, const T* a, const T* b, T* out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a[i] - b[i];
}
template <typename T> __global__ __launch_bounds__(256) void k_dot(long long n, const T* a, const T* b, double* partials) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        acc += (double)a[i] * (double)b[i];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// alpha = rs/dAd;  x += alpha d;  r -= alpha Ad;  partial <r, r>
template <typename T>
__global__ __launch_bounds__(256) void k_cg1(long long n, T* x, T* r, const T* d, const T* Ad, const double* rs,
                                              const double* dAd, double* partials) {
    __shared__ double sm[16];
    const double den = *dAd;
    const T alpha = (den > 0.0) ? (T)(*rs / den) : T(0);
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        x[i] = x[i] + alpha * d[i];
        const T rn = r[i] - alpha * Ad[i];
        r[i] = rn;
        acc += (double)rn * (double)rn;
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}
// beta = rs_new/rs;  d = r + beta d
template <typename T>
__global__ __launch_bounds__(256) void k_cg2(long long n, T* d, const T* r, const double* rs_new, const double* rs) {
    const double den = *rs;
    const T beta = (den > 0.0) ? (T)(*rs_new / den) : T(0);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        d[i] = r[i] + beta * d[i];
}
// README.md:122-123: x <- x - step * ((x - x0) + lambda G); partial 1/2 |x - x0|^2
template <typename T>
__global__ __launch_bounds__(256) void k_sgstep(long long n, T* x, const T* x0, const T* G, T step, T lambda, double* partials) {
    __shared__ double sm[16];
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const T xv = x[i], x0v = x0[i];
        const T xn = xv - step * ((xv - x0v) + lambda * G[i]);
        x[i] = xn;
        const double e = (double)xn - (double)x0v;
        acc += 0.5 * e * e;
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

// =============================================================================================
// host side: argument checking, launch geometry, C-ABI
// =============================================================================================
using namespace tv;

static thread_local std::string g_err;
static int fail(int code, const char* msg) {
    g_err = msg;
    return code;
}
static int hipfail(hipError_t e, const char* where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return (int)e;
}
#define HIP_TRY(call)                                         \
    do {                                                      \
        hipError_t e__ = (call);                              \
        if (e__ != hipSuccess) return hipfail(e__, #call);    \
    } while (0)

static int make_dg(const tv_geom* g, DG& d) {
    if (g == nullptr) return fail(TV_E_ARG, "tv_geom is NULL");
    if (g->nz < 1 || g->m < 1 || g->ny < 1 || g->nx < 1) return fail(TV_E_ARG, "every dimension must be >= 1");
    if (g->nz_global < g->nz || g->z0 < 0 || g->z0 + g->nz > g->nz_global)
        return fail(TV_E_ARG, "slab [z0, z0+nz) must lie inside [0, nz_global)");
    if (g->scheme < 0 || g->scheme > 3) return fail(TV_E_ARG, "unknown scheme");
    if (g->dtype != TV_F32 && g->dtype != TV_F64) return fail(TV_E_ARG, "unknown dtype");
    if (g->nz_global > 60000 || g->m > 65535 || g->ny > (1 << 24) || g->nx > (1 << 24))
        return fail(TV_E_ARG, "dimension too large for the launch grid");
    if (!(g->reg_z_over_reg >= 0.0) || !(g->reg_time >= 0.0) || !(g->factor_reg_static >= 0.0))
        return fail(TV_E_ARG, "weights must be non-negative numbers");
    d.nz = (int)g->nz; d.m = (int)g->m; d.ny = (int)g->ny; d.nx = (int)g->nx;
    d.nzg = (int)g->nz_global; d.z0 = (int)g->z0;
    d.za = (g->nz_global > 1 && g->reg_z_over_reg > 0.0) ? 1 : 0;
    d.ta = (g->m > 1 && g->reg_time > 0.0) ? 1 : 0;
    const int per = (g->scheme == TV_HYBRID) ? 2 : 1;
    d.nd = per * (2 + d.za + d.ta);
    d.ch_z = 2 * per;
    d.ch_t = d.ch_z + (d.za ? per : 0);
    d.z_two = (g->scheme == TV_CENTRAL && g->nz_global == 2) ? 1 : 0;
    d.t_two = (g->scheme == TV_CENTRAL && g->m == 2) ? 1 : 0;
    d.s_t = (long long)g->ny * g->nx;
    d.s_z = d.s_t * g->m;
    d.s_dz = d.s_z * d.nd;
    d.mask = g->mask_static;
    return 0;
}

template <typename T> static WT<T> make_w(const tv_geom* g) {
    WT<T> w;
    w.wz = (T)std::sqrt(g->reg_z_over_reg);
    w.wt = (T)std::sqrt(g->reg_time);
    w.sf = (T)std::sqrt(g->factor_reg_static);
    return w;
}

struct LC { dim3 grid, block; long long nblocks; };
static LC launch_cfg(const DG& d, int V, int planes) {
    const int nxv = d.nx / V;
    int bx = 1;
    while (bx < nxv && bx < 64) bx <<= 1;
    const int by = 256 / bx;
    const long long tx = (nxv + bx - 1) / bx, ty = (d.ny + by - 1) / by;
    LC lc;
    lc.block = dim3(bx, by, 1);
    lc.grid = dim3((unsigned)(tx * ty), (unsigned)d.m, (unsigned)planes);
    lc.nblocks = tx * ty * d.m * planes;
    return lc;
}

static const int kFlatBlocks = 2048;    // grid-stride kernels: 256 CUs x 8 blocks
static const int kStage = 256;          // second-level partials

// layout of the scratch buffer: [partials ... nmax][stage kStage]
static long long max_partials(const DG& d) {
    LC lc = launch_cfg(d, 1, d.nz + 2);
    return lc.nblocks > kFlatBlocks ? lc.nblocks : kFlatBlocks;
}

static int reduce_partials(double* ws, long long n, long long nmax, double* result, hipStream_t st) {
    if (n <= 4096) {
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, st, ws, n, result);
    } else {
        double* stage = ws + nmax;
        hipLaunchKernelGGL(k_reduce, dim3(kStage), dim3(256), 0, st, ws, n, stage);
        hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), 0, st, stage, (long long)kStage, result);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

static bool aligned16(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (p != nullptr && (reinterpret_cast<uintptr_t>(p) & 15u) != 0) return false;
    return true;
}

// call f.template operator()<S, T, V>() for the run-time (scheme, dtype, vec)
template <typename F> static int dispatch(int scheme, int dtype, bool vec, F&& f) {
#define TV_CASE(SC)                                                                  \
    case SC:                                                                         \
        if (dtype == TV_F32) {                                                       \
            if (vec) return f.template operator()<SC, float, 4>();                   \
            return f.template operator()<SC, float, 1>();                            \
        }                                                                            \
        return f.template operator()<SC, double, 1>();
    switch (scheme) {
        TV_CASE(0) TV_CASE(1) TV_CASE(2) TV_CASE(3)
    }
#undef TV_CASE
    return fail(TV_E_ARG, "unknown scheme");
}

static int check_x_halos(const tv_geom* g, const DG& d, const void* xp, const void* xn) {
    if (!d.za) return 0;
    const bool need_prev = (g->scheme != TV_UPWIND), need_next = (g->scheme != TV_DOWNWIND);
    if (need_prev && g->z0 > 0 && xp == nullptr) return fail(TV_E_HALO, "previous-slab halo plane required");
    if (need_next && g->z0 + g->nz < g->nz_global && xn == nullptr) return fail(TV_E_HALO, "next-slab halo plane required");
    return 0;
}
static int check_y_halos(const tv_geom* g, const DG& d, const void* yp, const void* yn) {
    if (!d.za) return 0;
    // backward-looking adjoint (upwind-type, central) reads the previous slab; forward-looking the next
    const bool need_prev = (g->scheme != TV_DOWNWIND), need_next = (g->scheme != TV_UPWIND);
    if (need_prev && g->z0 > 0 && yp == nullptr) return fail(TV_E_HALO, "previous-slab gradient halo required");
    if (need_next && g->z0 + g->nz < g->nz_global && yn == nullptr) return fail(TV_E_HALO, "next-slab gradient halo required");
    return 0;
}

extern "C" {

const char* tv_last_error(void) { return g_err.c_str(); }
int tv_version(void) { return 100; }

int tv_num_channels(const tv_geom* g) {
    DG d;
    int rc = make_dg(g, d);
    return rc ? rc : d.nd;
}

size_t tv_workspace_bytes(const tv_geom* g) {
    DG d;
    if (make_dg(g, d)) return 0;
    return (size_t)(max_partials(d) + kStage + 16) * sizeof(double);
}

// ---------------------------------------------------------------------------------------------
int tv_D(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* dout, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (x == nullptr || dout == nullptr) return fail(TV_E_ARG, "NULL array");
    if (int rc = check_x_halos(g, d, x_prev, x_next)) return rc;
    const bool vec = (d.nx % 4 == 0) && aligned16({x, x_prev, x_next, dout});
    hipStream_t st = (hipStream_t)stream;
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
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (a == nullptr || out == nullptr) return fail(TV_E_ARG, "NULL array");
    if (int rc = check_y_halos(g, d, ab_prev, ab_next)) return rc;
    const bool vec = (d.nx % 4 == 0) && aligned16({a, b, ab_prev, ab_next, base, out});
    hipStream_t st = (hipStream_t)stream;
    const bool plain_store = (base == nullptr && alpha == 1.0);
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        WT<T> w = make_w<T>(g);
        if (b == nullptr) {
            SrcPlain<T, V> src{(const T*)a, (const T*)ab_prev, (const T*)ab_next};
            if (plain_store) {
                StoreDT<T, V> epi{(T*)out, nullptr};
                hipLaunchKernelGGL((k_DT<S, T, V, SrcPlain<T, V>, StoreDT<T, V>>), lc.grid, lc.block, 0, st, d, w, src, epi);
            } else {
                AxpyDT<T, V> epi{(T*)out, (const T*)base, (T)alpha, nullptr};
                hipLaunchKernelGGL((k_DT<S, T, V, SrcPlain<T, V>, AxpyDT<T, V>>), lc.grid, lc.block, 0, st, d, w, src, epi);
            }
        } else {
            SrcDiff<T, V> src{(const T*)a, (const T*)b, (const T*)ab_prev, (const T*)ab_next};
            AxpyDT<T, V> epi{(T*)out, (const T*)base, (T)alpha, nullptr};
            hipLaunchKernelGGL((k_DT<S, T, V, SrcDiff<T, V>, AxpyDT<T, V>>), lc.grid, lc.block, 0, st, d, w, src, epi);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int tv_l21(const tv_geom* g, const void* dimg, int32_t nd, void* norms, double* result, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (dimg == nullptr || result == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (nd < 1) return fail(TV_E_CHANNELS, "nd must be >= 1");
    // the l2,1 norm does not care which scheme produced the channels: honour the caller's nd
    d.nd = nd;
    d.s_dz = d.s_z * nd;
    const bool vec = (d.nx % 4 == 0) && aligned16({dimg, norms});
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
    if (int rc = make_dg(g, d)) return rc;
    if (x == nullptr || G == nullptr || norms_ext == nullptr || tvout == nullptr || ws == nullptr)
        return fail(TV_E_ARG, "NULL array");
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr)))
        return fail(TV_E_HALO, "tv_subgrad on a slab needs two halo planes on each interior side");
    const bool vec = (d.nx % 4 == 0) && aligned16({x, x_prev, x_next, norms_ext});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
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
        // pass 2: gather
        LC lg = launch_cfg(d, 1, d.nz);
        XA<T> X{d, (const T*)x, (const T*)x_prev, (const T*)x_next, 2};
        hipLaunchKernelGGL((k_gather<S, T, 0>), lg.grid, lg.block, 0, st, X, w, (const T*)norms_ext, T(0), (T*)G, (double*)nullptr);
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int tv_normal_op(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, double rho, void* out,
                 double* dot, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (x == nullptr || out == nullptr || dot == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    const int e_lo = (g->z0 > 0) ? 1 : 0, e_hi = (g->z0 + g->nz < g->nz_global) ? 1 : 0;
    if (d.za && ((e_lo && x_prev == nullptr) || (e_hi && x_next == nullptr)))
        return fail(TV_E_HALO, "tv_normal_op on a slab needs two halo planes on each interior side");
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    return dispatch(g->scheme, g->dtype, false, [&]<int S, typename T, int V>() -> int {
        LC lg = launch_cfg(d, 1, d.nz);
        XA<T> X{d, (const T*)x, (const T*)x_prev, (const T*)x_next, 2};
        hipLaunchKernelGGL((k_gather<S, T, 1>), lg.grid, lg.block, 0, st, X, make_w<T>(g), (const T*)nullptr, (T)rho, (T*)out,
                           (double*)ws);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lg.nblocks, nmax, dot, st);
    });
}

// ---------------------------------------------------------------------------------------------
int tv_cp_dual(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* q, double sigma_D,
               double lambda, double* tvout, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (x == nullptr || q == nullptr || tvout == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (!(lambda > 0.0)) return fail(TV_E_ARG, "lambda must be > 0");
    if (int rc = check_x_halos(g, d, x_prev, x_next)) return rc;
    const bool vec = (d.nx % 4 == 0) && aligned16({x, x_prev, x_next, q});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
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
    if (int rc = make_dg(g, d)) return rc;
    if (q == nullptr || x == nullptr || x0 == nullptr || p == nullptr || fid == nullptr || ws == nullptr)
        return fail(TV_E_ARG, "NULL array");
    if (int rc = check_y_halos(g, d, q_prev, q_next)) return rc;
    const bool vec = (d.nx % 4 == 0) && aligned16({q, q_prev, q_next, x, x0, p});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        SrcPlain<T, V> src{(const T*)q, (const T*)q_prev, (const T*)q_next};
        CpPrimal<T, V> epi{(T*)x, (const T*)x0, (T*)p, (T)tau, (T)sigma_A, (T)(1.0 / (1.0 + sigma_A)), (double*)ws};
        hipLaunchKernelGGL((k_DT<S, T, V, SrcPlain<T, V>, CpPrimal<T, V>>), lc.grid, lc.block, 0, st, d, make_w<T>(g), src, epi);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lc.nblocks, nmax, fid, st);
    });
}

int tv_admm_zu(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* z, void* u,
               double thresh, double* tvout, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (x == nullptr || z == nullptr || u == nullptr || tvout == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (int rc = check_x_halos(g, d, x_prev, x_next)) return rc;
    const bool vec = (d.nx % 4 == 0) && aligned16({x, x_prev, x_next, z, u});
    hipStream_t st = (hipStream_t)stream;
    const long long nmax = max_partials(d);
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        LC lc = launch_cfg(d, V, d.nz);
        AdmmZU<S, T, V> epi{(T*)z, (T*)u, (T)thresh, (double*)ws};
        hipLaunchKernelGGL((k_D<S, T, V, AdmmZU<S, T, V>>), lc.grid, lc.block, 0, st, d, make_w<T>(g), (const T*)x,
                           (const T*)x_prev, (const T*)x_next, 1, 0, epi);
        HIP_TRY(hipGetLastError());
        return reduce_partials((double*)ws, lc.nblocks, nmax, tvout, st);
    });
}

// ---------------------------------------------------------------------------------------------
static long long nvox(const DG& d) { return d.s_z * d.nz; }

int tv_sub(int32_t dtype, int64_t n, const void* a, const void* b, void* out, void* stream) {
    if (n < 0 || a == nullptr || b == nullptr || out == nullptr) return fail(TV_E_ARG, "bad argument");
    if (dtype != TV_F32 && dtype != TV_F64) return fail(TV_E_ARG, "unknown dtype");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return 0;
    if (dtype == TV_F32) hipLaunchKernelGGL(k_sub<float>, dim3(kFlatBlocks), dim3(256), 0, st, (long long)n, (const float*)a, (const float*)b, (float*)out);
    else hipLaunchKernelGGL(k_sub<double>, dim3(kFlatBlocks), dim3(256), 0, st, (long long)n, (const double*)a, (const double*)b, (double*)out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tv_dot(const tv_geom* g, const void* a, const void* b, double* result, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (a == nullptr || b == nullptr || result == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    if (g->dtype == TV_F32) hipLaunchKernelGGL(k_dot<float>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (const float*)a, (const float*)b, (double*)ws);
    else hipLaunchKernelGGL(k_dot<double>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (const double*)a, (const double*)b, (double*)ws);
    HIP_TRY(hipGetLastError());
    return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), result, st);
}

int tv_cg_step1(const tv_geom* g, void* x, void* r, const void* dvec, const void* Ad, const double* rs, const double* dAd,
                double* rs_new, void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (!x || !r || !dvec || !Ad || !rs || !dAd || !rs_new || !ws) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    if (g->dtype == TV_F32) hipLaunchKernelGGL(k_cg1<float>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (float*)x, (float*)r, (const float*)dvec, (const float*)Ad, rs, dAd, (double*)ws);
    else hipLaunchKernelGGL(k_cg1<double>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (double*)x, (double*)r, (const double*)dvec, (const double*)Ad, rs, dAd, (double*)ws);
    HIP_TRY(hipGetLastError());
    return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), rs_new, st);
}

int tv_cg_step2(const tv_geom* g, void* dvec, const void* r, const double* rs_new, const double* rs, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (!dvec || !r || !rs_new || !rs) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    if (g->dtype == TV_F32) hipLaunchKernelGGL(k_cg2<float>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (float*)dvec, (const float*)r, rs_new, rs);
    else hipLaunchKernelGGL(k_cg2<double>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (double*)dvec, (const double*)r, rs_new, rs);
    HIP_TRY(hipGetLastError());
    return 0;
}

int tv_subgrad_step(const tv_geom* g, void* x, const void* x0, const void* G, double step, double lambda, double* fid,
                    void* ws, void* stream) {
    DG d;
    if (int rc = make_dg(g, d)) return rc;
    if (!x || !x0 || !G || !fid || !ws) return fail(TV_E_ARG, "NULL array");
    hipStream_t st = (hipStream_t)stream;
    if (g->dtype == TV_F32) hipLaunchKernelGGL(k_sgstep<float>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (float*)x, (const float*)x0, (const float*)G, (float)step, (float)lambda, (double*)ws);
    else hipLaunchKernelGGL(k_sgstep<double>, dim3(kFlatBlocks), dim3(256), 0, st, nvox(d), (double*)x, (const double*)x0, (const double*)G, step, lambda, (double*)ws);
    HIP_TRY(hipGetLastError());
    return reduce_partials((double*)ws, kFlatBlocks, max_partials(d), fid, st);
}

}  // extern "C"
