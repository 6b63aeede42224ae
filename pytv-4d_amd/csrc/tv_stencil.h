// tv_stencil.h -- per-site stencil arithmetic shared by the one-site-per-thread kernels
// (tv_kernels.hip) and the plane-marching kernels (tv_march.h): the neighbourhood struct, the
// gradient channels of one site for the four schemes, and the fused epilogues.
//
// Per-voxel definitions follow SURVEY 8a-1 / 8a-2, i.e. pytv/tv_operators_CPU.py:117-154,198-218,
// 264-284,330-358 (D) and :398-448,487-516,554-583,622-658 (D^T).
#pragma once
#include <cmath>
#include <type_traits>

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

// How a kernel reaches an array.  PlainMem: ordinary loads / stores (every kernel whose inputs were complete before it started).
// CohMem (tv_small.hip, the persistent small-volume kernels): the array is written by OTHER blocks of the SAME launch -- blocks that may sit
// on another XCD with its own L2 -- so every access is an agent-coherent (sc1) raw-buffer access relative to the array's base
// (profiles/r6_coherence_probe.txt: sc1 store + sc1 load is never stale, plain loads are; repeated sc1 loads of a line hit the L2).
struct PlainMem {
    template <typename T, int V> __device__ __forceinline__ Vec<T, V> ld(const T* p) const { return vload<T, V>(p); }
    template <typename T> __device__ __forceinline__ T ld1(const T* p) const { return *p; }
    template <typename T, int V> __device__ __forceinline__ void st(T* p, const Vec<T, V>& v) const { vstore<T, V>(p, v); }
};
struct CohMem {
    __amdgpu_buffer_rsrc_t r;
    const char* base;
    static constexpr int AUX = 16;                   // sc1 on gfx942 / gfx950
    template <typename T> __device__ __forceinline__ static CohMem make(const T* b, long long nbytes) {
        CohMem m;
        m.r = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)(nbytes > 0x7fffffffll ? 0x7fffffffll : nbytes), 0x00020000);
        m.base = reinterpret_cast<const char*>(b);
        return m;
    }
    template <typename T> __device__ __forceinline__ int off(const T* p) const { return (int)(reinterpret_cast<const char*>(p) - base); }
    template <typename T> __device__ __forceinline__ T ld1(const T* p) const {
        if constexpr (sizeof(T) == 4) {
            return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, off(p), 0, AUX));
        } else {
            typedef int v2i __attribute__((ext_vector_type(2)));
            return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(r, off(p), 0, AUX));
        }
    }
    template <typename T, int V> __device__ __forceinline__ Vec<T, V> ld(const T* p) const {
        if constexpr (V == 1) {
            Vec<T, V> o; o.v[0] = ld1<T>(p); return o;
        } else {
            static_assert(sizeof(Vec<T, V>) == 16, "16-byte lanes");
            typedef int v4i __attribute__((ext_vector_type(4)));
            return __builtin_bit_cast(Vec<T, V>, __builtin_amdgcn_raw_buffer_load_b128(r, off(p), 0, AUX));
        }
    }
    template <typename T, int V> __device__ __forceinline__ void st(T* p, const Vec<T, V>& v) const {
        if constexpr (V == 1 && sizeof(T) == 4) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v.v[0]), r, off(p), 0, AUX);
        } else if constexpr (V == 1) {
            typedef int v2i __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v.v[0]), r, off(p), 0, AUX);
        } else {
            typedef int v4i __attribute__((ext_vector_type(4)));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), r, off(p), 0, AUX);
        }
    }
};

template <typename T, int V, bool NEXT, bool PREV, typename MP = PlainMem>
__device__ __forceinline__ void load_xn(const DG& g, const T* plane_c, const T* plane_p, const T* plane_n,
                                        const Coord& c, XN<T, V>& o, const MP& mp = MP()) {
    const long long off = (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
    const T* p = plane_c + off;
    o.c = mp.template ld<T, V>(p);
    o.col0 = c.col0;
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    o.nr = o.pr = o.nc = o.pc = o.nz = o.pz = o.nt = o.pt = zero;
    o.h_nr = o.h_pr = o.h_nz = o.h_pz = o.h_nt = o.h_pt = false;
    if (NEXT) {
        o.h_nr = (c.y + 1 < g.ny);
        if (o.h_nr) o.nr = mp.template ld<T, V>(p + g.rp);
        const T tail = (c.col0 + V < g.nx) ? mp.template ld1<T>(p + V) : T(0);
        o.nc = shift_left<T, V>(o.c, tail);
        if (g.za) {
            o.h_nz = (plane_n != nullptr);
            if (o.h_nz) o.nz = mp.template ld<T, V>(plane_n + off);
        }
        if (g.ta) {
            o.h_nt = (c.t + 1 < g.m);
            if (o.h_nt) o.nt = mp.template ld<T, V>(p + g.s_t);
        }
    }
    if (PREV) {
        o.h_pr = (c.y > 0);
        if (o.h_pr) o.pr = mp.template ld<T, V>(p - g.rp);
        const T head = (c.col0 > 0) ? mp.template ld1<T>(p - 1) : T(0);
        o.pc = shift_right<T, V>(o.c, head);
        if (g.za) {
            o.h_pz = (plane_p != nullptr);
            if (o.h_pz) o.pz = mp.template ld<T, V>(plane_p + off);
        }
        if (g.ta) {
            o.h_pt = (c.t > 0);
            if (o.h_pt) o.pt = mp.template ld<T, V>(p - g.s_t);
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
        for (int i = 0; i < V; ++i) b_c.v[i] = ((unsigned)(n.col0 + i - 1) < (unsigned)(g.nx - 1)) ? n.c.v[i] - n.pc.v[i] : T(0);   // 1 <= col <= nx - 1 (pad columns of a pitched row: 0)
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
        T* base = d + (long long)c.zl * g.s_dz + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
        for_each_channel<S>(g, [&](auto slot, int ch) { vstore<T, V>(base + (long long)ch * g.s_z, o[decltype(slot)::value]); });
        return 0.0;
    }
};

// 1/|Dx| per voxel into an array with one extra plane in front; 0 where |Dx| == 0 (the reference sets
// |Dx| := +inf there so that the site contributes nothing, pytv/tv_GPU.py:88 -- same effect), and also where
// |Dx| is so small that its reciprocal would overflow.  Storing the reciprocal makes pass 2 division-free.
// ONE rule for "the gradient is zero" in every sub-gradient kernel (two-pass NormEpi, one-pass inv_norm, marching
// D_norms): |Dx|^2 below the smallest normal number of the type
template <typename T> __device__ __forceinline__ T tiny_sumsq();
template <> __device__ __forceinline__ float tiny_sumsq<float>() { return 0x1p-126f; }
template <> __device__ __forceinline__ double tiny_sumsq<double>() { return 0x1p-1022; }
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
            n.v[i] = (s.v[i] >= tiny_sumsq<T>()) ? T(1) / r : T(0);
        }
        vstore<T, V>(norms_ext + (long long)(c.zl + 1) * g.s_z + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0, n);
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
        T* base = q + (long long)c.zl * g.s_dz + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
        Vec<T, V> v[8];
        Vec<T, V> vs = vsplat<T, V>(T(0));
        for_each_channel<S>(g, [&](auto slot, int ch) {
            constexpr int k = decltype(slot)::value;
            v[k] = vload_s<T, V, S == CENTRAL>(base + (long long)ch * g.s_z) + sigma * o[k];
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
            vstore_s<T, V, S == CENTRAL>(base + (long long)ch * g.s_z, v[k] * scale);
        });
        return acc;
    }
};

// ADMM: v = Dx + u;  z = v * max(0, 1 - thresh/|v|);  u = v - z.   tform: the first array receives t = z - u_new
// (= 2 z - v) instead of z: the next right-hand side x0 + rho D^T (z - u) then reads ONE gradient array
template <int S, typename T, int V> struct AdmmZU {
    static constexpr bool REDUCES = true;
    T* z;
    T* u;
    T thresh;
    double* partials;
    int tform = 0;
    __device__ __forceinline__ double operator()(const DG& g, const Coord& c, const Vec<T, V> (&o)[8]) const {
        const long long off = (long long)c.zl * g.s_dz + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
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
            const Vec<T, V> un = v[k] - zz;
            vstore<T, V>(z + off + (long long)ch * g.s_z, tform ? zz - un : zz);
            vstore<T, V>(u + off + (long long)ch * g.s_z, un);
        });
        return acc;
    }
};

// out = x + rho * D^T D x from the HYBRID channel slots (D^T D is the same operator for upwind, downwind and
// hybrid: sum_a w_a^2 (bwd_a - fwd_a)); the slots hold s*w*fwd / s*w*bwd with s = 1/sqrt(2).  Marching path only.
template <int S, typename T, int V> struct NormalEpi {
    static constexpr bool REDUCES = true;
    static_assert(S == HYBRID, "evaluated with the hybrid channel slots");
    T* out;
    T rho, wz, wt, sf;
    const uint8_t* mask;
    double* partials;
    __device__ __forceinline__ double operator()(const DG& g, const Coord& c, const Vec<T, V> (&o)[8], const Vec<T, V>& xc) const {
        const T inv_s = T(1.4142135623730950488);
        Vec<T, V> r = (o[2] - o[0]) + (o[3] - o[1]);
        if (g.za) r = r + wz * (o[5] - o[4]);
        if (g.ta) {
            Vec<T, V> rt = wt * (o[7] - o[6]);                     // the slots already carry one mask factor
            if (mask != nullptr || g.tf != nullptr) rt = rt * mask_factor<T, V>(g, sf, c.y, c.col0);
            r = r + rt;
        }
        Vec<T, V> ov;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            ov.v[i] = xc.v[i] + rho * (inv_s * r.v[i]);
            acc += (double)xc.v[i] * (double)ov.v[i];
        }
        vstore<T, V>(out + (long long)c.zl * g.s_z + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0, ov);
        return acc;
    }
};

// =============================================================================================
// sub-gradient of one site (radius-1 schemes):  with f = forward, b = backward difference at the site,
//   D up-channel at p-e equals (w b)(p), D down-channel at p+e equals (w f)(p), so
//   G(p) = s * sum_a [ up: d_b/n(p-e) - d_f/n(p) ] + [ down: d_b/n(p) - d_f/n(p+e) ],  d = ((w diff) mf) s as in D;
//   1/n comes precomputed from pass 1 (NormEpi), so this is multiplications only
// =============================================================================================
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> div_where(const Vec<T, V>& d, const Vec<T, V>& inv_n, bool valid) {
    Vec<T, V> r = vsplat<T, V>(T(0));
    if (valid) r = d * inv_n;                                         // inv_n == 0 where |Dx| == 0
    return r;
}

// sub-gradient of one site-vector from the radius-1 neighbourhoods of x (xs) and of 1/|Dx| (ns)
// mf_prev / mf_next (weight volume only): the time-channel factor of the voxel one frame back / ahead -- D_up(p - e_t)
// carries the factor of frame t-1 and D_down(p + e_t) that of frame t+1; nullptr: the factor does not depend on t
template <int S, typename T, int V>
__device__ __forceinline__ Vec<T, V> subgrad_site(const DG& g, const WT<T>& w, const XN<T, V>& xs, const XN<T, V>& ns,
                                                  const Vec<T, V>& mf, const Vec<T, V>* mf_prev = nullptr,
                                                  const Vec<T, V>* mf_next = nullptr) {
    static_assert(S != CENTRAL, "radius-2 scheme");
    constexpr bool UP = (S == UPWIND || S == HYBRID), DN = (S == DOWNWIND || S == HYBRID);
    const T s = (S == HYBRID) ? Consts<T>::inv_sqrt2() : T(1);
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    Vec<T, V> r = zero;
    auto axis = [&](const Vec<T, V>& nxt, const Vec<T, V>& prv, bool hn, bool hp, const Vec<T, V>& n_nxt, const Vec<T, V>& n_prv,
                    T wa, bool weighted, bool timeax) {
        Vec<T, V> f = hn ? nxt - xs.c : zero, b = hp ? xs.c - prv : zero;
        if (weighted) { f = wa * f; b = wa * b; }
        if (timeax) { f = f * mf; b = b * mf; }
        if (S == HYBRID) { f = s * f; b = s * b; }
        if (UP) r = r + (div_where<T, V>(b, n_prv, hp) - div_where<T, V>(f, ns.c, hn));
        if (DN) r = r + (div_where<T, V>(b, ns.c, hp) - div_where<T, V>(f, n_nxt, hn));
    };
    axis(xs.nr, xs.pr, xs.h_nr, xs.h_pr, ns.nr, ns.pr, T(1), false, false);
#pragma unroll
    for (int i = 0; i < V; ++i) {      // columns: validity per element
        const int col = xs.col0 + i;
        const bool hn = col < g.nx - 1, hp = col > 0;
        T fi = hn ? xs.nc.v[i] - xs.c.v[i] : T(0), bi = hp ? xs.c.v[i] - xs.pc.v[i] : T(0);
        if (S == HYBRID) { fi *= s; bi *= s; }
        T acc = T(0);
        if (UP) acc += (hp ? bi * ns.pc.v[i] : T(0)) - (hn ? fi * ns.c.v[i] : T(0));
        if (DN) acc += (hp ? bi * ns.c.v[i] : T(0)) - (hn ? fi * ns.nc.v[i] : T(0));
        r.v[i] += acc;
    }
    if (g.za) axis(xs.nz, xs.pz, xs.h_nz, xs.h_pz, ns.nz, ns.pz, w.wz, true, false);
    if (g.ta && mf_prev == nullptr) axis(xs.nt, xs.pt, xs.h_nt, xs.h_pt, ns.nt, ns.pt, w.wt, true, true);
    if (g.ta && mf_prev != nullptr) {
        Vec<T, V> f = xs.h_nt ? xs.nt - xs.c : zero, b = xs.h_pt ? xs.c - xs.pt : zero;
        f = w.wt * f; b = w.wt * b;
        Vec<T, V> f_c = f * mf, b_c = b * mf, b_p = b * (*mf_prev), f_n = f * (*mf_next);
        if (S == HYBRID) { f_c = s * f_c; b_c = s * b_c; b_p = s * b_p; f_n = s * f_n; }
        if (UP) r = r + (div_where<T, V>(b_p, ns.pt, xs.h_pt) - div_where<T, V>(f_c, ns.c, xs.h_nt));
        if (DN) r = r + (div_where<T, V>(b_c, ns.c, xs.h_pt) - div_where<T, V>(f_n, ns.nt, xs.h_nt));
    }
    if (S == HYBRID) r = s * r;
    return r;
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
template <typename T, int V> struct AxpyDT {     // out = base + beta * base2 + alpha * r
    static constexpr bool REDUCES = false;
    T* out;
    const T* base;
    T alpha;
    double* partials;
    const T* base2 = nullptr;       // tv_DT_axpy2: the primal step of Chambolle-Pock with a data-fidelity operator,
    T beta = T(0);                  // x - tau A^T p - tau D^T q in one pass (base = x, base2 = A^T p, beta = alpha = -tau)
    __device__ __forceinline__ double operator()(long long off, const Vec<T, V>& r) const {
        Vec<T, V> b = (base != nullptr) ? vload_s<T, V>(base + off) : vsplat<T, V>(T(0));
        if (base2 != nullptr) b = b + beta * vload_s<T, V>(base2 + off);
        vstore_s<T, V>(out + off, b + alpha * r);
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
        const Vec<T, V> xv = vload_s<T, V>(x + off), x0v = vload_s<T, V>(x0 + off), pv = vload_s<T, V>(p + off);
        Vec<T, V> pn, xn;
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            pn.v[i] = (pv.v[i] + sigma_a * (xv.v[i] - x0v.v[i])) * inv_1p_sigma_a;
            xn.v[i] = (xv.v[i] - tau * pn.v[i]) - tau * r.v[i];
            const double e = (double)xn.v[i] - (double)x0v.v[i];
            acc += 0.5 * e * e;
        }
        vstore_s<T, V>(p + off, pn);
        vstore_s<T, V>(x + off, xn);
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


}  // namespace tv
