// tv_site.h -- the per-site bodies of the one-site-per-thread kernels (tv_kernels.hip: k_D, k_DT, k_subgrad_vec, k_subgrad_central_vec) as
// device functions of an explicit site coordinate, parametrised by how memory is reached (tv_stencil.h: PlainMem / CohMem).  Shared by the
// ordinary kernels (one launch per operator apply) and by the persistent small-volume kernels of tv_small.hip (many iterations per launch,
// sites revisited by the same thread, neighbour values written by other blocks of the same launch).
//
// Per-voxel definitions follow SURVEY 8a-1 / 8a-2, i.e. pytv/tv_operators_CPU.py:117-154,198-218,264-284,330-358 (D) and
// :398-448,487-516,554-583,622-658 (D^T); the sub-gradient follows pytv/tv_CPU.py:91-126,176-190,239-253,302-330.
#pragma once
#include "tv_stencil.h"

namespace tv {

// forward operator at ONE site-vector (the body of k_D): gradient channels of the voxels (c.zl, c.t, c.y, c.col0 ...) handed to the epilogue
template <int S, typename T, int V, typename Epi, typename MP>
__device__ __forceinline__ double d_site(const DG& g, const WT<T>& w, const T* x, const T* xp, const T* xn, int hp, const Coord& c,
                                         const Epi& epi, const MP& mp) {
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
        load_xn<T, V, NEXT, PREV, MP>(g, pc, pp, pn, c, n, mp);
        Vec<T, V> mf = g.ta ? mask_factor<T, V>(g, w.sf, c.y, c.col0) : vsplat<T, V>(T(1));
        if (g.ta && g.wv != nullptr) mf = mf * vol_factor<T, V>(g, c.zl, c.t, c.y, c.col0);   // the channel lives at this voxel
        Vec<T, V> o[8];
        d_slots<S, T, V>(g, w, n, mf, o);
        acc = epi(g, c, o);
    }
    return acc;
}

// transposed operator at ONE site-vector as a gather (the body of k_DT): returns (D^T y) at the voxels of c; inpl receives the offset inside the plane
template <int S, typename T, int V, typename Src>
__device__ __forceinline__ Vec<T, V> dt_site(const DG& g, const WT<T>& w, const Src& src, const Coord& c, long long& inpl_out) {
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    const long long inpl = (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;   // offset inside a plane
    const long long offd = (long long)c.zl * g.s_dz + inpl;                          // channel 0 of this voxel
    const int gz = g.z0 + c.zl;
    Vec<T, V> r = zero, rt = zero;

    // ---- one (channel, mode) at a time; MODE as in adj_axis -------------------------------
    auto rows = [&](auto mode, int ch) {
        constexpr int M = decltype(mode)::value;
        const long long o = offd + (long long)ch * g.s_z;
        const Vec<T, V> lo = (c.y >= 1 && M != 1) ? src.ld(o - g.rp) : zero;
        const Vec<T, V> ce = (M != 2) ? src.ld(o) : zero;
        const Vec<T, V> hi = (c.y + 1 < g.ny && M != 0) ? src.ld(o + g.rp) : zero;
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
        Vec<T, V> lo = (c.t >= 1 && M != 1) ? src.ld(o - g.s_t) : zero;
        Vec<T, V> ce = (M != 2) ? src.ld(o) : zero;
        Vec<T, V> hi = (c.t + 1 < g.m && M != 0) ? src.ld(o + g.s_t) : zero;
        if (g.wv != nullptr) {      // weight volume: exact adjoint, every sample carries its own voxel's factor
            lo = lo * vol_factor<T, V>(g, c.zl, c.t - 1, c.y, c.col0);
            ce = ce * vol_factor<T, V>(g, c.zl, c.t, c.y, c.col0);
            hi = hi * vol_factor<T, V>(g, c.zl, c.t + 1, c.y, c.col0);
        }
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
    zero_pad_cols<T, V>(g, c.col0, r);
    inpl_out = inpl;
    return r;
}

// sub-gradient at ONE site-vector from x and 1/|Dx| (pass 2 of the two-pass form, radius-1 schemes: the body of k_subgrad_vec)
template <int S, typename T, int V, typename MPX, typename MPN>
__device__ __forceinline__ Vec<T, V> sg_site(const DG& g, const WT<T>& w, const T* x, const T* xp, const T* xn, const T* norms_ext, const Coord& c,
                                             const MPX& mpx, const MPN& mpn) {
    const T* pc = zplane<T>(g, x, xp, xn, 2, c.zl);
    const T* pp = g.za ? zplane<T>(g, x, xp, xn, 2, c.zl - 1) : nullptr;
    const T* pn = g.za ? zplane<T>(g, x, xp, xn, 2, c.zl + 1) : nullptr;
    XN<T, V> xs, ns;
    load_xn<T, V, true, true, MPX>(g, pc, pp, pn, c, xs, mpx);
    const T* nc = norms_ext + (long long)(c.zl + 1) * g.s_z;
    load_xn<T, V, true, true, MPN>(g, nc, pp ? nc - g.s_z : nullptr, pn ? nc + g.s_z : nullptr, c, ns, mpn);
    const Vec<T, V> mf2 = g.ta ? mask_factor<T, V>(g, w.sf, c.y, c.col0) : vsplat<T, V>(T(1));
    Vec<T, V> r;
    if (g.ta && g.wv != nullptr) {
        const Vec<T, V> mf = mf2 * vol_factor<T, V>(g, c.zl, c.t, c.y, c.col0);
        const Vec<T, V> mfp = mf2 * vol_factor<T, V>(g, c.zl, c.t - 1, c.y, c.col0), mfn = mf2 * vol_factor<T, V>(g, c.zl, c.t + 1, c.y, c.col0);
        r = subgrad_site<S, T, V>(g, w, xs, ns, mf, &mfp, &mfn);
    } else {
        r = subgrad_site<S, T, V>(g, w, xs, ns, mf2);
    }
    zero_pad_cols<T, V>(g, c.col0, r);
    return r;
}

// central-scheme sub-gradient at ONE site-vector (the body of k_subgrad_central_vec)
template <typename T, int V, typename MPX, typename MPN>
__device__ __forceinline__ Vec<T, V> sg_site_central(const DG& g, const WT<T>& w, const T* x, const T* xp, const T* xn, const T* norms_ext,
                                                     const Coord& c, const MPX& mpx, const MPN& mpn) {
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    const long long inpl = (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
    const T* pc = zplane<T>(g, x, xp, xn, 2, c.zl) + inpl;
    const T* nc = norms_ext + (long long)(c.zl + 1) * g.s_z + inpl;
    const Vec<T, V> xc = mpx.template ld<T, V>(pc);
    const Vec<T, V> mf2 = g.ta ? mask_factor<T, V>(g, w.sf, c.y, c.col0) : vsplat<T, V>(T(1));
    // time-channel factor of the voxel one frame back / this one / one frame ahead (they differ only with a weight volume)
    Vec<T, V> mf = mf2, mfa = mf2, mfb = mf2;
    if (g.ta && g.wv != nullptr) {
        mf = mf2 * vol_factor<T, V>(g, c.zl, c.t, c.y, c.col0);
        mfa = mf2 * vol_factor<T, V>(g, c.zl, c.t - 1, c.y, c.col0);
        mfb = mf2 * vol_factor<T, V>(g, c.zl, c.t + 1, c.y, c.col0);
    }
    const T h = T(0.5);
    Vec<T, V> r = zero;
    // term of one axis from the vectors two steps away (x) and one step away (norms); wa < 0: unweighted
    auto cen = [&](int pos, int n, const Vec<T, V>& xm2, const Vec<T, V>& xp2, const Vec<T, V>& nm1, const Vec<T, V>& np1, T wa,
                   bool weighted, bool timeax) {
        if (pos - 1 > 0 && pos - 1 < n - 1) {
            Vec<T, V> d = xc - xm2;
            if (weighted) d = wa * d;
            if (timeax) d = d * mfa;
            d = h * d;
#pragma unroll
            for (int i = 0; i < V; ++i) r.v[i] += d.v[i] * nm1.v[i];
        }
        if (pos + 1 > 0 && pos + 1 < n - 1) {
            Vec<T, V> d = xp2 - xc;
            if (weighted) d = wa * d;
            if (timeax) d = d * mfb;
            d = h * d;
#pragma unroll
            for (int i = 0; i < V; ++i) r.v[i] -= d.v[i] * np1.v[i];
        }
    };
    auto fwd = [&](int pos, int n, const Vec<T, V>& xm1, const Vec<T, V>& xp1, const Vec<T, V>& nm1, const Vec<T, V>& n0, T wa,
                   bool timeax) {
        if (pos >= 1) {             // g(p-e), d(p-e) = 1/2 w (x(p) - x(p-e))
            Vec<T, V> d = wa * (xc - xm1);
            if (timeax) d = d * mfa;
            d = h * d;
#pragma unroll
            for (int i = 0; i < V; ++i) r.v[i] += d.v[i] * nm1.v[i];
        }
        if (pos <= n - 2) {         // - g(p), d(p) = 1/2 w (x(p+e) - x(p))
            Vec<T, V> d = wa * (xp1 - xc);
            if (timeax) d = d * mf;
            d = h * d;
#pragma unroll
            for (int i = 0; i < V; ++i) r.v[i] -= d.v[i] * n0.v[i];
        }
    };
    const long long nx = g.rp;      // row pitch
    cen(c.y, g.ny, (c.y >= 2) ? mpx.template ld<T, V>(pc - 2 * nx) : zero, (c.y + 2 < g.ny) ? mpx.template ld<T, V>(pc + 2 * nx) : zero,
        (c.y >= 1) ? mpn.template ld<T, V>(nc - nx) : zero, (c.y + 1 < g.ny) ? mpn.template ld<T, V>(nc + nx) : zero, T(1), false, false);
    {   // columns, per element
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const int col = c.col0 + i;
            if (col - 1 > 0 && col - 1 < g.nx - 1) r.v[i] += (h * (xc.v[i] - mpx.template ld1<T>(pc + i - 2))) * mpn.template ld1<T>(nc + i - 1);
            if (col + 1 > 0 && col + 1 < g.nx - 1) r.v[i] -= (h * (mpx.template ld1<T>(pc + i + 2) - xc.v[i])) * mpn.template ld1<T>(nc + i + 1);
        }
    }
    if (g.za) {
        const int gz = g.z0 + c.zl;
        const T* nm1 = (gz >= 1) ? nc - g.s_z : nullptr;
        const T* np1 = (gz + 1 < g.nzg) ? nc + g.s_z : nullptr;
        if (g.z_two) {
            const T* pm = zplane<T>(g, x, xp, xn, 2, c.zl - 1);
            const T* pq = zplane<T>(g, x, xp, xn, 2, c.zl + 1);
            fwd(gz, g.nzg, pm ? mpx.template ld<T, V>(pm + inpl) : zero, pq ? mpx.template ld<T, V>(pq + inpl) : zero, nm1 ? mpn.template ld<T, V>(nm1) : zero,
                mpn.template ld<T, V>(nc), w.wz, false);
        } else {
            const T* pm = zplane<T>(g, x, xp, xn, 2, c.zl - 2);
            const T* pq = zplane<T>(g, x, xp, xn, 2, c.zl + 2);
            cen(gz, g.nzg, pm ? mpx.template ld<T, V>(pm + inpl) : zero, pq ? mpx.template ld<T, V>(pq + inpl) : zero, nm1 ? mpn.template ld<T, V>(nm1) : zero,
                np1 ? mpn.template ld<T, V>(np1) : zero, w.wz, true, false);
        }
    }
    if (g.ta) {
        if (g.t_two)
            fwd(c.t, g.m, (c.t >= 1) ? mpx.template ld<T, V>(pc - g.s_t) : zero, (c.t + 1 < g.m) ? mpx.template ld<T, V>(pc + g.s_t) : zero,
                (c.t >= 1) ? mpn.template ld<T, V>(nc - g.s_t) : zero, mpn.template ld<T, V>(nc), w.wt, true);
        else
            cen(c.t, g.m, (c.t >= 2) ? mpx.template ld<T, V>(pc - 2 * g.s_t) : zero, (c.t + 2 < g.m) ? mpx.template ld<T, V>(pc + 2 * g.s_t) : zero,
                (c.t >= 1) ? mpn.template ld<T, V>(nc - g.s_t) : zero, (c.t + 1 < g.m) ? mpn.template ld<T, V>(nc + g.s_t) : zero, w.wt, true, true);
    }
    Vec<T, V> hr = h * r;
    zero_pad_cols<T, V>(g, c.col0, hr);
    return hr;
}

}  // namespace tv
