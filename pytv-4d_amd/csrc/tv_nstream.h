// tv_nstream.h -- out = (I + rho D^T D) x from x alone as a STREAMING kernel (tv_normal_op: the inner operator of the ADMM
// x-update's conjugate-gradient loop), radius-1 schemes (upwind / downwind / hybrid share D^T D = sum_a w_a^2 (bwd_a -
// fwd_a) as long as the time weight does not vary along t), fp32, 16-byte lanes, any number of frames (windows of 8).
//
// Same tile as k_D_stream (tv_dstream.h): a wave covers 4 rows x 16 lanes, row neighbours are 16-lane shuffles, column
// neighbours one-lane DPP shifts, halo rows / edge elements one load each for the lanes that need them; no LDS tile; x is
// read once, every load is issued one plane ahead of its use.  State: planes z-1, z of the M frames, the unfinished result
// of plane z-1 and the halo rows / edge elements of plane z in registers.  Since late round 5 every access is a raw buffer
// access through a descriptor (no branch around a memory operation), planes are loaded in place and the epilogue's operands
// are requested a frame ahead: how and why is written where it happens (ns_epi_load_buf, the frame loop of k_normal_stream)
// and in EXPERIMENTS.md 5.7.
//
// Modes (NormalArgs): b == nullptr:  out = A x,  dots = { <x, out>, <x, x> }
//                     b != nullptr:  out = b - A x (and out2 = out when given: r and the first search direction of CG),
//                                    dots = { <out, out>, <x, x> }
#pragma once
#include <type_traits>
#include "tv_device.h"
#include "tv_stencil.h"
#include "tv_fused.h"
#include "tv_dstream.h"

namespace tv {

template <typename T> struct NormalArgsT {
    const T* x;
    const T* xp;           // TWO planes z0-2, z0-1 (or nullptr)
    const T* xn;           // TWO planes z0+nz, z0+nz+1 (or nullptr)
    const T* b;            // or nullptr
    T* out;
    T* out2;               // or nullptr
    T rho;
    double* part0;         // per-block partials of the first / second dot product
    double* part1;
    // Chebyshev step (round 3, tv_cheb_step; needs b): out = [add +] x + alpha (b - A x) + beta (x - y);  y (nullptr: 0), add, ref or nullptr
    // part0 <- |b - A x|^2, part1 <- |out - ref|^2 (ref given) or |x|^2
    const T* y;
    const T* add;
    const T* ref;
    T alpha, beta;
    int cheb;
    T yscale;              // y == nullptr: y = yscale * b (0: no y) -- the second Chebyshev step, whose e_1 = a_0 b need not exist in memory
};
using NormalArgs = NormalArgsT<float>;

// epilogue of one site-vector: xm = x, ax = A x there.  CHEB is a TEMPLATE parameter: as a run-time branch in the same kernel the
// Chebyshev form cost the CG instantiations 35 % (tv_normal_op 0.95 -> 1.30 ms at 64x8x1024x1024: the extra pointers and the
// branch sit in the hot loop of a kernel that is at its register limit).
#ifndef TV_NSTREAM_NT
#define TV_NSTREAM_NT 1          // the streamed-once arrays of the epilogue (out, b, y, add, ref) non-temporal: tv_normal_op -2 - 5 %, Chebyshev ADMM -3 - 4 % (0: plain)
#endif
template <typename T, int V> __device__ __forceinline__ void NSTU(T* ubase, unsigned voff, const Vec<T, V>& v) {
#if TV_NSTREAM_NT
    stu_s_t<T, V>(ubase, voff, v);
#else
    stu_t<T, V>(ubase, voff, v);
#endif
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> NLDU(const T* ubase, unsigned voff) {
#if TV_NSTREAM_NT
    return ldu_s_t<T, V>(ubase, voff);
#else
    return ldu_t<T, V>(ubase, voff);
#endif
}
// The streamed operands of one site-vector's epilogue.  Round 5 (late): they are REQUESTED ONE FRAME AHEAD of the epilogue that consumes
// them (ns_epi_load_buf at the end of frame t for frame t + 1).  Loaded where they were consumed they were the youngest loads in flight, so
// every frame ended in s_waitcnt vmcnt(0): a full memory round trip per frame with nothing else to do, which also drained the plane-ahead
// requests of the stencil -- the kernel ran at the latency of memory, not its bandwidth (Chebyshev step 0.54 of peak at 8 waves per CU).
template <typename T, int V> struct NsEpiIn { Vec<T, V> b, y, add, ref; };
// Every stream goes through ONE descriptor per plane (tv_fused.h, "raw buffer access": base = frame 0 of the plane, num_records = the plane's
// bytes, 0 for an absent stream or plane) + the frame's byte offset as the instruction's scalar offset (`soff`): the four loads are
// unconditional, the frame loop is straight-line code, and a frame costs no descriptor arithmetic.
// po: element offset of the plane, valid: the plane's epilogue will run
template <typename T, int V, int CHEB>
__device__ __forceinline__ void ns_epi_load_buf(const NormalArgsT<T>& a, long long po, bool valid, int pbytes, unsigned boff, int soff, NsEpiIn<T, V>& in) {
    constexpr int NT = TV_NSTREAM_NT ? BUF_NT : 0;
    // CHEB == 2 (round 6): the first step of a Chebyshev solve as its own instantiation -- b == x, no y / add / ref: NO operand loads at all.
    // In the general form the four loads of that step go to absent streams and cost no memory traffic, but they sit in the in-order queue
    // BEHIND the plane-ahead loads of the frame before, and the epilogue's wait for them (vmcnt(6 .. 3) per frame) is therefore a wait for
    // stencil loads issued two frames ago: ~ 2 KB per wave in flight, 4 TB/s.  Without them a frame waits only for what was requested a
    // whole plane step ago.
    if constexpr (CHEB == 1) {
        // b == x (the first step of a Chebyshev solve: e_2 from r alone, x = b = r): the value is in registers already -- one stream less
        in.b = buf_ld<T, V, NT>(buf_rsrc<T>(a.b + po, valid && a.b != a.x, pbytes), boff, soff);
        in.y = buf_ld<T, V, NT>(buf_rsrc<T>(a.y + po, valid && a.y != nullptr, pbytes), boff, soff);
        in.add = buf_ld<T, V, NT>(buf_rsrc<T>(a.add + po, valid && a.add != nullptr, pbytes), boff, soff);
        in.ref = buf_ld<T, V, NT>(buf_rsrc<T>(a.ref + po, valid && a.ref != nullptr, pbytes), boff, soff);
    }
    // (the CG forms load b where they use it, inside the branch that tells them apart: tv_normal_op without b has no operand at all, and a
    // load from an absent stream still takes its slot in the queue)
}
// BUF: the stores go through descriptors too (plane offset po, plane bytes, frame offset soff; voff = the lane's offset or BUF_OOB: lanes
// outside the frame drop their store); otherwise fo = po + the frame's element offset and plain global accesses
template <typename T, int V, int CHEB, bool BUF = false>
__device__ __forceinline__ void ns_epilogue(const NormalArgsT<T>& a, long long fo, unsigned voff, const Vec<T, V>& xm, const Vec<T, V>& ax,
                                            const NsEpiIn<T, V>& in, double& acc0, double& acc1, int pbytes = 0, int soff = 0) {
    constexpr int NT = TV_NSTREAM_NT ? BUF_NT : 0;
    Vec<T, V> o;
    if constexpr (CHEB != 0) {
        const Vec<T, V> bv = (CHEB == 2 || a.b == a.x) ? xm : in.b;
        // no y: y = yscale * b (0 after e_0 = 0).  An absent stream reads as 0 and yscale is 0 next to a real y (tv_cheb_step checks), so one
        // fma covers both cases without a select; `add` likewise needs none
        Vec<T, V> res;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const T yv = in.y.v[i] + a.yscale * bv.v[i];
            res.v[i] = bv.v[i] - ax.v[i];
            o.v[i] = in.add.v[i] + ((xm.v[i] + a.alpha * res.v[i]) + a.beta * (xm.v[i] - yv));
        }
        if (a.part0 != nullptr) {          // the dot products on request only (tv_cheb_step, dots == NULL: ~ 10 % of the frame's instructions)
            const Vec<T, V> rv = (a.ref != nullptr) ? in.ref : xm;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                acc0 += (double)res.v[i] * (double)res.v[i];
                const double e = (a.ref != nullptr) ? (double)o.v[i] - (double)rv.v[i] : (double)xm.v[i];
                acc1 += e * e;
            }
        }
    } else if (a.b == nullptr) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            o.v[i] = ax.v[i];
            acc0 += (double)xm.v[i] * (double)o.v[i];
            acc1 += (double)xm.v[i] * (double)xm.v[i];
        }
    } else {
        Vec<T, V> bv;
        if constexpr (BUF) bv = buf_ld<T, V, NT>(buf_rsrc<T>(a.b + fo, true, pbytes), voff, soff);
        else bv = NLDU<T, V>(a.b + fo, voff);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            o.v[i] = bv.v[i] - ax.v[i];
            acc0 += (double)o.v[i] * (double)o.v[i];
            acc1 += (double)xm.v[i] * (double)xm.v[i];
        }
        if (a.out2 != nullptr) {
            if constexpr (BUF) buf_st<T, V, NT>(buf_rsrc<T>(a.out2 + fo, true, pbytes), voff, o, soff);
            else NSTU<T, V>(a.out2 + fo, voff, o);
        }
    }
    if constexpr (BUF) buf_st<T, V, NT>(buf_rsrc<T>(a.out + fo, true, pbytes), voff, o, soff);
    else NSTU<T, V>(a.out + fo, voff, o);
}

#ifndef TV_NS_EXIT_FROM
#define TV_NS_EXIT_FROM(M) ((M) / 2)
#endif
constexpr int NS_TWN = TV_TWN;

// The shared z term is added to one plane at one step and subtracted from its neighbour at the next, possibly by a different
// block (chunk seam) or another rank (slab seam): these three operations must round the same way wherever they are compiled
// (slab == unsharded bit for bit, tests/test_gpu_parity.py).  Left to the compiler they do not: it fuses m * (a - b) + r into an
// fma in some specialisations of the plane loop (first step of a chunk, trailing step) and not in others, and the __fmul_rn /
// __fadd_rn intrinsics do not prevent that (plain * and + from the force-included HIP headers, contraction allowed).  So the
// three operations are single VALU instructions the compiler cannot look into.
__device__ __forceinline__ float ns_vmul(float a, float b) { float r; asm("v_mul_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ns_vadd(float a, float b) { float r; asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float ns_vsub(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double ns_vmul(double a, double b) { double r; asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double ns_vadd(double a, double b) { double r; asm("v_add_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double ns_vsub(double a, double b) { double r; asm("v_add_f64 %0, %1, -%2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <typename T, int V> __device__ __forceinline__ Vec<T, V> ns_mul(T m, const Vec<T, V>& a, const Vec<T, V>& b) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = ns_vmul(m, ns_vsub(a.v[i], b.v[i]));
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> ns_add(const Vec<T, V>& a, const Vec<T, V>& b) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = ns_vadd(a.v[i], b.v[i]);
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> ns_sub(const Vec<T, V>& a, const Vec<T, V>& b) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = ns_vsub(a.v[i], b.v[i]);
    return r;
}

// RAGGED (windows only): the frame count of the volume is not a multiple of 8, the last window is short (see the frame loop)
template <int M, bool TWIN, typename T = float, int CHEB = 0, bool RAGGED = false>
__global__ __launch_bounds__(ST_THREADS, TV_WAVES ? TV_WAVES : (M >= 6 ? 2 : 3)) void k_normal_stream(DG g, WT<T> w, NormalArgsT<T> a, int zchunk, int nchunks) {
    static_assert(TWIN || !RAGGED, "only windows can be ragged");
    constexpr int V = 16 / (int)sizeof(T);          // columns per 16-byte lane: 4 floats / 2 doubles (round 3)
    using VT = Vec<T, V>;
    __shared__ double sm[16];
    const int lane = (int)threadIdx.x, wave = (int)threadIdx.y;
    const int row = lane >> 4, lx = lane & 15;
    const int nxv = (g.nx + V - 1) / V;
    const int tiles_x = (nxv + ST_BCV - 1) / ST_BCV, tiles_y = (g.ny + ST_BR - 1) / ST_BR;
    const int Mg = TWIN ? g.m : M;
    const int nwin = TWIN ? (Mg + NS_TWN - 1) / NS_TWN : 1;
    const long long ntiles = (long long)tiles_x * tiles_y, total = ntiles * nchunks * nwin, per_xcd = (total + 7) / 8;
    const long long lid = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;       // XCD-aware order (tv_dstream.h)
    double acc0 = 0.0, acc1 = 0.0;
    if (lid < total) {
        // (64-bit divisions run on the vector unit: say that the results are wave-uniform, or descriptors and scalar offsets derived from them
        // are wrapped in a loop over their values)
        const int win = __builtin_amdgcn_readfirstlane((int)(lid / (ntiles * nchunks)));
        const int chunk = __builtin_amdgcn_readfirstlane((int)((lid / ntiles) % nchunks)), tile = __builtin_amdgcn_readfirstlane((int)(lid % ntiles));
        const int t0 = TWIN ? win * NS_TWN : 0;
        const int bx = tile % tiles_x, by = tile / tiles_x;
        const int col0 = (bx * ST_BCV + (wave % ST_NWX) * 16 + lx) * V, y = (by * ST_NWY + wave / ST_NWX) * 4 + row;
        const bool ok = (col0 < g.nx) && (y < g.ny);
        const unsigned voff = ok ? (unsigned)(((long long)y * g.rp + col0) * (long long)sizeof(T)) : 0u;
        const unsigned row_bytes = (unsigned)g.rp * (unsigned)sizeof(T);
        const int zs = chunk * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        const VT zero = vsplat<T, V>(T(0));
        VT mf2 = vsplat<T, V>(T(1));
        if (g.ta) {
            const VT mf = mask_factor<T, V>(g, w.sf, ok ? y : 0, ok ? col0 : 0);
            mf2 = (w.wt * w.wt) * (mf * mf);
        }
        const T wz2 = g.za ? w.wz * w.wz : T(0);
        // existence of the in-plane neighbours as multipliers (no branches around vectors in the frame loop)
        const T m_pr = (ok && y > 0) ? T(1) : T(0), m_nr = (ok && y + 1 < g.ny) ? T(1) : T(0);
        const T m_c0 = (ok && col0 > 0) ? T(1) : T(0), m_c3 = (ok && col0 + V < g.nx) ? T(1) : T(0);
        // ragged PITCHED rows (round 4): the last lane of a row holds pad columns; the difference across the frame's last column (and
        // those between pads) must not count.  me[i] = 1 iff columns col0 + i - 1 and col0 + i both exist (i >= 1)
        T me[V];
#pragma unroll
        for (int i = 0; i < V; ++i) me[i] = (ok && col0 + i < g.nx) ? T(1) : T(0);
        const bool want_up = (row == 0) && ok && (y > 0), want_dn = (row == 3) && ok && (y + 1 < g.ny);
        const unsigned hoff = want_up ? voff - row_bytes : voff + row_bytes;
        // a tile row needs BOTH halo rows when the wave tile is a single row high at the frame border: rows 0 and 3 differ,
        // so one predicated load per lane is enough (row 0 reads y-1, row 3 reads y+1)
        const bool want_le = (lx == 0) && ok && (col0 > 0), want_re = (lx == 15) && ok && (col0 + V < g.nx);
        const unsigned eoff = want_le ? voff - (unsigned)sizeof(T) : voff + 16u;
        const int NV = ((TWIN ? Mg : g.m) - t0 < M) ? (TWIN ? Mg : g.m) - t0 : M;          // frames of this window (non-TWIN: M, but the compiler must not know -- see the frame loop)
        auto fvalid = [&](int t) { return t < NV; };
        auto foff = [&](int t) { return (long long)(t0 + t) * g.s_t; };
        // State: plane z (C), plane z-1 (P), the result of plane z-1 still waiting for its forward z term (R), the halo rows /
        // border elements of plane z (H, E).  The z term is SHARED: dz(z) = wz^2 (x(z) - x(z-1)) is the backward term of plane z
        // and minus the forward term of plane z-1, so step z finishes and stores plane z-1 and x(z+1) is never needed inside
        // the step.  That lets the centre vectors, halo rows and border elements of plane z+1 all be requested in the same
        // frame of step z -- the moment the owners of those lines request them (round 2: requested one step apart they cost
        // a second trip to memory, 17 GB read for an 8.6 GB image).
        // Registers: XA / XB hold planes z and z - 1 in turn.  The z loop is unrolled by two steps with the roles swapped: plane z + 1 is loaded
        // into the slot of plane z - 1 once its epilogue is through.  (With one step per iteration P[t] <- C[t] <- load is a rotation through
        // three registers that the compiler closes with copies at the loop's end -- copies of values still in flight, i.e. a wait for the
        // youngest loads once per step.)
        VT XA[M], XB[M], R[M], H[M];
        T E[M];
        auto plane = [&](int zl) { return g.za ? zplane<T>(g, a.x, a.xp, a.xn, 2, zl) : ((zl >= 0 && zl < g.nz) ? a.x + (long long)zl * g.s_z : nullptr); };
        // every access: plane descriptor + the frame's scalar offset + per-lane offset (tv_fused.h, "raw buffer access"); lanes / planes that do
        // not take part are out of range for the hardware, not branched around (frames t >= NV are never reached)
        const int fbytes = (int)(g.s_t * (long long)sizeof(T)), pbytes = (int)(g.s_z * (long long)sizeof(T));
        const unsigned boff = ok ? voff : BUF_OOB;
        const unsigned bhoff = (want_up || want_dn) ? hoff : BUF_OOB, beoff = (want_le || want_re) ? eoff : BUF_OOB;
        auto pdesc = [&](const T* pl, bool valid) { return buf_rsrc<T>(pl, pl != nullptr && valid, pbytes); };
        auto soff = [&](int t) { return (t0 + t) * fbytes; };
        // TWIN (M > 8: windows of 8 frames): the frames just outside the window, x(z, t0 - 1) and x(z, t0 + M), are needed for the time
        // differences of the window's first / last frame.  Round 5: they are REQUESTED A PLANE AHEAD like everything else (Wp / Wn);
        // until then each was a load issued where it was consumed -- two exposed memory round trips per plane step.
        // (Ahead of the other prologue loads: the counter is in-order and the loop's first wait for Wp must not drain them.)
        const bool w_prev = TWIN && g.ta && t0 > 0, w_next = TWIN && g.ta && t0 + M < Mg;
        VT Wp = zero, Wn = zero;
        if (TWIN) {
            const T* pc = plane(zs);
            Wp = buf_ld<T, V>(pdesc(pc, w_prev), boff, soff(-1));
            Wn = buf_ld<T, V>(pdesc(pc, w_next), boff, soff(M));
        }
        {
            const T* pp = g.za ? plane(zs - 1) : nullptr;
            const T* pc = plane(zs);
            const Rsrc rp = pdesc(pp, true), rc = pdesc(pc, true);
#pragma unroll
            for (int t = 0; t < M; ++t) {
                const bool fv = fvalid(t);          // (the prologue is not part of the frame loop: frames beyond the window load nothing)
                XB[t] = buf_ld<T, V>(rp, fv ? boff : BUF_OOB, soff(t));
                XA[t] = buf_ld<T, V>(rc, fv ? boff : BUF_OOB, soff(t));
                R[t] = zero;
                H[t] = buf_ld<T, V>(rc, fv ? bhoff : BUF_OOB, soff(t));
                E[t] = buf_ld1<T>(rc, fv ? beoff : BUF_OOB, soff(t));
            }
        }
        NsEpiIn<T, V> ein{zero, zero, zero, zero};         // operands of the NEXT epilogue, requested a frame ahead (the first one belongs to step zs + 1, frame 0)
        // one plane step: Cc = plane z, Pp = plane z - 1 (finished and stored here, then overwritten by plane z + 1).  FIRST: step zs, no epilogue
        auto step = [&](auto first_tag, int z, VT (&Cc)[M], VT (&Pp)[M]) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value;
            st_sync_plane();
            const bool in_chunk = (z < ze), next_in = (z + 1 < ze);
            const int gz = g.z0 + z;
            const T mz = (g.za && gz > 0 && gz < g.nzg) ? wz2 : T(0);
            // plane requested now (consumed at step z + 1); the step behind the chunk needs the centre vectors only
            const T* pn = (next_in || (g.za && z + 1 == ze)) ? plane(z + 1) : nullptr;
            // descriptors of plane z + 1: the centre vectors, and (the plane belongs to the chunk) its halo rows / border elements / seam frames
            const Rsrc rn_c = pdesc(pn, true), rn_h = pdesc(pn, next_in), rn_wp = pdesc(pn, next_in && w_prev), rn_wn = pdesc(pn, next_in && w_next);
            VT cold = zero;
            if (w_prev && in_chunk) cold = Wp;
#pragma unroll
            for (int t = 0; t < M; ++t) {
                // A step leaves its frame loop early in the short last window of a volume whose frame count is not a multiple of 8 -- and, as
                // far as the compiler can tell, anywhere (NV is computed from a kernel argument in every instantiation).  Deliberate: the early
                // exits are edges into ONE block behind the step, so every register a frame loads into (plane z + 1, halo row, border element,
                // epilogue operands) is a phi of "loaded" and "kept" there and has to be the register it was: loads IN PLACE.  Written without
                // the exits (straight-line frames, or exits that do anything on their way) the same kernel takes 256 VGPRs + 300 - 900 B of
                // scratch at M = 8 and runs at half the speed (tv_cheb_step 3.4 ms against 1.7 at 64x8x1024x1024).
                // The price: s_waitcnt counts along the shortest static path from a load to its consumer, and the path "leave after frame 0"
                // is short -- with an exit in front of every frame, frame 0 of every step waits for (nearly) everything in flight.  So the exits
                // start at frame M / 2 (frames 0 .. M/2 - 1 are straight-line; still no spills), where the shortest path is half a step long:
                // one drain per TWO steps is left (the loop header's), from one per step.  A volume whose last window may be shorter than
                // that is the RAGGED instantiation: exits from frame 1 on.
                if (t >= (RAGGED ? 1 : TV_NS_EXIT_FROM(M)) && t >= NV) break;
                st_sync_frame();
                const int tg = t0 + t;
                const VT c = Cc[t], h = H[t], xm = Pp[t];
                const VT dz = ns_mul<T, V>(mz, c, xm);
                const VT rfin = ns_sub<T, V>(R[t], dz);            // plane z-1 is complete
                if (in_chunk) {
                    // ---- - Laplacian-like sum: (c - prev) - (next - c) per axis, missing neighbours drop their term ------
                    const VT sdn = shfl_down16_t<T, V>(c), sup = shfl_up16_t<T, V>(c);
                    const VT nr = (row == 3) ? h : sdn, pr = (row == 0) ? h : sup;
                    VT r = m_pr * (c - pr) - m_nr * (nr - c);
                    {
                        // the cross-lane moves are executed by EVERY lane (a DPP read from a lane that a branch has switched
                        // off returns 0), the select comes afterwards
                        const T from_l = dpp_from_left(c.v[V - 1]), from_r = dpp_from_right(c.v[0]);
                        const T left = (lx == 0) ? E[t] : from_l;
                        const T right = (lx == 15) ? E[t] : from_r;
                        // interior elements of the vector have both column neighbours inside the frame unless the row is ragged (me[])
                        T e[V];         // e[i] = c[i] - c[i-1] (e[0]: against the left neighbour), one more for the right neighbour
#pragma unroll
                        for (int i = 1; i < V; ++i) e[i] = me[i] * (c.v[i] - c.v[i - 1]);
                        r.v[0] += m_c0 * (c.v[0] - left) - e[1];
#pragma unroll
                        for (int i = 1; i + 1 < V; ++i) r.v[i] += e[i] - e[i + 1];
                        r.v[V - 1] += e[V - 1] - m_c3 * (right - c.v[V - 1]);
                    }
                    if (g.ta) {
                        VT tt = zero;
                        if (tg > 0) tt = tt + (c - cold);
                        if (t + 1 < M) { if (tg + 1 < Mg) tt = tt - (Cc[(t + 1 < M) ? t + 1 : t] - c); }
                        else if (TWIN && tg + 1 < Mg) tt = tt - (Wn - c);
                        r = r + mf2 * tt;
                    }
                    R[t] = ns_add<T, V>(r, dz);
                }
                cold = c;
                // ---- epilogue of plane z-1 --------------------------------------------------------------------------------------------
                if constexpr (!FIRST) {
                    VT ax;
#pragma unroll
                    for (int i = 0; i < V; ++i) ax.v[i] = xm.v[i] + a.rho * rfin.v[i];
                    ns_epilogue<T, V, CHEB, true>(a, (long long)(z - 1) * g.s_z, boff, xm, ax, ein, acc0, acc1, pbytes, soff(t));
                }
                // ---- the operands of the next epilogue: frame t + 1 of this step, or frame 0 of the next one.  FIRST in the queue: the counter is
                // in-order and these are needed a frame from now, the plane loads below a step from now ------------------------------------
                {
                    const bool more = (t + 1 < M) && fvalid((t + 1 < M) ? t + 1 : t);
                    if constexpr (CHEB != 0)
                        ns_epi_load_buf<T, V, CHEB>(a, (long long)(more ? z - 1 : z) * g.s_z, more ? !FIRST : (z < ze), pbytes, boff,
                                                    __builtin_amdgcn_readfirstlane(soff(more ? t + 1 : 0)), ein);      // (uniform; said so, or the compiler loops over its values)
                }
                // ---- plane z + 1 into the slot of plane z - 1, its halo row / border element, the seam frames ---------------------------
                {
                    Pp[t] = buf_ld<T, V>(rn_c, boff, soff(t));
                    H[t] = buf_ld<T, V>(rn_h, bhoff, soff(t));
                    E[t] = buf_ld1<T>(rn_h, beoff, soff(t));
                    if (TWIN && t == 0) Wp = buf_ld<T, V>(rn_wp, boff, soff(-1));          // consumed at the start of step z + 1
                    if (TWIN && t == M - 1) Wn = buf_ld<T, V>(rn_wn, boff, soff(M));       // ... at its end
                }
            }
        };
        step(std::true_type{}, zs, XA, XB);
        for (int z = zs + 1; z <= ze; z += 2) {              // step ze only finishes plane ze - 1
            step(std::false_type{}, z, XB, XA);
            if (z + 1 > ze) break;
            step(std::false_type{}, z + 1, XA, XB);
        }
    }
    if (a.part0 == nullptr) return;          // (Chebyshev step without dot products)
    acc0 = block_sum(acc0, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part0[blockIdx.x] = acc0;
    acc1 = block_sum(acc1, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part1[blockIdx.x] = acc1;
}

// =============================================================================================
// central scheme: D^T D = 1/4 sum_a w_a^2 [ v(p-e) (x(p) - x(p-2e)) - v(p+e) (x(p+2e) - x(p)) ],  v(q) = 1 iff q is an
// interior point of the axis -- a radius-1 stencil on the STRIDE-2 sub-lattice of every axis: even and odd planes (rows,
// columns, frames) decouple.  So the block marches the even planes of its z-chunk, then the odd ones, with planes z-2, z,
// z+2 in registers (3 M vectors, as for the other schemes); rows y+-2 are 32-lane shuffles inside the 4-row wave tile (two
// halo rows on either side: ONE predicated load per lane, rows 0 / 1 upwards, rows 2 / 3 downwards), columns +-2 come from
// the vector itself and two DPP moves per side (edge lanes: an 8-byte load), frames t+-2 are registers.
// Two-point z / t axes (forward stencil) stay on the one-site kernel (the host does not launch this one then).
// =============================================================================================
template <typename T, int V> __device__ __forceinline__ Vec<T, V> shfl_up32_t(const Vec<T, V>& v) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = __shfl_up(v.v[i], 32, 64);
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> shfl_down32_t(const Vec<T, V>& v) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = __shfl_down(v.v[i], 32, 64);
    return r;
}
// the two elements just outside a lane's vector (x(col0-2), x(col0-1) or x(col0+V), x(col0+V+1)): one 2-element load
template <typename T> struct E2 { T a, b; };
template <typename T> __device__ __forceinline__ E2<T> ldu_e2(const T* ubase, unsigned voff) {
    const Vec<T, 2> v = *reinterpret_cast<const Vec<T, 2>*>(reinterpret_cast<const char*>(ubase) + voff);
    return E2<T>{v.v[0], v.v[1]};
}

template <typename T> __device__ __forceinline__ E2<T> buf_ld_e2(Rsrc r, unsigned off) {
    if constexpr (sizeof(T) == 4) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
        // through ONE 64-bit integer: read as v.x / v.y the compiler narrows the load to its first dword (hipcc 7.2, -O3) and .b is garbage
        const unsigned long long q = __builtin_bit_cast(unsigned long long, v);
        return E2<T>{__builtin_bit_cast(T, (unsigned)q), __builtin_bit_cast(T, (unsigned)(q >> 32))};
    } else {
        const Vec<T, 2> v = buf_ld<T, 2>(r, off);
        return E2<T>{v.v[0], v.v[1]};
    }
}

// T: float (4 columns per 16-byte lane) or -- round 4 -- double (2 columns: the +-2 column neighbours are then whole neighbour lanes)
template <int M, bool TWIN, int CHEB = 0, typename T = float, bool RAGGED = false>
__global__ __launch_bounds__(ST_THREADS, TV_WAVES ? TV_WAVES : (M >= 6 ? 2 : 3)) void k_normal_stream_cen(DG g, WT<T> w, NormalArgsT<T> a, int zchunk, int nchunks) {
    constexpr int V = 16 / (int)sizeof(T);
    using VT = Vec<T, V>;
    __shared__ double sm[16];
    const int lane = (int)threadIdx.x, wave = (int)threadIdx.y;
    const int row = lane >> 4, lx = lane & 15;
    const int nxv = (g.nx + V - 1) / V;
    const int tiles_x = (nxv + ST_BCV - 1) / ST_BCV, tiles_y = (g.ny + ST_BR - 1) / ST_BR;
    const int Mg = TWIN ? g.m : M;
    const int nwin = TWIN ? (Mg + NS_TWN - 1) / NS_TWN : 1;
    const long long ntiles = (long long)tiles_x * tiles_y, total = ntiles * nchunks * nwin, per_xcd = (total + 7) / 8;
    const long long lid = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    double acc0 = 0.0, acc1 = 0.0;
    if (lid < total) {
        // (64-bit divisions run on the vector unit: say that the results are wave-uniform, or descriptors and scalar offsets derived from them
        // are wrapped in a loop over their values)
        const int win = __builtin_amdgcn_readfirstlane((int)(lid / (ntiles * nchunks)));
        const int chunk = __builtin_amdgcn_readfirstlane((int)((lid / ntiles) % nchunks)), tile = __builtin_amdgcn_readfirstlane((int)(lid % ntiles));
        const int t0 = TWIN ? win * NS_TWN : 0;
        const int bx = tile % tiles_x, by = tile / tiles_x;
        const int col0 = (bx * ST_BCV + (wave % ST_NWX) * 16 + lx) * V, y = (by * ST_NWY + wave / ST_NWX) * 4 + row;
        const bool ok = (col0 < g.nx) && (y < g.ny);
        const unsigned voff = ok ? (unsigned)(((long long)y * g.rp + col0) * (long long)sizeof(T)) : 0u;
        const unsigned row_bytes = (unsigned)g.rp * (unsigned)sizeof(T);
        const int zs = chunk * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        const VT zero = vsplat<T, V>(T(0));
        VT mf2 = vsplat<T, V>(T(1));
        if (g.ta) {
            const VT mf = mask_factor<T, V>(g, w.sf, ok ? y : 0, ok ? col0 : 0);
            mf2 = (w.wt * w.wt) * (mf * mf);
        }
        const T wz2 = g.za ? w.wz * w.wz : T(0);
        // a term exists iff the channel it comes from does: backward term <=> p_a >= 2, forward term <=> p_a <= n_a - 3
        const T m_pr = (ok && y >= 2) ? T(1) : T(0), m_nr = (ok && y + 2 < g.ny) ? T(1) : T(0);
        T m_cb[V], m_cf[V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            m_cb[i] = (ok && col0 + i >= 2 && col0 + i < g.nx) ? T(1) : T(0);      // (col < nx: pad columns of a ragged pitched row stay zero)
            m_cf[i] = (ok && col0 + i + 2 < g.nx) ? T(1) : T(0);
        }
        const bool want_up = (row <= 1) && ok && (y >= 2), want_dn = (row >= 2) && ok && (y + 2 < g.ny);
        const unsigned hoff = want_up ? voff - 2u * row_bytes : voff + 2u * row_bytes;
        const bool want_le = (lx == 0) && ok && (col0 >= 2), want_re = (lx == 15) && ok && (col0 + V < g.nx);
        const unsigned eoff = want_le ? voff - 2u * (unsigned)sizeof(T) : voff + 16u;
        const int NV = ((TWIN ? Mg : g.m) - t0 < M) ? (TWIN ? Mg : g.m) - t0 : M;          // frames of this window (k_normal_stream: why it is opaque)
        auto fvalid = [&](int t) { return t < NV; };
        auto foff = [&](int t) { return (long long)(t0 + t) * g.s_t; };
        auto plane = [&](int zl) { return g.za ? zplane<T>(g, a.x, a.xp, a.xn, 2, zl) : ((zl >= 0 && zl < g.nz) ? a.x + (long long)zl * g.s_z : nullptr); };
        // Round 5 (late): the structure of k_normal_stream -- descriptors instead of branches around loads, planes loaded in place (XA / XB
        // swap roles, two steps per loop iteration), epilogue operands and the window's seam frames requested ahead of their use.  One
        // descriptor per FRAME here: with k_normal_stream's per-plane descriptors + scalar frame offsets the M = 8 instantiations of this
        // kernel need 256 VGPRs and 28 - 72 B of scratch (they hold four more seam frames and two-element border loads).
        const int fbytes = (int)(g.s_t * (long long)sizeof(T));
        const unsigned boff = ok ? voff : BUF_OOB;
        const unsigned bhoff = (want_up || want_dn) ? hoff : BUF_OOB, beoff = (want_le || want_re) ? eoff : BUF_OOB;
        auto frame = [&](const T* pl, int t, bool valid) { return buf_rsrc<T>(pl + foff(t), pl != nullptr && valid && fvalid(t), fbytes); };
        // the frames outside a window that its time differences reach: t0 - 2, t0 - 1 (Wp2, Wp1), t0 + M, t0 + M + 1 (Wn1, Wn2)
        const bool w_prev = TWIN && g.ta && t0 > 0;
        auto seam = [&](const T* pl, int t, bool valid) { return buf_rsrc<T>(pl + foff(t), pl != nullptr && valid && TWIN && g.ta && t0 + t >= 0 && t0 + t < Mg, fbytes); };
        // same delayed-store structure as k_normal_stream, on each of the two stride-2 plane lattices: step z finishes plane z-2
        VT XA[M], XB[M], R[M], H[M];
        E2<T> E[M];
        for (int par = 0; par < 2; ++par) {
            const int zfirst = zs + par;
            if (zfirst >= ze) break;
            VT Wp1 = zero, Wp2 = zero, Wn1 = zero, Wn2 = zero;
            {
                const T* pp = g.za ? plane(zfirst - 2) : nullptr;
                const T* pc = plane(zfirst);
                if (TWIN) {
                    Wp2 = buf_ld<T, V>(seam(pc, -2, true), boff);
                    Wp1 = buf_ld<T, V>(seam(pc, -1, true), boff);
                    Wn1 = buf_ld<T, V>(seam(pc, M, true), boff);
                    Wn2 = buf_ld<T, V>(seam(pc, M + 1, true), boff);
                }
#pragma unroll
                for (int t = 0; t < M; ++t) {
                    XB[t] = buf_ld<T, V>(frame(pp, t, true), boff);
                    const Rsrc rc = frame(pc, t, true);
                    XA[t] = buf_ld<T, V>(rc, boff);
                    R[t] = zero;
                    H[t] = buf_ld<T, V>(rc, bhoff);
                    E[t] = buf_ld_e2<T>(rc, beoff);
                }
            }
            NsEpiIn<T, V> ein{zero, zero, zero, zero};
            auto step = [&](auto first_tag, int z, VT (&Cc)[M], VT (&Pp)[M]) __attribute__((always_inline)) {
                constexpr bool FIRST = decltype(first_tag)::value;
                st_sync_plane();
                const bool in_chunk = (z < ze), next_in = (z + 2 < ze);
                const int gz = g.z0 + z;
                const T mz = (g.za && gz >= 2 && gz < g.nzg) ? wz2 : T(0);
                const T* pn = (next_in || (g.za && in_chunk)) ? plane(z + 2) : nullptr;
                VT cold1 = zero, cold2 = zero;         // x(z, t-1), x(z, t-2)
                if (w_prev && in_chunk) { cold1 = Wp1; cold2 = Wp2; }
#pragma unroll
                for (int t = 0; t < M; ++t) {
                    if (t >= (RAGGED ? 1 : TV_NS_EXIT_FROM(M)) && t >= NV) break;                 // (k_normal_stream: what the early exits are for)
                    st_sync_frame();
                    const int tg = t0 + t;
                    const VT c = Cc[t], h = H[t], xm = Pp[t];
                    const VT dz = ns_mul<T, V>(mz, c, xm);
                    const VT rfin = ns_sub<T, V>(R[t], dz);
                    if (in_chunk) {
                        // rows y-2 / y+2: rows 2, 3 take y-2 from rows 0, 1 of the tile (32 lanes up), rows 0, 1 from the halo load
                        const VT sup = shfl_up32_t<T, V>(c), sdn = shfl_down32_t<T, V>(c);
                        const VT pr = (row <= 1) ? h : sup, nr = (row >= 2) ? h : sdn;
                        VT r = m_pr * (c - pr) - m_nr * (nr - c);
                        {
                            // the last two elements of the lane on the left, the first two of the lane on the right (V == 2: its whole vector)
                            const T l2 = dpp_from_left(c.v[V - 2]), l3 = dpp_from_left(c.v[V - 1]);       // executed by every lane
                            const T r0 = dpp_from_right(c.v[0]), r1 = dpp_from_right(c.v[1]);
                            const T lm[2] = {(lx == 0) ? E[t].a : l2, (lx == 0) ? E[t].b : l3};           // x(col0-2), x(col0-1)
                            const T rp[2] = {(lx == 15) ? E[t].a : r0, (lx == 15) ? E[t].b : r1};         // x(col0+V), x(col0+V+1)
#pragma unroll
                            for (int i = 0; i < V; ++i) {
                                const T xb = (i >= 2) ? c.v[(i >= 2) ? i - 2 : 0] : lm[(i < 2) ? i : 0];
                                const T xf = (i + 2 < V) ? c.v[(i + 2 < V) ? i + 2 : 0] : rp[(i + 2 >= V) ? i + 2 - V : 0];
                                r.v[i] += m_cb[i] * (c.v[i] - xb) - m_cf[i] * (xf - c.v[i]);
                            }
                        }
                        if (g.ta) {
                            VT tt = zero;
                            if (tg >= 2) tt = tt + (c - cold2);
                            if (tg + 2 < Mg) {
                                if (t + 2 < M) tt = tt - (Cc[(t + 2 < M) ? t + 2 : t] - c);
                                else if (TWIN) tt = tt - (((t + 2 == M) ? Wn1 : Wn2) - c);
                            }
                            r = r + mf2 * tt;
                        }
                        R[t] = ns_add<T, V>(r, dz);
                    }
                    cold2 = cold1;
                    cold1 = c;
                    // ---- epilogue of plane z-2 ----------------------------------------------------------------------------------------
                    if constexpr (!FIRST) {
                        const long long fo = (long long)(z - 2) * g.s_z + foff(t);
                        VT ax;
#pragma unroll
                        for (int i = 0; i < V; ++i) ax.v[i] = xm.v[i] + a.rho * (T(0.25) * rfin.v[i]);
                        ns_epilogue<T, V, CHEB, true>(a, fo, boff, xm, ax, ein, acc0, acc1, fbytes, 0);          // (per-frame descriptors here: this kernel has no SGPRs to spare, see below)
                    }
                    // ---- operands of the next epilogue (frame t + 1 of this step or frame 0 of the lattice's next step), then plane z + 2 into
                    // the slot of plane z - 2 with its halo rows / border elements and the seam frames ---------------------------------------
                    {
                        const bool more = (t + 1 < M) && fvalid((t + 1 < M) ? t + 1 : t);
                        const long long fo_n = more ? (long long)(z - 2) * g.s_z + foff(t + 1) : (long long)z * g.s_z + foff(0);
                        if constexpr (CHEB != 0) ns_epi_load_buf<T, V, CHEB>(a, fo_n, more ? !FIRST : (z < ze), fbytes, boff, 0, ein);
                    }
                    {
                        Pp[t] = buf_ld<T, V>(frame(pn, t, true), boff);
                        const Rsrc rn = frame(pn, t, next_in);
                        H[t] = buf_ld<T, V>(rn, bhoff);
                        E[t] = buf_ld_e2<T>(rn, beoff);
                        if (TWIN && t == 0) {
                            Wp2 = buf_ld<T, V>(seam(pn, -2, next_in), boff);
                            Wp1 = buf_ld<T, V>(seam(pn, -1, next_in), boff);
                        }
                        if (TWIN && t == M - 2) Wn1 = buf_ld<T, V>(seam(pn, M, next_in), boff);
                        if (TWIN && t == M - 1) Wn2 = buf_ld<T, V>(seam(pn, M + 1, next_in), boff);
                    }
                }
            };
            step(std::true_type{}, zfirst, XA, XB);
            for (int z = zfirst + 2; z < ze + 2; z += 4) {          // the step behind the chunk only finishes the lattice's last plane
                step(std::false_type{}, z, XB, XA);
                if (z + 2 >= ze + 2) break;
                step(std::false_type{}, z + 2, XA, XB);
            }
        }
    }
    if (a.part0 == nullptr) return;          // (Chebyshev step without dot products)
    acc0 = block_sum(acc0, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part0[blockIdx.x] = acc0;
    acc1 = block_sum(acc1, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part1[blockIdx.x] = acc1;
}

}  // namespace tv
