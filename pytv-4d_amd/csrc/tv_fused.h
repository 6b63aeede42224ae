// tv_fused.h -- ONE-SWEEP Chambolle-Pock iteration (README.md:145-157 of the reference) for all four schemes,
// fp32 and (round 3) fp64, 16-byte lanes (4 or 2 columns).  The same sweep, with the template parameter ALG, is the dual side of
// the ADMM outer iteration (ALG_ADMM: tv_admm_fused) and the TV part of Chambolle-Pock with a data-fidelity operator (ALG_CPOP:
// tv_cpop_fused) -- see the comment at ALG_CP below.
//
// The two-kernel form (tv_cp_dual + tv_cp_primal) reads the dual variable q twice and writes it once
// per iteration: 30 words/voxel at Nd = 8.  Here the primal update of plane z-1 is done in the same
// sweep that produces q'(z), lagging one plane behind, so q is read once and written once:
//
//   step z, frame t (thread = one (row, 4-col) site, marching z inside a z-chunk, frames unrolled):
//     q'(z,t) = proj(q + sigma D x)                       (x(z-1), x(z) in registers, x(z+1) loaded)
//     R[t]  (adjoint accumulator of plane z-1)  -= wz q'_zdown(z,t)        -> now complete:
//     x_out(z-1,t) = x(z-1,t) - tau p' - tau s R[t],  p' = (p + sigma_A (x - x0)) / (1 + sigma_A)
//     R[t]  := own-site terms of q'(z,t) + row / col neighbour terms (wave shuffles)
//              + wz q'_zup(z-1,t) (carried) + wt q'_tup(z,t-1) (carried);   R[t-1] -= wt q'_tdown(z,t)
//
// A wave covers a CP_TR-row x (64 / CP_TR * 4)-col tile -- 8 rows x 8 lanes (32 columns) since round 2, 4 rows x 16 lanes
// (64 columns) in round 1 (-DTV_FUSED_TR=4): row and column neighbours of both x and q' are lane shuffles, so there is
// no LDS tile and no barrier.  Terms that live in ANOTHER wave's tile or another z-chunk (tile-edge rows / columns,
// chunk-edge planes) are left out by the sweep and added by the thin fix-up kernel k_cp_fixup afterwards (it touches
// only those rows / planes: ~1.65 words/voxel with 8-row tiles, ~2.5 with 4-row tiles).
// x is ping-ponged (x_in -> x_out) because neighbouring tiles read old halo values at their own pace.
#pragma once
#include "tv_device.h"
#include "tv_stencil.h"

namespace tv {

using F4 = Vec<float, 4>;

// Wave tile of the one-sweep CP kernel: CP_TR rows x CP_TL lanes (4 columns each); CP_NW waves side by side form the
// block tile (CP_TR rows x CP_BC columns) and hand their tile-edge COLUMN terms to each other through LDS.
// Default (round 2): 8 x 8 lanes per wave, 8 waves per block = 8 rows x 256 columns, one 512-thread block per CU at
// M = 8 (the per-thread R / U slots take 128 KiB).  Only 2 of 8 rows are tile-edge rows, which halves the row fix-up
// and the x halo rows, and the block tile keeps the 256-column period of the sparse column-edge fix-up.
// Measured on the north-star volume (profiles/r2_ab_sweep_tr8nw8.txt): iteration 37.7 -> 35.7 ms (sweep 33.4 -> 32.8,
// fix-up 4.3 -> 2.9 ms); upwind / downwind / central -1..2 %; 128x16x1024x1024 40.2 -> 38.3 ms.  Round 1's 4 x 16 lanes,
// 4 waves: -DTV_FUSED_TR=4 -DTV_FUSED_NW=4.  (8-row tiles in a 4-wave block, 128 columns: row fix-up halves but the
// column-edge fix-up goes 0.5 -> 2.1 ms -- no net gain, round 1.)
#ifndef TV_FUSED_TR
#define TV_FUSED_TR 8
#endif
#ifndef TV_FUSED_NW
#define TV_FUSED_NW 8
#endif
#ifndef TV_ADMM_T_NT
#define TV_ADMM_T_NT 0           // 1: the sparse t' stores of the ADMM sweep non-temporal like its u / r stores (A/B)
#endif
#ifndef TV_FUSED_PFQ_TWIN
#define TV_FUSED_PFQ_TWIN 1      // the central dual prefetch (PFQ) in the windowed (M > 8) instantiations too (round 3)
#endif
#ifndef TV_FUSED_PFQ
#define TV_FUSED_PFQ 2
#endif
#ifndef TV_FUSED_PFX
#define TV_FUSED_PFX 1
#endif
#ifndef TV_FUSED_XE
#define TV_FUSED_XE 0
#endif
#ifndef TV_FUSED_EA
#define TV_FUSED_EA 1
#endif
constexpr int CP_NW = TV_FUSED_NW;
constexpr int CP_TR = TV_FUSED_TR, CP_TL = 64 / CP_TR, CP_WC = 4 * CP_TL, CP_BC = CP_NW * CP_WC;
constexpr int CP_LSH = (CP_TL == 16) ? 4 : 3;
static_assert(CP_TR == 4 || CP_TR == 8, "wave tile: 4 x 16 or 8 x 8");
// the same for lanes of V columns (V = 4 floats, 2 doubles: round 3)
template <int V> struct CPG { static constexpr int WC = V * CP_TL, BC = CP_NW * V * CP_TL; };

struct FusedCoord {
    int lane, row, lx, col0, y, zs, ze;
    bool ok;
    long long inpl;
};

// block (64, CP_NW): wave = threadIdx.y covers columns [CP_WC w, CP_WC w + CP_WC) of a CP_TR-row x CP_BC-col block tile
// (default: 8 waves side by side, 8 rows x 256 columns)
__device__ __forceinline__ FusedCoord fused_coord(const DG& g, int zchunk, int chunk0) {
    FusedCoord c;
    c.lane = (int)threadIdx.x;
    c.row = c.lane >> 4;
    c.lx = c.lane & 15;
    const int nxv = g.nx / 4;
    const int tiles_x = (nxv + 63) / 64;
    const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
    c.col0 = (bx * 64 + (int)threadIdx.y * 16 + c.lx) * 4;
    c.y = by * 4 + c.row;
    c.ok = (c.col0 < g.nx) && (c.y < g.ny);
    c.zs = ((int)blockIdx.y + chunk0) * zchunk;
    c.ze = (c.zs + zchunk < g.nz) ? c.zs + zchunk : g.nz;
    c.inpl = (long long)c.y * g.rp + c.col0;
    return c;
}

// coordinates of the one-sweep CP kernel: block (64, 4), wave = threadIdx.y covers columns [CP_WC w, CP_WC (w + 1))
// of a CP_TR-row x CP_BC-column block tile
template <int V = 4> __device__ __forceinline__ FusedCoord cp_coord(const DG& g, int zchunk, int chunk0) {
    FusedCoord c;
    c.lane = (int)threadIdx.x;
    c.row = c.lane >> CP_LSH;
    c.lx = c.lane & (CP_TL - 1);
    const int nxv = (g.nx + V - 1) / V;
    const int tiles_x = (nxv + CP_NW * CP_TL - 1) / (CP_NW * CP_TL);
    const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
    c.col0 = (bx * CP_NW * CP_TL + (int)threadIdx.y * CP_TL + c.lx) * V;
    c.y = by * CP_TR + c.row;
    c.ok = (c.col0 < g.nx) && (c.y < g.ny);
    c.zs = ((int)blockIdx.y + chunk0) * zchunk;
    c.ze = (c.zs + zchunk < g.nz) ? c.zs + zchunk : g.nz;
    c.inpl = (long long)c.y * g.rp + c.col0;
    return c;
}

// does the sweep leave a term of this site-vector to the fix-up kernel?  (shared by both kernels)
// (central: the adjoint of every channel reaches both ways, so it counts as "up" and "down" here)
// time windows (M > 8): the sweep works on windows of CP_TWN frames; the adjoint terms that cross a window seam
// are the fix-up's.  Is frame t a seam frame with a missing time term?  (central: the neighbour's channel must be
// defined, i.e. the neighbour frame must be an interior one)
#ifndef TV_TWN
#define TV_TWN 8                 // (tv_dstream.h: the window-of-4 experiment)
#endif
#ifndef TV_WAVES
#define TV_WAVES 0
#endif
constexpr int CP_TWN = TV_TWN;
template <int S> __device__ __forceinline__ bool is_seam_frame(const DG& g, int t) {
    constexpr bool UP = (S != DOWNWIND), DN = (S != UPWIND), CEN = (S == CENTRAL);
    if (g.m <= CP_TWN || !g.ta) return false;
    const int k = t % CP_TWN;
    return (UP && k == 0 && t >= 1) || (DN && k == CP_TWN - 1 && t <= g.m - (CEN ? 3 : 2));
}

template <int S, bool XW, int V = 4>
__device__ __forceinline__ bool fused_needs_fixup(const DG& g, int zl, int y, int col0, int zchunk) {
    constexpr bool UP = (S != DOWNWIND), DN = (S != UPWIND);
    constexpr int CM = XW ? CPG<V>::BC - 1 : CPG<V>::WC - 1;   // column period of the tiles whose edges are left to the fix-up
    bool f = false;
    // rows: every wave tile (CP_TR rows); columns: only the BLOCK tile edges (CP_BC columns) -- the CP_NW waves
    // of a block hand their edge columns to each other through LDS inside the sweep
    if (UP) f = f || ((y & (CP_TR - 1)) == 0 && y >= 1) || ((col0 & CM) == 0 && col0 >= 1);
    if (DN) f = f || ((y & (CP_TR - 1)) == CP_TR - 1 && y <= g.ny - 2) || ((col0 & CM) == CM - (V - 1) && col0 + V <= g.nx - 1);
    if (g.za) {
        const int gz = g.z0 + zl;
        const int zs = (zl / zchunk) * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        if (UP) f = f || (zl == zs && gz >= 1);
        if (DN) f = f || (zl == ze - 1 && gz <= g.nzg - 2);
    }
    return f;
}
// ... including the time-window seams
template <int S, bool XW, int V = 4>
__device__ __forceinline__ bool fused_needs_fixup(const DG& g, int zl, int y, int col0, int zchunk, int t) {
    if (is_seam_frame<S>(g, t)) return true;
    return fused_needs_fixup<S, XW, V>(g, zl, y, col0, zchunk);
}


// uniform 64-bit base (SGPR pair) + 32-bit per-lane byte offset: lets the compiler use the scalar-base form of
// global_load / global_store (one VGPR of address for every stream of the site instead of a 64-bit VGPR pair each)
__device__ __forceinline__ F4 ldu(const float* ubase, unsigned voff) {
    return *reinterpret_cast<const F4*>(reinterpret_cast<const char*>(ubase) + voff);
}
__device__ __forceinline__ float ldu1(const float* ubase, unsigned voff) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ubase) + voff);
}
// dtype-generic forms (round 3: the one-sweep kernel, the streaming kernels): a lane holds V = 16 / sizeof(T) columns
template <typename T, int V> __device__ __forceinline__ Vec<T, V> ldu_t(const T* ubase, unsigned voff) {
    return *reinterpret_cast<const Vec<T, V>*>(reinterpret_cast<const char*>(ubase) + voff);
}
template <typename T> __device__ __forceinline__ T ldu1_t(const T* ubase, unsigned voff) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(ubase) + voff);
}
template <typename T, int V> __device__ __forceinline__ void stu_t(T* ubase, unsigned voff, const Vec<T, V>& v) {
    *reinterpret_cast<Vec<T, V>*>(reinterpret_cast<char*>(ubase) + voff) = v;
}
// streamed-once data (the dual variable q, x0, p): non-temporal loads / stores (-DTV_FUSED_NT=1: A/B of round 3 -- a copy with
// 8 read + 8 write streams and this kernel's 512-thread blocks gains 1.3 % from it, tools/archive/bwtest3 "mix"; mixing non-temporal loads
// with plain stores LOSES 7 %)
#ifndef TV_FUSED_NT
#define TV_FUSED_NT 1
#endif
__device__ __forceinline__ F4 ldu_s(const float* ubase, unsigned voff) {
    const F4* p = reinterpret_cast<const F4*>(reinterpret_cast<const char*>(ubase) + voff);
#if TV_FUSED_NT
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    F4 r;
    r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z; r.v[3] = v.w;
    return r;
#else
    return *p;
#endif
}
__device__ __forceinline__ void stu_s(float* ubase, unsigned voff, const F4& v) {
    F4* p = reinterpret_cast<F4*>(reinterpret_cast<char*>(ubase) + voff);
#if TV_FUSED_NT
    typedef float nt_f4 __attribute__((ext_vector_type(4)));
    nt_f4 w;
    w.x = v.v[0]; w.y = v.v[1]; w.z = v.v[2]; w.w = v.v[3];
    __builtin_nontemporal_store(w, reinterpret_cast<nt_f4*>(p));
#else
    *p = v;
#endif
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> ldu_s_t(const T* ubase, unsigned voff) {
    const Vec<T, V>* p = reinterpret_cast<const Vec<T, V>*>(reinterpret_cast<const char*>(ubase) + voff);
#if TV_FUSED_NT
    typedef T nt_v __attribute__((ext_vector_type(V)));
    const nt_v v = __builtin_nontemporal_load(reinterpret_cast<const nt_v*>(p));
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = v[i];
    return r;
#else
    return *p;
#endif
}
template <typename T, int V> __device__ __forceinline__ void stu_s_t(T* ubase, unsigned voff, const Vec<T, V>& v) {
    Vec<T, V>* p = reinterpret_cast<Vec<T, V>*>(reinterpret_cast<char*>(ubase) + voff);
#if TV_FUSED_NT
    typedef T nt_v __attribute__((ext_vector_type(V)));
    nt_v w;
#pragma unroll
    for (int i = 0; i < V; ++i) w[i] = v.v[i];
    __builtin_nontemporal_store(w, reinterpret_cast<nt_v*>(p));
#else
    *p = v;
#endif
}
// ---- raw buffer access (round 5, streaming kernels) ---------------------------------------------------------------------------------
// A wave-uniform descriptor (base = one frame of one plane of one stream, num_records = bytes of a frame, 0 when the frame / plane /
// stream does not exist) + a per-lane byte offset (BUF_OOB for lanes that must not take part).  The hardware's range check is the
// predicate: such a load returns 0, such a store is dropped -- NO branch and NO phi around a memory operation.  That is what lets the
// compiler count (s_waitcnt vmcnt(n)): a load under a uniform branch reaches its consumer through a copy that has to wait for the load
// where it was ISSUED, which with one in-order counter means vmcnt(0) once per frame (tv_subgrad2.h found the same for its kernel).
using Rsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned BUF_OOB = 0x80000000u;        // beyond any frame (the hosts send frames of >= 2^31 bytes to other kernels)
typedef int buf_v4i __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ Rsrc buf_rsrc(const T* base, bool valid, int nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, valid ? nbytes : 0, 0x00020000);
}
// soff: a wave-uniform byte offset added to the descriptor's base (the frame inside a plane: one descriptor per plane and stream instead of one
// per frame -- fewer scalar instructions and live SGPRs).  The descriptors that take it span the whole plane, so a lane is in range
// whether or not the hardware counts soff in its range check.
template <typename T, int V, int AUX = 0> __device__ __forceinline__ Vec<T, V> buf_ld(Rsrc r, unsigned off, int soff = 0) {
    static_assert(sizeof(Vec<T, V>) == 16, "16-byte lanes");
    const buf_v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, soff, AUX);
    return __builtin_bit_cast(Vec<T, V>, v);
}
template <typename T, int AUX = 0> __device__ __forceinline__ T buf_ld1(Rsrc r, unsigned off, int soff = 0) {
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, soff, AUX));
    } else {
        typedef int v2i __attribute__((ext_vector_type(2)));
        const v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, soff, AUX);
        return __builtin_bit_cast(T, v);
    }
}
template <typename T, int V, int AUX = 0> __device__ __forceinline__ void buf_st(Rsrc r, unsigned off, const Vec<T, V>& v, int soff = 0) {
    static_assert(sizeof(Vec<T, V>) == 16, "16-byte lanes");
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(buf_v4i, v), r, (int)off, soff, AUX);
}
constexpr int BUF_NT = 2;                        // aux bit 1 = nt on gfx950 (what __builtin_nontemporal_load / _store emit)

__device__ __forceinline__ void stu(float* ubase, unsigned voff, const F4& v) {
    *reinterpret_cast<F4*>(reinterpret_cast<char*>(ubase) + voff) = v;
}

__device__ __forceinline__ F4 shfl_up16(const F4& v) {
    F4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = __shfl_up(v.v[i], 16, 64);
    return r;
}
__device__ __forceinline__ F4 shfl_down16(const F4& v) {
    F4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = __shfl_down(v.v[i], 16, 64);
    return r;
}

// the same one row of the CP wave tile up / down (CP_TL lanes)
template <typename T, int V> __device__ __forceinline__ Vec<T, V> cp_shfl_up(const Vec<T, V>& v) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = __shfl_up(v.v[i], CP_TL, 64);
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> cp_shfl_down(const Vec<T, V>& v) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = __shfl_down(v.v[i], CP_TL, 64);
    return r;
}

template <typename T> struct FusedArgsT {
    const T* x_in;
    const T* xp;          // plane z0-1 of x_in (or nullptr)
    const T* xn;          // plane z0+nz of x_in (or nullptr)
    T* q;
    const T* x0;
    T* p;
    T* x_out;
    T sigma, inv_lambda, tau, sigma_a, inv_1p_sigma_a;
    double* part_tv;
    double* part_fid;
    int full_store;       // ALG_ADMM: bit 0 = every sample of t' is stored (z stays recoverable; else only what the fix-up reads), bit 1 = the second partial is |x - x0|^2
                          // ALG_CP: bit 1 = the second partial is 1/2 |x_in - x0|^2 over all sites (TV_CP_FID_OF_INPUT)
    const T* q_in;        // where the dual variable is READ (round 4: q ping-pong, tv_cp_sweep; == q: in place, as before).  Reading one
                          // array and writing another is ~9 % faster than the in-place read-modify-write for this kernel's memory shape
                          // (tools/archive/bwtest4 variant 4: 5.98 against 5.50 TB/s) -- the price is a second q array
    double* part_fid2;    // ALG_CP, full_store bit 2 (TV_CP_FID_BOTH, round 5): per-block partials of 1/2 |x_out - x0|^2 over the complete sites, NEXT TO the
                          // fidelity of the input -- the last sweep of a lagged block returns both, so that no separate reduction closes the block
};
using FusedArgs = FusedArgsT<float>;

// ALG: which inner loop the sweep runs.
//   ALG_CP   (README.md:141-157): q <- proj(q + sigma D x); p, x updated one plane behind; partials: TV, 1/2 |x_out - x0|^2.
//   ALG_ADMM (round 3): the z / u update of scaled-form ADMM and the residual of the NEXT x-solve in one pass:
//            v = D x + u;  z = shrink(v, thresh);  u <- v - z  (in place, the array passed as `q`);  t' = (z - u) - D x;
//            r = (x0 - x) + rho D^T t'  =  [x0 + rho D^T (z - u)] - (I + rho D^T D) x          (one plane behind, array `x_out`)
//            partials: TV(x), <r, r>.  Arguments reused: sigma = thresh = reg / rho, tau = rho, p = the array t' goes to (same
//            layout as q).  Nobody but the fix-up reads t', so by default only the samples the fix-up needs are stored
//            (tile-edge rows / columns, chunk-edge planes, window-seam frames: ~0.3 words per voxel instead of Nd);
//            4 Nd + 6 words per voxel of the kernel trio tv_admm_tu + tv_DT_axpy + tv_normal_op2(rhs) become 2 Nd + 3.
//   ALG_CPOP (round 3): Chambolle-Pock with a user-supplied data-fidelity operator A (README.md:2,148; solvers.ChambollePockOperator):
//            the dual update of ALG_CP and x_out <- x - tau A^T p - tau D^T q' one plane behind; `p` is the image A^T p (read only),
//            x0 is not touched, the second partial is unused.  Its fix-up is the ALG_ADMM one with the coefficient tau.
constexpr int ALG_CP = 0, ALG_ADMM = 1, ALG_CPOP = 2;

// XW: the CP_NW waves of a block exchange their tile-edge column terms through LDS (one barrier per plane)
// TWIN: time windows for volumes with more than CP_TWN frames -- grid z = window, the block works on the frames
// [t0, t0 + M) of the volume (M = CP_TWN), reads x of the frame on either side of its window for the time differences
// and leaves the adjoint terms that cross a window seam to the fix-up (exactly like the z-chunk edges)
// T: float (4 columns per 16-byte lane) or double (2 columns, round 3): the same tile in lanes, half as wide in columns.
template <int S, int M, bool XW, bool TWIN = false, typename T = float, int ALG = ALG_CP>
__global__ __launch_bounds__(64 * CP_NW, TV_WAVES ? TV_WAVES : 2) void k_cp_fused(DG g, WT<T> w, FusedArgsT<T> a, int zchunk, int chunk0) {
    constexpr int V = 16 / (int)sizeof(T);
    using VT = Vec<T, V>;
    __shared__ double sm[16];
    const FusedCoord c = cp_coord<V>(g, zchunk, chunk0);
    const int t0 = TWIN ? (int)blockIdx.z * CP_TWN : 0;       // first frame of this block's window
    const int Mg = TWIN ? g.m : M;                             // frames of the volume
    const unsigned voff = (unsigned)c.inpl * (unsigned)sizeof(T);          // byte offset of this lane's vector inside a frame (frames <= 2^30 px)
    const unsigned row_bytes = (unsigned)g.rp * (unsigned)sizeof(T);
    // UP: some channel's adjoint takes y^(p-e) (forward differences; central: every channel); DN: ... y^(p+e).
    // CEN: central -- ONE channel per axis plays both roles, has no own-site term and is defined on interior
    // points only; its two-point z / t axes (z_fwd / t_fwd) behave like upwind axes.
    constexpr bool UP = (S != DOWNWIND), DN = (S != UPWIND), CEN = (S == CENTRAL);
    constexpr bool NEXT = UP, PREV = DN;       // forward differences need x(+e), backward x(-e)
    const bool z_fwd = CEN && g.z_two, t_fwd = CEN && g.t_two;
    const VT zero = vsplat<T, V>(T(0));
    const VT mf = g.ta ? mask_factor<T, V>(g, w.sf, c.ok ? c.y : 0, c.ok ? c.col0 : 0) : vsplat<T, V>(T(1));
    const T s = (S == HYBRID) ? Consts<T>::inv_sqrt2() : (CEN ? T(0.5) : T(1));
    double acc_tv = 0.0, acc_fid = 0.0, acc_fid2 = 0.0;
    // x planes z (C) and z-1 (P) live in registers; the adjoint accumulators R (plane z-1 waiting
    // for its z+1 term) and the carried z-up terms U live in LDS, private per thread ([frame][thread]:
    // conflict-free 16-byte lanes, no barrier needed) -- 4 M vectors of state do not fit the VGPR file
    __shared__ VT lds_R[M][64 * CP_NW];
    __shared__ VT lds_U[M][64 * CP_NW];
    // column terms that cross the 64-column wave tiles INSIDE the block: each wave publishes, per plane and
    // frame, the col-up value of its last column and the col-down value of its first column (per row);
    // the neighbouring wave adds them one plane later (after the per-plane barrier), double-buffered
    __shared__ T edge_cu[XW ? 2 : 1][XW ? M : 1][CP_NW][CP_TR];
    __shared__ T edge_cd[XW ? 2 : 1][XW ? M : 1][CP_NW][CP_TR];
    // edge_flag[w] = number of planes of this chunk whose edge columns wave w has published.  A wave starts
    // plane z only after both neighbours published plane z-1, so neighbouring waves stay within one plane of
    // each other (that is what makes two buffers enough) -- a pairwise hand-off, not a block-wide barrier.
    __shared__ volatile int edge_flag[CP_NW];
    // XE (round 2, -DTV_FUSED_XE=1, OFF by default): the waves of a block also hand each other the first / last COLUMN of x
    // of the plane they are about to work on (published together with the column terms of the plane just finished, same
    // counters, double-buffered by plane parity) instead of loading those 4-byte elements.  Parity green, but no gain:
    // hybrid 36.4 / 36.4 vs 35.9 / 37.4 ms, upwind 26.7 / 25.4 vs 23.7 / 25.8, central equal (profiles/r2_ab_xe.txt) --
    // the edge loads are not what the sweep's ~14 GB of over-read consist of.
    constexpr bool XE = XW && (TV_FUSED_XE != 0);
    __shared__ T edge_xl[XE ? 2 : 1][XE ? M : 1][CP_NW][CP_TR];
    __shared__ T edge_xr[XE ? 2 : 1][XE ? M : 1][CP_NW][CP_TR];
    const int wave = (int)threadIdx.y;
    if (XW && c.lane == 0) edge_flag[wave] = 0;
    const int tid = (int)threadIdx.y * 64 + (int)threadIdx.x;
    VT C[M], P[M];
    // EA (round 2): the 4-byte column neighbours across the wave-tile border (first / last lane of a row segment) are requested
    // ONE PLANE AHEAD, in the same frame as the wide load of that plane -- i.e. while the neighbouring wave fetches the very
    // line they live in.  Requested a plane later (as before) the line had left the L2 again (a CU streams ~0.7 MB per plane
    // through its 128 KiB share) and two thirds of these loads went to the fabric as their own requests: 9.8 GB per sweep of
    // the north-star volume on the read counter (tools/archive/pmc_calib.sh shows the same +14 % on a kernel with exactly known bytes).
    constexpr bool EA = (TV_FUSED_EA != 0) && !(XW && (TV_FUSED_XE != 0));
    const bool e_le = PREV && (c.lx == 0) && c.ok && (c.col0 > 0);
    const bool e_re = NEXT && (c.lx == CP_TL - 1) && c.ok && (c.col0 + V < g.nx);
    const unsigned eoff = e_le ? voff - (unsigned)sizeof(T) : voff + 16u;
    T E[EA ? M : 1];
    // PFQ (round 2, central): the four dual channels of the NEXT frame are requested at the top of the current one.  Central
    // has 1 + 4 streams per frame (hybrid: 1 + 8) but hybrid's per-site arithmetic (every channel plays both adjoint roles),
    // so its waves waited ~77 % of the time with too few bytes in flight: sweep 26.6 - 27.4 -> 21.2 - 23.0 ms on the
    // north-star volume (0.52 -> 0.63 of peak; profiles/r2_ab_pfq.txt).  Measured for upwind / downwind too: 3 - 7 % SLOWER
    // there (they already request plane z+1 of x a step ahead), so it stays off; hybrid has no registers for it.
    // Round 5: upwind / downwind too, on wide fp32 frames (XW) -- re-measured on the kernel as it is now, one box, product against
    // -DTV_FUSED_PFQ=2 (profiles/r5f_pfq_ab.txt): ADMM sweep of the configs[4] slab 5.18 -> 4.52 ms (upwind), 5.09 -> 4.51 (downwind);
    // CP sweep of the north-star volume 20.10 -> 19.36 (upwind), 20.85 -> 19.70 (downwind).  Not on narrow frames (XW false: the M >= 5
    // downwind instantiations spill up to 484 B with it) and not in fp64 (unmeasured).  TV_FUSED_PFQ: 0 off, 1 central only, 2 (default) this.
    constexpr bool PFQ = (TV_FUSED_PFQ != 0) && (S == CENTRAL || (TV_FUSED_PFQ == 2 && S != HYBRID && XW && sizeof(T) == 4)) &&
                         (!TWIN || TV_FUSED_PFQ_TWIN != 0);
    VT qpre[PFQ ? 4 : 1];
    if (PFQ) {
#pragma unroll
        for (int k = 0; k < 4; ++k) qpre[k] = zero;
        if (c.ok) {
            const T* qb0 = a.q_in + (long long)c.zs * g.s_dz + (long long)t0 * g.s_t;
            for_each_channel<S>(g, [&](auto slot, int ch) {
                constexpr int k = decltype(slot)::value;
                qpre[k & 3] = ldu_s_t<T, V>(qb0 + (long long)ch * g.s_z, voff);
            });
        }
    }
    {
        const T* pc = zplane<T>(g, a.x_in, a.xp, a.xn, 1, c.zs);
        const T* pp = (PREV && g.za) ? zplane<T>(g, a.x_in, a.xp, a.xn, 1, c.zs - 1) : nullptr;
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const bool fok = c.ok && (t0 + t < Mg);
            C[t] = fok ? ldu_t<T, V>(pc + (long long)(t0 + t) * g.s_t, voff) : zero;
            P[t] = (fok && pp != nullptr) ? ldu_t<T, V>(pp + (long long)(t0 + t) * g.s_t, voff) : zero;
            if (EA) E[t] = (fok && (e_le || e_re)) ? ldu1_t<T>(pc + (long long)(t0 + t) * g.s_t, eoff) : T(0);
            lds_R[t][tid] = zero;
            lds_U[t][tid] = zero;
            if (XE) {
                if (c.lx == 0) edge_xl[0][t][wave][c.row] = C[t].v[0];
                if (c.lx == CP_TL - 1) edge_xr[0][t][wave][c.row] = C[t].v[V - 1];
            }
        }
    }
    if (XW) __syncthreads();      // counters zeroed, the first plane's x edges published

    // finalise plane zf (its x values are in `xv`), adjoint accumulator `racc` (un-scaled)
    // PFX (round 5): the operands of the lagged primal update (x0, p of plane z - 1) are requested at the TOP of the frame that finalises them,
    // through frame descriptors (no branch), instead of inside finalize() behind everything else: one exposed memory round trip per frame
    // less.  pre0 / pre1: those operands when `pre` is set (x0 and p; ADMM: x0; CPOP: A^T p).  Measured with the product library and a
    // -DTV_FUSED_PFX=1 variant interleaved on one box, 24 constructions of the north-star problem each (profiles/r5i_pfx_ab.txt): hybrid
    // sweep 32.8 -> 31.8 ms on average, best allocation 31.0 -> 30.7; ADMM hybrid 14.11 -> 14.01 ms, upwind 10.66 -> 10.48.  Not for central
    // (its M = 8 instantiations spill 36 - 100 B with it) nor the hybrid single-window instantiations for 6 - 8 frames (12 - 88 B).
    constexpr bool PFX = (TV_FUSED_PFX != 0) && XW && sizeof(T) == 4 &&
                         (S == UPWIND || S == DOWNWIND || (S == HYBRID && (TWIN || M <= 5)));
    auto finalize = [&](int zf, int t, const VT& xv, VT racc, bool pre = false, const VT& pre0 = VT{}, const VT& pre1 = VT{}) {
        if (!c.ok || t0 + t >= Mg) return;
        const int eb = (zf - c.zs) & 1;
        if (XW) {
            if (UP && c.lx == 0 && wave > 0) racc.v[0] += edge_cu[eb][t][wave - 1][c.row];
            if (DN && c.lx == CP_TL - 1 && wave < CP_NW - 1) racc.v[V - 1] -= edge_cd[eb][t][wave + 1][c.row];
        }
        const long long foff = (long long)zf * g.s_z + (long long)(t0 + t) * g.s_t;      // uniform
        if constexpr (ALG == ALG_ADMM) {      // r = (x0 - x) + rho D^T t'
            const VT x0v = (PFX && pre) ? pre0 : ldu_s_t<T, V>(a.x0 + foff, voff);
            VT ro;
            double r2 = 0.0, f2 = 0.0;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const T e = x0v.v[i] - xv.v[i];
                ro.v[i] = e + a.tau * (s * racc.v[i]);
                r2 += (double)ro.v[i] * (double)ro.v[i];
                f2 += (double)e * (double)e;
            }
            stu_s_t<T, V>(a.x_out + foff, voff, ro);
            // second partial: <r, r> over the complete sites (what CG starts from), or -- full_store bit 1, the Chebyshev x-solve, which
            // needs no <r, r> -- |x - x0|^2 over ALL sites (the fidelity of the iterate: one word less in the solve's last step)
            if (a.full_store & 2) acc_fid += f2;
            else if (!fused_needs_fixup<S, XW, V>(g, zf, c.y, c.col0, zchunk, t0 + t)) acc_fid += r2;
            return;
        }
        if constexpr (ALG == ALG_CPOP) {      // x_out = (x - tau A^T p) - tau D^T q'   (the arithmetic of tv_DT_axpy2)
            const VT atp = (PFX && pre) ? pre0 : ldu_s_t<T, V>(a.p + foff, voff);
            VT xo;
#pragma unroll
            for (int i = 0; i < V; ++i) xo.v[i] = (xv.v[i] - a.tau * atp.v[i]) - a.tau * (s * racc.v[i]);
            stu_s_t<T, V>(a.x_out + foff, voff, xo);
            return;
        }
        const VT x0v = (PFX && pre) ? pre0 : ldu_s_t<T, V>(a.x0 + foff, voff), pv = (PFX && pre) ? pre1 : ldu_s_t<T, V>(a.p + foff, voff);
        VT pn, xo;
        double e2 = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            pn.v[i] = (pv.v[i] + a.sigma_a * (xv.v[i] - x0v.v[i])) * a.inv_1p_sigma_a;
            xo.v[i] = (xv.v[i] - a.tau * pn.v[i]) - a.tau * (s * racc.v[i]);
            const double e = (double)xo.v[i] - (double)x0v.v[i];
            e2 += 0.5 * e * e;
        }
        stu_s_t<T, V>(a.p + foff, voff, pn);
        stu_s_t<T, V>(a.x_out + foff, voff, xo);
        // (the two accumulators are updated UNCONDITIONALLY with selected VALUES: selecting the accumulator instead makes the compiler keep
        // both in a dynamically indexed stack slot -- 24 B of scratch and a load / store pair per site in every instantiation)
        double add1 = 0.0, add2 = 0.0;
        const bool lagged = (a.full_store & 2) != 0, both = (a.full_store & 4) != 0;       // uniform
        if (!lagged || both) {
            if (!fused_needs_fixup<S, XW, V>(g, zf, c.y, c.col0, zchunk, t0 + t)) { if (lagged) add2 = e2; else add1 = e2; }
        }
        if (lagged) {
            // fidelity of the INPUT iterate over ALL sites (round 4, tv_cp_sweep flag TV_CP_FID_OF_INPUT): x_in is complete and x0 is
            // in registers, so the fix-up no longer has to read x0 for the sites it completes -- the solver takes 1/2 |x_k - x0|^2
            // from sweep k (README.md:157 pairs it with the TV of x_{k-1}, which sweep k - 1 delivered).  TV_CP_FID_BOTH (round 5): the
            // fidelity of the OUTPUT over the complete sites goes to the second accumulator as well (the fix-up, called with x0, adds the rest)
            double f2 = 0.0;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const double e = (double)xv.v[i] - (double)x0v.v[i];
                f2 += 0.5 * e * e;
            }
            add1 = f2;
        }
        acc_fid += add1;
        acc_fid2 += add2;
    };

    // wait until the neighbouring waves have published `planes` planes (all waves of a block are resident and
    // a wave never waits for a plane its neighbour can only reach after this wave moves on: no deadlock)
    auto wait_neighbours = [&](int planes) {
        if (wave > 0) while (edge_flag[wave - 1] < planes) __builtin_amdgcn_s_sleep(1);
        if (wave < CP_NW - 1) while (edge_flag[wave + 1] < planes) __builtin_amdgcn_s_sleep(1);
        __threadfence_block();
    };
    for (int z = c.zs; z < c.ze; ++z) {
        if (XW && z > c.zs) wait_neighbours(z - c.zs);
        const int gz = g.z0 + z;
        const T* pc = zplane<T>(g, a.x_in, a.xp, a.xn, 1, z);
        const T* pn = zplane<T>(g, a.x_in, a.xp, a.xn, 1, z + 1);
        const bool has_pz = PREV && g.za && (gz > 0);
        const bool has_nz = NEXT && g.za && (pn != nullptr);
        const bool load_next = (pn != nullptr) && c.ok && (has_nz || (z + 1 < c.ze));
        // upwind / central: x(z+1) feeds the forward z difference of THIS plane, so a load issued inside the frame
        // would be one exposed latency per frame with only 1 + Nd = 5 streams in flight per wave -- request all M
        // frames of plane z+1 here instead (these two schemes have the registers for it; hybrid has 1 + 8 streams
        // per frame and no registers left; downwind needs x(z+1) only one plane later)
        constexpr bool PFN = (S == UPWIND || S == CENTRAL);
        VT Nn[PFN ? M : 1];
        if (PFN) {
#pragma unroll
            for (int t = 0; t < M; ++t) Nn[t] = (load_next && t0 + t < Mg) ? ldu_t<T, V>(pn + (long long)(t0 + t) * g.s_t, voff) : zero;
        }
        T En[(PFN && EA) ? M : 1];      // the border elements of plane z+1 travel with its wide loads
        if (PFN && EA) {
#pragma unroll
            for (int t = 0; t < M; ++t)
                En[t] = (load_next && (e_le || e_re) && t0 + t < Mg) ? ldu1_t<T>(pn + (long long)(t0 + t) * g.s_t, eoff) : T(0);
        }
        VT cold = zero;        // x(z, t-1)
        VT ut_prev = zero;     // wt * q'_tup(z, t-1) * mf, already valid-masked
        VT r_prev = zero;      // accumulator of frame t-1 of THIS plane, still missing its time-down term
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const int tg = t0 + t;                                                // frame of the volume
            if (TWIN && tg >= Mg) break;                                          // ragged last window (block-uniform)
            const long long toff = (long long)tg * g.s_t;                         // uniform
            VT N;
            if constexpr (PFN) N = Nn[t];
            else N = load_next ? ldu_t<T, V>(pn + toff, voff) : zero;
            T e_cur = T(0);
            if (EA) {
                e_cur = E[t];
                if constexpr (PFN) E[t] = En[t];
                else E[t] = (load_next && (e_le || e_re)) ? ldu1_t<T>(pn + toff, eoff) : T(0);
            }
            VT fx0 = zero, fx1 = zero;
            if constexpr (PFX) {
                const long long fo = (long long)(z - 1) * g.s_z + toff;
                const int fb = (int)(g.s_t * (long long)sizeof(T));
                const unsigned bo = c.ok ? voff : BUF_OOB;
                constexpr int NTX = TV_FUSED_NT ? BUF_NT : 0;
                if constexpr (ALG == ALG_CPOP) fx0 = buf_ld<T, V, NTX>(buf_rsrc<T>(a.p + fo, z > c.zs, fb), bo);
                else fx0 = buf_ld<T, V, NTX>(buf_rsrc<T>(a.x0 + fo, z > c.zs, fb), bo);
                if constexpr (ALG == ALG_CP) fx1 = buf_ld<T, V, NTX>(buf_rsrc<T>(a.p + fo, z > c.zs, fb), bo);
            }
            VT qcur[PFQ ? 4 : 1];
            if (PFQ) {
#pragma unroll
                for (int k = 0; k < 4; ++k) qcur[k] = qpre[k];
                const bool wrap = (t + 1 >= M) || (TWIN && t0 + t + 1 >= Mg);         // the window's last frame: on to the next plane
                const int tn = wrap ? 0 : t + 1, zn = wrap ? z + 1 : z;
                if (c.ok && zn < c.ze) {
                    const T* qbn = a.q_in + (long long)zn * g.s_dz + (long long)(t0 + tn) * g.s_t;
                    for_each_channel<S>(g, [&](auto slot, int ch) {
                        constexpr int k = decltype(slot)::value;
                        qpre[k & 3] = ldu_s_t<T, V>(qbn + (long long)ch * g.s_z, voff);
                    });
                }
            }
            // per-VOXEL weight on the time channels (tv_geom::time_weight_vol, the reference's to-do README.md:258): one more
            // streamed read per frame.  D scales the time channels of a voxel by ITS factor, and so does the adjoint below
            // (every q' sample is scaled by its own voxel's factor before the difference): round 3
            VT mft = mf;
            if (g.wv != nullptr && g.ta && c.ok) mft = mf * ldu_s_t<T, V>(static_cast<const T*>(g.wv) + (long long)z * g.s_z + toff, voff);
            // ------------------------------------------------ neighbourhood of x(z, t)
            XN<T, V> n;
            n.c = C[t];
            n.col0 = c.col0;
            n.nr = n.pr = n.nc = n.pc = n.nz = n.pz = n.nt = n.pt = zero;
            n.h_nr = n.h_pr = n.h_nz = n.h_pz = n.h_nt = n.h_pt = false;
            {   // halo rows of the wave tile: one predicated load (row 0 lanes read y-1, row 3 lanes y+1)
                const bool want_up = PREV && (c.row == 0) && c.ok && (c.y > 0);
                const bool want_dn = NEXT && (c.row == CP_TR - 1) && c.ok && (c.y + 1 < g.ny);
                VT halo = zero;
                if (want_up || want_dn) halo = ldu_t<T, V>(pc + toff, want_up ? voff - row_bytes : voff + row_bytes);
                if (NEXT) {
                    n.h_nr = c.ok && (c.y + 1 < g.ny);
                    const VT sdn = cp_shfl_down<T, V>(C[t]);
                    n.nr = (c.row == CP_TR - 1) ? halo : sdn;
                }
                if (PREV) {
                    n.h_pr = c.ok && (c.y > 0);
                    const VT sup = cp_shfl_up<T, V>(C[t]);
                    n.pr = (c.row == 0) ? halo : sup;
                }
            }
            {   // columns: adjacent lane inside the 16-lane row segment, one scalar at the segment edges
                const bool le = PREV && (c.lx == 0) && c.ok && (c.col0 > 0);
                const bool re = NEXT && (c.lx == CP_TL - 1) && c.ok && (c.col0 + V < g.nx);
                // inside the block the neighbouring wave published these elements (XE); the block's own edges are loaded
                const bool le_lds = XE && le && (wave > 0), re_lds = XE && re && (wave < CP_NW - 1);
                T edge = T(0);
                if (EA) edge = e_cur;
                else if ((le && !le_lds) || (re && !re_lds)) edge = ldu1_t<T>(pc + toff, le ? voff - (unsigned)sizeof(T) : voff + 16u);
                if (XE) {
                    const int xb = (z - c.zs) & 1;
                    if (le_lds) edge = edge_xr[xb][t][wave - 1][c.row];
                    if (re_lds) edge = edge_xl[xb][t][wave + 1][c.row];
                }
                if (NEXT) {
                    const T sh = __shfl_down(C[t].v[0], 1, 64);
                    n.nc = shift_left<T, V>(C[t], (c.lx == CP_TL - 1) ? edge : sh);
                }
                if (PREV) {
                    const T sh = __shfl_up(C[t].v[V - 1], 1, 64);
                    n.pc = shift_right<T, V>(C[t], (c.lx == 0) ? edge : sh);
                }
            }
            if (NEXT) {
                n.h_nz = has_nz; n.nz = N;
                if (t + 1 < M) { n.h_nt = (g.ta != 0) && (tg + 1 < Mg); n.nt = C[(t + 1 < M) ? t + 1 : t]; }
                else if (TWIN && g.ta && tg + 1 < Mg) { n.h_nt = true; n.nt = c.ok ? ldu_t<T, V>(pc + toff + g.s_t, voff) : zero; }   // across the seam
            }
            if (PREV) {
                n.h_pz = has_pz; n.pz = P[t];
                if (t > 0) { n.h_pt = (g.ta != 0); n.pt = cold; }
                else if (TWIN && g.ta && tg > 0) { n.h_pt = true; n.pt = c.ok ? ldu_t<T, V>(pc + toff - g.s_t, voff) : zero; }        // across the seam
            }
            VT o[8];
            d_slots<S, T, V>(g, w, n, mft, o);
            // ------------------------------------------------ dual update (README.md:149-151)
            VT v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = zero;
            T* qbase = a.q + (long long)z * g.s_dz + toff;                    // uniform
            const T* qbase_in = a.q_in + (long long)z * g.s_dz + toff;        // uniform (== qbase unless q is ping-ponged)
            VT vs = zero;
            if (c.ok) {
                for_each_channel<S>(g, [&](auto slot, int ch) {
                    constexpr int k = decltype(slot)::value;
                    const VT qv = PFQ ? qcur[k & 3] : ldu_s_t<T, V>(qbase_in + (long long)ch * g.s_z, voff);
                    v[k] = (ALG == ALG_ADMM) ? o[k] + qv : qv + a.sigma * o[k];
                    vs = vs + v[k] * v[k];
                });
                const VT ds = sumsq_slots<T, V>(o);
                VT scale;
                if constexpr (ALG == ALG_ADMM) {
                    // group soft threshold, the arithmetic of AdmmZU (tv_stencil.h): z = v scale, u = v - z, t = z - u
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        acc_tv += (double)tsqrt(ds.v[i]);
                        const T nv = tsqrt(vs.v[i]);
                        scale.v[i] = (nv > T(0)) ? tmax(T(0), T(1) - a.sigma / nv) : T(0);
                    }
                    // which samples of t' the fix-up will read (k_cp_fixup): the row channels on the rows next to a wave-tile
                    // seam, the column channels on the vectors next to a block-tile seam, the z channels on chunk-edge planes, the
                    // time channels on window-seam frames
                    const bool lastv = (c.lx == CP_TL - 1) && (!XW || wave == CP_NW - 1), firstv = (c.lx == 0) && (!XW || wave == 0);
                    const bool st_u[4] = {UP && c.row == CP_TR - 1, UP && lastv, UP && z == c.ze - 1, UP && TWIN && (t == M - 1 || tg + 1 >= Mg)};
                    const bool st_d[4] = {DN && c.row == 0, DN && firstv, DN && z == c.zs, DN && TWIN && t == 0};
                    T* tbase = a.p + (long long)z * g.s_dz + toff;
                    for_each_channel<S>(g, [&](auto slot, int ch) {
                        constexpr int k = decltype(slot)::value;
                        const VT zz = v[k] * scale;
                        const VT un = v[k] - zz;
                        stu_s_t<T, V>(qbase + (long long)ch * g.s_z, voff, un);
                        v[k] = (zz - un) - o[k];                                     // t': what the adjoint below is taken of
                        constexpr int axis = (S == HYBRID) ? ((k < 4) ? (k & 1) : (k >> 1)) : k;          // 0 rows, 1 cols, 2 z, 3 t
                        constexpr bool up_role = (S != HYBRID) || (k == 0 || k == 1 || k == 4 || k == 6);
                        constexpr bool dn_role = (S != HYBRID) || !up_role;
                        if ((a.full_store & 1) || (up_role && st_u[axis]) || (dn_role && st_d[axis]))
#if TV_ADMM_T_NT
                            stu_s_t<T, V>(tbase + (long long)ch * g.s_z, voff, v[k]);
#else
                            stu_t<T, V>(tbase + (long long)ch * g.s_z, voff, v[k]);
#endif
                    });
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        acc_tv += (double)tsqrt(ds.v[i]);
                        scale.v[i] = T(1) / tmax(T(1), tsqrt(vs.v[i]) * a.inv_lambda);
                    }
                    for_each_channel<S>(g, [&](auto slot, int ch) {
                        constexpr int k = decltype(slot)::value;
                        v[k] = v[k] * scale;
                        stu_s_t<T, V>(qbase + (long long)ch * g.s_z, voff, v[k]);
                    });
                }
            }
            // slots: non-hybrid 0 rows, 1 cols, 2 z, 3 t ; hybrid 0 ru, 1 cu, 2 rd, 3 cd, 4 zu, 5 zd, 6 tu, 7 td
            constexpr int k_ru = 0, k_cu = 1, k_rd = (S == HYBRID) ? 2 : 0, k_cd = (S == HYBRID) ? 3 : 1;
            constexpr int k_zu = (S == HYBRID) ? 4 : 2, k_zd = (S == HYBRID) ? 5 : 2;
            constexpr int k_tu = (S == HYBRID) ? 6 : 3, k_td = (S == HYBRID) ? 7 : 3;
            // y^ validity: an up channel is defined where the site has a next neighbour, a down
            // channel where it has a previous one; everything else counts as zero
            VT qru = zero, qrd = zero, qcu = zero, qcd = zero, qzu = zero, qzd = zero, qtu = zero, qtd = zero;
            if (UP) {
                if (c.ok && c.y + 1 < g.ny && (!CEN || c.y > 0)) qru = v[k_ru];
#pragma unroll
                for (int i = 0; i < V; ++i)
                    qcu.v[i] = (c.ok && c.col0 + i < g.nx - 1 && (!CEN || c.col0 + i > 0)) ? v[k_cu].v[i] : T(0);
                if (g.za && c.ok && gz + 1 < g.nzg && (!CEN || z_fwd || gz > 0)) qzu = w.wz * v[k_zu];
                if (g.ta && c.ok && tg + 1 < Mg && (!CEN || t_fwd || tg > 0)) qtu = (w.wt * v[k_tu]) * mft;
            }
            if (DN) {
                if (c.ok && c.y > 0 && (!CEN || c.y + 1 < g.ny)) qrd = v[k_rd];
#pragma unroll
                for (int i = 0; i < V; ++i)
                    qcd.v[i] = (c.ok && c.col0 + i > 0 && (!CEN || c.col0 + i < g.nx - 1)) ? v[k_cd].v[i] : T(0);
                if (g.za && c.ok && gz > 0 && (!CEN || (!z_fwd && gz + 1 < g.nzg))) qzd = w.wz * v[k_zd];
                if (g.ta && c.ok && tg > 0 && (!CEN || (!t_fwd && tg + 1 < Mg))) qtd = (w.wt * v[k_td]) * mft;
            }
            // ------------------------------------------------ lagged primal update of plane z-1
            if (z > c.zs) {
                VT rf = lds_R[t][tid];
                if (DN) rf = rf - qzd;
                finalize(z - 1, t, P[t], rf, PFX, fx0, fx1);
            }
            // ------------------------------------------------ adjoint accumulator of plane z
            VT r = zero;
            if (UP) {
                if (!CEN) r = r - qru - qcu - qzu - qtu;              // own-site terms (central has none ...
                else {
                    if (z_fwd) r = r - qzu;                           // ... except on its two-point axes)
                    if (t_fwd) r = r - qtu;
                }
                const VT above = cp_shfl_up<T, V>(qru);                      // q'_rowup of the row above
                if (c.row > 0) r = r + above;
                const T lft = __shfl_up(qcu.v[V - 1], 1, 64);         // q'_colup one column to the left
                r = r + shift_right<T, V>(qcu, (c.lx == 0) ? T(0) : lft);
                if (g.za) {
                    r = r + lds_U[t][tid];                            // wz q'_zup(z-1, t)
                    lds_U[t][tid] = qzu;
                }
                r = r + ut_prev;                                      // wt q'_tup(z, t-1)
                ut_prev = qtu;
            }
            if (DN) {
                if (!CEN) r = r + qrd + qcd + qzd + qtd;
                const VT below = cp_shfl_down<T, V>(qrd);                    // q'_rowdown of the row below
                if (c.row < CP_TR - 1) r = r - below;
                const T rgt = __shfl_down(qcd.v[0], 1, 64);       // q'_coldown one column to the right
                r = r - shift_left<T, V>(qcd, (c.lx == CP_TL - 1) ? T(0) : rgt);
                r_prev = r_prev - qtd;                                // time-down term of frame t-1
            }
            if (XW) {
                const int eb = (z - c.zs) & 1;
                if (UP && c.lx == CP_TL - 1) edge_cu[eb][t][wave][c.row] = qcu.v[V - 1];
                if (DN && c.lx == 0) edge_cd[eb][t][wave][c.row] = qcd.v[0];
            }
            if (t > 0) lds_R[(t > 0) ? t - 1 : 0][tid] = r_prev;     // frame t-1 is complete up to its z+1 term
            r_prev = r;
            if (t == M - 1 || (TWIN && tg + 1 >= Mg)) lds_R[t][tid] = r;
            cold = C[t];
            P[t] = C[t];
            C[t] = N;
            if (XE) {                 // x edges of the plane the block works on next
                const int nb = (z + 1 - c.zs) & 1;
                if (c.lx == 0) edge_xl[nb][t][wave][c.row] = N.v[0];
                if (c.lx == CP_TL - 1) edge_xr[nb][t][wave][c.row] = N.v[V - 1];
            }
        }
        if (XW) {                     // publish: edge columns of plane z are visible before the counter moves
            __threadfence_block();
            if (c.lane == 0) edge_flag[wave] = z - c.zs + 1;
        }
    }
    // last plane of the chunk: its z+1 term (if any) is the fix-up kernel's
    if (XW) wait_neighbours(c.ze - c.zs);
#pragma unroll
    for (int t = 0; t < M; ++t) finalize(c.ze - 1, t, P[t], lds_R[t][tid]);

    acc_tv = block_sum(acc_tv, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part_tv[linear_block_id()] = acc_tv;
    acc_fid = block_sum(acc_fid, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) a.part_fid[linear_block_id()] = acc_fid;
    if constexpr (ALG == ALG_CP) {
        if (a.full_store & 4) {               // uniform
            acc_fid2 = block_sum(acc_fid2, sm);
            if (threadIdx.x == 0 && threadIdx.y == 0) a.part_fid2[linear_block_id()] = acc_fid2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fix-up: add the adjoint terms the sweep could not see (other wave tiles, other z-chunks, other
// ranks) to x_out and account the fidelity of those sites.  One site-vector per thread; vectors
// with nothing missing return at once.
// ---------------------------------------------------------------------------------------------
template <typename T> struct FixupArgsT {
    const T* q;
    const T* qp;          // plane z0-1 of the z-up channel (previous rank) or nullptr
    const T* qn;          // plane z0+nz of the z-down channel (next rank) or nullptr
    T* x_out;
    const T* x0;
    T tau;
    int chunk0;           // class 1: first chunk whose edge planes are visited
};
using FixupArgs = FixupArgsT<float>;

// all missing terms of one site-vector; returns its fidelity 1/2 |x_out - x0|^2 (0 if nothing was missing)
#ifndef TV_FIXUP_NT
#define TV_FIXUP_NT 1            // the vector streams of the fix-up (q' rows, x_out read-modify-write, x0) non-temporal: 2.92 -> 2.73 ms on the north star (0: plain)
#endif
template <typename T, int V> __device__ __forceinline__ Vec<T, V> FXLD(const T* p) {
#if TV_FIXUP_NT
    return vload_s<T, V>(p);
#else
    return vload<T, V>(p);
#endif
}
template <typename T, int V> __device__ __forceinline__ void FXST(T* p, const Vec<T, V>& a) {
#if TV_FIXUP_NT
    vstore_s<T, V>(p, a);
#else
    vstore<T, V>(p, a);
#endif
}
// ALG_ADMM: `q` is the array of t', `x_out` the residual r, tau = -rho (r += rho s m), the returned partial is r^2 (no x0)
template <int S, bool XW, typename T = float, int ALG = ALG_CP>
__device__ __forceinline__ double fixup_site(const DG& g, const WT<T>& w, const FixupArgsT<T>& a, int zchunk, int zl, int t, int y,
                                             int col0) {
    constexpr int V = 16 / (int)sizeof(T);
    using VT = Vec<T, V>;
    constexpr bool UP = (S != DOWNWIND), DN = (S != UPWIND), CEN = (S == CENTRAL);
    if (!fused_needs_fixup<S, XW, V>(g, zl, y, col0, zchunk, t)) return 0.0;
    constexpr int CM = XW ? CPG<V>::BC - 1 : CPG<V>::WC - 1;
    const int c_ru = 0, c_cu = 1, c_rd = (S == HYBRID) ? 2 : 0, c_cd = (S == HYBRID) ? 3 : 1;
    const int c_zu = g.ch_z, c_zd = (S == HYBRID) ? g.ch_z + 1 : g.ch_z;
    const bool z_fwd = CEN && g.z_two;
    const long long inpl = (long long)t * g.s_t + (long long)y * g.rp + col0;
    const T* qb = a.q + (long long)zl * g.s_dz + inpl;
    const VT zero = vsplat<T, V>(T(0));
    VT m = zero;
    // a missing term counts only where the neighbour's channel is defined (central: interior points of the axis)
    if (UP && (y & (CP_TR - 1)) == 0 && y >= (CEN ? 2 : 1)) m = m + FXLD<T, V>(qb + (long long)c_ru * g.s_z - g.rp);
    if (DN && (y & (CP_TR - 1)) == CP_TR - 1 && y <= g.ny - (CEN ? 3 : 2)) m = m - FXLD<T, V>(qb + (long long)c_rd * g.s_z + g.rp);
    if (UP && (col0 & CM) == 0 && col0 >= (CEN ? 2 : 1)) m.v[0] += qb[(long long)c_cu * g.s_z - 1];
    if (DN && (col0 & CM) == CM - (V - 1) && col0 + V <= g.nx - (CEN ? 2 : 1)) m.v[V - 1] -= qb[(long long)c_cd * g.s_z + V];
    if (g.za) {
        const int gz = g.z0 + zl;
        const int zs = (zl / zchunk) * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        if (UP && zl == zs && gz >= ((CEN && !z_fwd) ? 2 : 1)) {
            const VT u = (zl >= 1) ? FXLD<T, V>(qb + (long long)c_zu * g.s_z - g.s_dz) : FXLD<T, V>(a.qp + inpl);
            m = m + w.wz * u;
        }
        if (DN && !z_fwd && zl == ze - 1 && gz <= g.nzg - (CEN ? 3 : 2)) {
            const VT d = (zl + 1 < g.nz) ? FXLD<T, V>(qb + (long long)c_zd * g.s_z + g.s_dz) : FXLD<T, V>(a.qn + inpl);
            m = m - w.wz * d;
        }
    }
    if (g.ta && g.m > CP_TWN) {      // time-window seams: the neighbouring frame belongs to another window of the sweep
        const int c_tu = g.ch_t, c_td = (S == HYBRID) ? g.ch_t + 1 : g.ch_t;
        const int k = t % CP_TWN;
        VT mt = zero;       // every sample with the per-voxel factor of ITS frame (1 without a weight volume)
        if (UP && k == 0 && t >= 1) mt = mt + FXLD<T, V>(qb + (long long)c_tu * g.s_z - g.s_t) * vol_factor<T, V>(g, zl, t - 1, y, col0);
        if (DN && k == CP_TWN - 1 && t <= g.m - (CEN ? 3 : 2))
            mt = mt - FXLD<T, V>(qb + (long long)c_td * g.s_z + g.s_t) * vol_factor<T, V>(g, zl, t + 1, y, col0);
        m = m + (w.wt * mt) * mask_factor<T, V>(g, w.sf, y, col0);
    }
    const T s = (S == HYBRID) ? Consts<T>::inv_sqrt2() : (CEN ? T(0.5) : T(1));
    const long long off = (long long)zl * g.s_z + inpl;
    const VT xv = FXLD<T, V>(a.x_out + off);
    VT x0v = zero;
    if constexpr (ALG == ALG_CP) { if (a.x0 != nullptr) x0v = FXLD<T, V>(a.x0 + off); }      // x0 NULL: the caller takes the fidelity elsewhere
    VT xo;
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        xo.v[i] = xv.v[i] - a.tau * (s * m.v[i]);
        const double e = (double)xo.v[i] - (double)x0v.v[i];
        acc += (ALG == ALG_CP ? 0.5 : 1.0) * e * e;
    }
    FXST<T, V>(a.x_out + off, xo);
    return acc;
}

// is row y one of the rows whose every vector misses a row term ("fix-up rows")?
template <int S> __device__ __forceinline__ bool is_fix_row(const DG& g, int y) {
    constexpr bool UP = (S != DOWNWIND), DN = (S != UPWIND);
    return (UP && (y & (CP_TR - 1)) == 0 && y >= 1) || (DN && (y & (CP_TR - 1)) == CP_TR - 1 && y <= g.ny - 2);
}
// is local plane zl a z-chunk edge plane with a missing z term?
template <int S> __device__ __forceinline__ bool is_fix_plane(const DG& g, int zl, int zchunk) {
    constexpr bool UP = (S != DOWNWIND), DN = (S != UPWIND);
    if (!g.za) return false;
    const int gz = g.z0 + zl;
    const int zs = (zl / zchunk) * zchunk;
    const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
    return (UP && zl == zs && gz >= 1) || (DN && zl == ze - 1 && gz <= g.nzg - 2);
}

// The fix-up work is split into three densely mapped classes so that no wave idles:
//   CLS 0  fix-up rows (every plane): each wave = 256 columns of ONE fix-up row; all terms of those sites
//   CLS 1  chunk-edge planes, the remaining rows: generic (tiles, m, plane-list) mapping
//   CLS 2  the sparse column-edge vectors of the remaining rows on the remaining planes: one per thread
// grid: CLS 0 (tiles_x * row groups, m, nz); CLS 1 (tiles_x * tiles_y, m, 2 * nchunks); CLS 2 (ceil(cands/256), m, nz)
template <int S, int CLS, bool XW, typename T = float, int ALG = ALG_CP>
__global__ __launch_bounds__(256) void k_cp_fixup(DG g, WT<T> w, FixupArgsT<T> a, int zchunk, int zb, int zn, double* partials) {
    constexpr int V = 16 / (int)sizeof(T);
    // (central row groups are laid out like hybrid ones: both the top and the bottom row of every wave tile)
    // the call covers local planes [zb, zb + zn): classes 0 and 2 have grid z = zn; class 1 has two grid-z
    // slots (first / last plane) per z-chunk intersecting the range, starting at chunk a.chunk0
    __shared__ double sm[16];
    const int nxv = (g.nx + V - 1) / V;
    const int tiles_x = (nxv + 63) / 64;
    double acc = 0.0;
    if (CLS == 0) {
        const int bx = (int)blockIdx.x % tiles_x, grp = (int)blockIdx.x / tiles_x;
        const int ty = (int)threadIdx.y;
        int y;
        // hybrid / central: first and last row of two consecutive wave tiles; up / down: one row of four tiles
        if (S == HYBRID || S == CENTRAL) y = grp * 2 * CP_TR + ((ty & 1) ? CP_TR - 1 : 0) + ((ty & 2) ? CP_TR : 0);
        else y = grp * 4 * CP_TR + CP_TR * ty + (S == DOWNWIND ? CP_TR - 1 : 0);
        const int col0 = (bx * 64 + (int)threadIdx.x) * V;
        if (col0 < g.nx && y < g.ny && is_fix_row<S>(g, y) && !is_seam_frame<S>(g, (int)blockIdx.y))
            acc = fixup_site<S, XW, T, ALG>(g, w, a, zchunk, zb + (int)blockIdx.z, (int)blockIdx.y, y, col0);
    } else if (CLS == 1) {
        const int k = (int)blockIdx.z, chunk = a.chunk0 + (k >> 1);
        const int zs = chunk * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        const int zl = (k & 1) ? ze - 1 : zs;
        const bool dup = (k & 1) && (ze - 1 == zs);                      // one-plane chunk: handled as its "zs"
        const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
        const int y = by * 4 + (int)threadIdx.y;
        const int col0 = (bx * 64 + (int)threadIdx.x) * V;
        if (!dup && zs < g.nz && zl >= zb && zl < zb + zn && col0 < g.nx && y < g.ny && is_fix_plane<S>(g, zl, zchunk) &&
            !is_fix_row<S>(g, y) && !is_seam_frame<S>(g, (int)blockIdx.y))
            acc = fixup_site<S, XW, T, ALG>(g, w, a, zchunk, zl, (int)blockIdx.y, y, col0);
    } else if (CLS == 3) {
        // time-window seam frames (M > CP_TWN): EVERY site of such a frame misses a time term; blockIdx.y counts
        // the seam frames: windows' first frames (up) and last frames (down), as 2 slots per window
        const int wdw = (int)blockIdx.y >> 1, t = wdw * CP_TWN + (((int)blockIdx.y & 1) ? CP_TWN - 1 : 0);
        const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
        const int y = by * 4 + (int)threadIdx.y;
        const int col0 = (bx * 64 + (int)threadIdx.x) * V;
        if (t < g.m && col0 < g.nx && y < g.ny && is_seam_frame<S>(g, t))
            acc = fixup_site<S, XW, T, ALG>(g, w, a, zchunk, zb + (int)blockIdx.z, t, y, col0);
    } else {
        // candidates per row: the first and the last vector of every block tile (XW) / wave tile
        constexpr int TW = XW ? CPG<V>::BC : CPG<V>::WC;
        const int ntile = (g.nx + TW - 1) / TW, ncand = 2 * ntile;
        const long long idx = (long long)blockIdx.x * 256 + (int)threadIdx.y * 64 + (int)threadIdx.x;
        const int zl = zb + (int)blockIdx.z;
        if (idx < (long long)g.ny * ncand && !is_fix_plane<S>(g, zl, zchunk) && !is_seam_frame<S>(g, (int)blockIdx.y)) {
            const int y = (int)(idx / ncand), cnd = (int)(idx % ncand);
            const int col0 = (cnd >> 1) * TW + ((cnd & 1) ? TW - V : 0);
            if (col0 < g.nx && !is_fix_row<S>(g, y)) acc = fixup_site<S, XW, T, ALG>(g, w, a, zchunk, zl, (int)blockIdx.y, y, col0);
        }
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[linear_block_id()] = acc;
}


// =============================================================================================
// sub-gradient pass 2, marching (radius-1 schemes, fp32): G from x and 1/|Dx| (norms_ext, one ghost plane in
// front), both fields fetched once: x planes z-1, z in registers, |Dx| planes z-1, z in per-thread LDS slots,
// row / column neighbours of both by lane shuffles inside the 4-row x 64-col wave tile (halo rows / edge
// columns: predicated loads).  No fix-up needed: this is a pure gather.
// =============================================================================================
template <int S, int M>
__global__ __launch_bounds__(256, 2) void k_subgrad_march(DG g, WT<float> w, const float* __restrict__ x, const float* __restrict__ xp,
                                                          const float* __restrict__ xn, const float* __restrict__ norms_ext,
                                                          float* __restrict__ G, int zchunk) {
    static_assert(S != CENTRAL, "radius-2 scheme");
    constexpr bool UP = (S == UPWIND || S == HYBRID), DN = (S == DOWNWIND || S == HYBRID);
    const FusedCoord c = fused_coord(g, zchunk, 0);
    const F4 zero = vsplat<float, 4>(0.f), one = vsplat<float, 4>(1.f);
    const F4 mf = g.ta ? mask_factor<float, 4>(g, w.sf, c.ok ? c.y : 0, c.ok ? c.col0 : 0) : one;
    __shared__ F4 lds_WC[M][256];
    __shared__ F4 lds_WP[M][256];
    const int tid = (int)threadIdx.y * 64 + (int)threadIdx.x;
    auto nplane = [&](int zl) -> const float* {          // |Dx| plane of LOCAL index zl (ghosts at -1 and nz), or nullptr
        const int gz = g.z0 + zl;
        return (gz >= 0 && gz < g.nzg) ? norms_ext + (long long)(zl + 1) * g.s_z : nullptr;
    };
    F4 XC[M], XP[M];
    {
        const float* pc = zplane<float>(g, x, xp, xn, 2, c.zs);
        const float* pp = g.za ? zplane<float>(g, x, xp, xn, 2, c.zs - 1) : nullptr;
        const float* qc = nplane(c.zs);
        const float* qp = g.za ? nplane(c.zs - 1) : nullptr;
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const long long off = (long long)t * g.s_t + c.inpl;
            XC[t] = c.ok ? vload<float, 4>(pc + off) : zero;
            XP[t] = (c.ok && pp) ? vload<float, 4>(pp + off) : zero;
            lds_WC[t][tid] = c.ok ? vload<float, 4>(qc + off) : one;
            lds_WP[t][tid] = (c.ok && qp && UP) ? vload<float, 4>(qp + off) : one;
        }
    }
    // neighbourhood of a field whose plane-z value of this frame is `cv`
    auto neighbourhood = [&](XN<float, 4>& n, const F4& cv, const float* plane_c, long long off, const F4& nzv, const F4& pzv,
                             const F4& ntv, const F4& ptv, bool h_nz, bool h_pz, bool h_nt, bool h_pt) {
        n.c = cv;
        n.col0 = c.col0;
        const bool want_up = (c.row == 0) && c.ok && (c.y > 0);
        const bool want_dn = (c.row == 3) && c.ok && (c.y + 1 < g.ny);
        F4 halo = zero;
        if (want_up || want_dn) halo = vload<float, 4>(plane_c + off + (want_up ? -(long long)g.rp : (long long)g.rp));
        n.h_nr = c.ok && (c.y + 1 < g.ny);
        n.h_pr = c.ok && (c.y > 0);
        const F4 sdn = shfl_down16(cv), sup = shfl_up16(cv);
        n.nr = (c.row == 3) ? halo : sdn;
        n.pr = (c.row == 0) ? halo : sup;
        const bool le = (c.lx == 0) && c.ok && (c.col0 > 0), re = (c.lx == 15) && c.ok && (c.col0 + 4 < g.nx);
        float edge = 0.f;
        if (le || re) edge = le ? plane_c[off - 1] : plane_c[off + 4];
        const float shr = __shfl_down(cv.v[0], 1, 64), shl = __shfl_up(cv.v[3], 1, 64);
        n.nc = shift_left<float, 4>(cv, (c.lx == 15) ? edge : shr);
        n.pc = shift_right<float, 4>(cv, (c.lx == 0) ? edge : shl);
        n.nz = nzv; n.pz = pzv; n.nt = ntv; n.pt = ptv;
        n.h_nz = h_nz; n.h_pz = h_pz; n.h_nt = h_nt; n.h_pt = h_pt;
    };
    for (int z = c.zs; z < c.ze; ++z) {
        const int gz = g.z0 + z;
        const float* pc = zplane<float>(g, x, xp, xn, 2, z);
        const float* pn = zplane<float>(g, x, xp, xn, 2, z + 1);
        const float* qc = nplane(z);
        const float* qn = nplane(z + 1);
        const bool has_pz = g.za && (gz > 0), has_nz = g.za && (pn != nullptr);
        const bool more = (z + 1 < c.ze);
        F4 xcold = zero, wcold = one;
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const long long off = (long long)t * g.s_t + c.inpl;
            const F4 XNv = (pn != nullptr && c.ok && (has_nz || more)) ? vload<float, 4>(pn + off) : zero;
            const F4 WNv = (qn != nullptr && c.ok && ((has_nz && DN) || more)) ? vload<float, 4>(qn + off) : one;
            const F4 wc = lds_WC[t][tid], wp = lds_WP[t][tid];
            const bool h_nt = g.ta && (t + 1 < M), h_pt = g.ta && (t > 0);
            XN<float, 4> xs, ns;
            neighbourhood(xs, XC[t], pc, off, XNv, XP[t], XC[(t + 1 < M) ? t + 1 : t], xcold, has_nz, has_pz, h_nt, h_pt);
            neighbourhood(ns, wc, qc, off, WNv, wp, lds_WC[(t + 1 < M) ? t + 1 : t][tid], wcold, has_nz, has_pz, h_nt, h_pt);
            if (c.ok) {
                const F4 r = subgrad_site<S, float, 4>(g, w, xs, ns, mf);
                vstore<float, 4>(G + (long long)z * g.s_z + off, r);
            }
            xcold = XC[t];
            wcold = wc;
            XP[t] = XC[t];
            XC[t] = XNv;
            lds_WP[t][tid] = wc;
            lds_WC[t][tid] = WNv;
        }
    }
}

}  // namespace tv
