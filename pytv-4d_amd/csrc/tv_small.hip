// tv_small.hip -- PERSISTENT kernels for small volumes (round 6): K iterations of Chambolle-Pock / of the sub-gradient descent in ONE launch.
//
// The reference's own shapes (README.md:76-79 rand(20,4,100,100); README.md:107-124, 141-157: 300 iterations on a 256 x 256 / 512 x 512
// image; pytv/tests.py:48 N = 100, Nz = 20) hold 0.07 - 1 Mvoxel: an iteration moves a few MB that never leave the caches, and the kernel
// pair + reductions the ordinary path launches per iteration cost more in launch gaps than in work (38 - 40 us per iteration at 0.8 Mvoxel,
// profiles/r5_small_frames.txt).  Here the whole loop runs inside one cooperative launch:
//
//   * the sites are cut into VIRTUAL blocks exactly like the one-site kernels cut them (256 threads = bx lanes of 16 bytes x by rows; tile,
//     frame, plane), and every launched block owns a CONTIGUOUS range of them for all iterations -- a thread revisits the same sites, its
//     private arrays (x0, p) stay in its XCD's L2 with ordinary accesses;
//   * an iteration has two phases (CP: dual update, primal update; descent: 1/|Dx|, gather + step), each the per-site body of the ordinary
//     kernels (tv_site.h), and after each phase a block waits ONLY for the blocks that own neighbouring sites: every block publishes a phase
//     counter in its own 128-byte line (one agent-coherent store) and polls the <= 24 lines of its neighbours -- ~1 us against 2.7 - 3.7 us
//     for a grid-wide barrier on 256 blocks (one hot counter serialises at ~14 ns per arrival: profiles/r6_barrier_bench.txt);
//   * MI355X has eight XCDs with eight L2s that are not coherent with each other inside a launch; a bulk agent-scope fence per wave costs
//     60 - 100 us per barrier (same file, variant C).  Arrays that neighbours read (x, q, 1/|Dx|) are therefore accessed with
//     agent-coherent (sc1) raw-buffer loads / stores only (tv_stencil.h: CohMem; profiles/r6_coherence_probe.txt: never stale, repeated
//     loads of a line hit the L2, data of the own XCD at ~8 TB/s, first touch of another XCD's data at ~3 TB/s), and the launch order is
//     XCD-aware: consecutive ranges sit on the same XCD, so most neighbours share an L2;
//   * in-place updates are safe: a phase's wait covers both "my neighbour's data is ready" and "my neighbour has finished reading what I am
//     about to overwrite" because the dependency relation is symmetric (the descent step ping-pongs x: its gather reads x at the sites it
//     would overwrite);
//   * TV / fidelity of every iteration: fp64 per-block partials [iteration][which][block], summed at the end by one deterministic pass.
//
// Arithmetic per site is that of tv_cp_dual / tv_cp_primal / tv_subgrad + tv_subgrad_step (IEEE sqrt / divide): results equal the ordinary
// small-volume path to rounding (the TV / fidelity sums group the sites differently).
#include <hip/hip_runtime.h>

#include "tv_host.h"
#include "tv_site.h"
#include "tv_stencil.h"

namespace tv {

// ---------------------------------------------------------------------------------------------------------------------------------
struct SmallPlan {
    int bx, by;             // a virtual block: bx lanes x by rows, bx * by == 256
    int tiles_x, tiles_y;   // virtual blocks per frame
    int T;                  // tiles_x * tiles_y
    int nvb;                // T * m * nz
    int per_block;          // virtual blocks per launched block (a contiguous range)
    int nblocks;            // launched blocks that own sites
    int grid;               // launched blocks: nblocks rounded up to a multiple of 8 (XCD-aware order)
};
constexpr int kFlagStride = 32;          // unsigned words per flag: one 128-byte line per block
constexpr int kSmallThreads = 256;
constexpr int kMaxSmallBlocks = 8192;
// The phase counters are never reset: a launch starts from the epoch its predecessor on the same workspace left in the line behind the
// flags (bumped by 2 n_iter by the reduction kernel that closes every call) -- no memset per launch.  The workspace is zero-filled ONCE by
// its owner before the first call.  Comparisons are wrap-safe.
__device__ __forceinline__ unsigned small_epoch_base(const unsigned* flags) { return flags[(long long)kMaxSmallBlocks * kFlagStride]; }

// blockIdx.x -> logical block id such that ids [k * grid/8, (k+1) * grid/8) run on XCD k (consecutive blockIdx values go round-robin
// over the eight XCDs; nothing but speed depends on that assumption -- every shared access is agent-coherent)
__device__ __forceinline__ int small_logical_id(const SmallPlan& sp) {
    const int b = (int)blockIdx.x, per = sp.grid >> 3;
    return (b & 7) * per + (b >> 3);
}

template <int V> __device__ __forceinline__ Coord small_coord(const DG& g, const SmallPlan& sp, int vb) {
    const int tile = vb % sp.T, rest = vb / sp.T;
    const int bxi = tile % sp.tiles_x, byi = tile / sp.tiles_x;
    const int tid = (int)threadIdx.x;
    const int tx = tid % sp.bx, ty = tid / sp.bx;
    const int nxv = (g.nx + V - 1) / V;
    Coord c;
    const int jv = bxi * sp.bx + tx;
    c.col0 = jv * V;
    c.y = byi * sp.by + ty;
    c.t = rest % g.m;
    c.zl = rest / g.m;
    c.ok = (jv < nxv) && (c.y < g.ny);
    return c;
}

// which launched block must lane `i` (0 .. 23) of the first wave wait for?  -1: nobody.
// Neighbouring sites lie one tile along x / y, one or two frames, one or two planes away (two: the central sub-gradient's radius-2 stencil):
// virtual-block offsets +-1, +-tiles_x, +-T, +-2T, +-T m, +-2 T m; a range of per_block virtual blocks shifted by an offset is owned by at most
// two blocks.  The relation is symmetric (every offset comes with its negative): what the in-place updates rely on.
__device__ __forceinline__ int small_dependency(const DG& g, const SmallPlan& sp, int L, int i) {
    if (i >= 24) return -1;
    const int which = i >> 2, neg = (i >> 1) & 1, last = i & 1;
    long long d = 0;
    switch (which) {
        case 0: d = 1; break;
        case 1: d = sp.tiles_x; break;
        case 2: d = sp.T; break;
        case 3: d = 2ll * sp.T; break;
        case 4: d = (long long)sp.T * g.m; break;
        default: d = 2ll * sp.T * g.m; break;
    }
    if (neg) d = -d;
    long long lo = (long long)L * sp.per_block, hi = lo + sp.per_block - 1;
    if (hi > sp.nvb - 1) hi = sp.nvb - 1;
    lo += d; hi += d;
    if (hi < 0 || lo > sp.nvb - 1) return -1;
    if (lo < 0) lo = 0;
    if (hi > sp.nvb - 1) hi = sp.nvb - 1;
    const int owner = (int)((last ? hi : lo) / sp.per_block);
    return owner == L ? -1 : owner;
}

// end of a phase: my stores are performed, my flag says so, my neighbours' flags say the same.
// SAFETY NET: the blocks of a launch wait for each other, so all of them must be resident at once.  The launch is cooperative and the
// grid is kept to half the wave slots of the chip (small_capacity: hipLaunchCooperativeKernel accepted 7 blocks of 256 threads per CU that
// the hardware then did not keep resident together -- a hang, round 6); should a wait still not end within ~2 s of polling, the block
// raises the ABORT word of the workspace, every block that sees it leaves its iteration loop, the closing reduction turns the whole
// history into NaN and the host fails loudly instead of hanging the GPU.  Returns true when the launch is being abandoned.
constexpr unsigned kSmallSpinLimit = 1u << 21;
__device__ __forceinline__ unsigned* small_abort_word(unsigned* flags) { return flags + (long long)kMaxSmallBlocks * kFlagStride + 1; }
__device__ __forceinline__ bool small_sync(unsigned* flags, int L, int dep, unsigned epoch, int* sh_abort) {
    __builtin_amdgcn_s_waitcnt(0);              // vmcnt(0): this wave's (write-through) stores have been acknowledged
    __syncthreads();
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + (long long)L * kFlagStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dep >= 0) {
            const unsigned* f = flags + (long long)dep * kFlagStride;
            unsigned spins = 0;
            while ((int)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0) {      // (wrap-safe)
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 && (spins >= kSmallSpinLimit || __hip_atomic_load(small_abort_word(flags), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_store(small_abort_word(flags), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *sh_abort = 1;
                    break;
                }
            }
        }
    }
    __syncthreads();
    return *sh_abort != 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// epilogues with agent-coherent accesses (the arithmetic of tv_stencil.h's CpDual / NormEpi / CpPrimal)
template <int S, typename T, int V> struct CpDualCoh {
    static constexpr bool REDUCES = true;
    T* q;
    T sigma, inv_lambda;
    CohMem mq;
    __device__ __forceinline__ double operator()(const DG& g, const Coord& c, const Vec<T, V> (&o)[8]) const {
        T* base = q + (long long)c.zl * g.s_dz + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
        Vec<T, V> v[8];
        Vec<T, V> vs = vsplat<T, V>(T(0));
        for_each_channel<S>(g, [&](auto slot, int ch) {
            constexpr int k = decltype(slot)::value;
            v[k] = mq.template ld<T, V>(base + (long long)ch * g.s_z) + sigma * o[k];
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
            mq.template st<T, V>(base + (long long)ch * g.s_z, v[k] * scale);
        });
        return acc;
    }
};

template <int S, typename T, int V> struct NormEpiCoh {
    static constexpr bool REDUCES = true;
    T* norms_ext;
    CohMem mn;
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
        mn.template st<T, V>(norms_ext + (long long)(c.zl + 1) * g.s_z + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0, n);
        return acc;
    }
};

template <typename T, int V> struct SrcCoh {          // the dual variable as the gather of D^T reads it; no halo planes (unsharded volume)
    const T* y;
    CohMem m;
    __device__ __forceinline__ Vec<T, V> ld(long long off) const { return m.template ld<T, V>(y + off); }
    __device__ __forceinline__ T lds(long long off) const { return m.template ld1<T>(y + off); }
    __device__ __forceinline__ Vec<T, V> ldp(long long) const { return vsplat<T, V>(T(0)); }
    __device__ __forceinline__ Vec<T, V> ldn(long long) const { return vsplat<T, V>(T(0)); }
};

// -DTV_SMALL_PROFILE (variant builds): thread 0 of every block records the 100 MHz wall clock at five points of every CP iteration
// (start | dual done | neighbours ready | primal done | neighbours ready) behind the partials: tools/small_volume_profile.py reads them
#ifdef TV_SMALL_PROFILE
#define TV_SMALL_MARK(k) do { if (threadIdx.x == 0) a.partials[(long long)a.n_iter * 2 * sp.nblocks + ((long long)it * sp.nblocks + L) * 5 + (k)] = (double)wall_clock64(); } while (0)
#else
#define TV_SMALL_MARK(k) do { } while (0)
#endif

template <typename T> struct SmallCpArgs {
    T* x;
    const T* x0;
    T* p;
    T* q;
    T sigma_D, inv_lambda, tau, sigma_A, inv_1p_sigma_A;
    int n_iter;
    unsigned* flags;
    double* partials;        // [n_iter][2][nblocks]
    long long x_bytes, q_bytes;
};

template <int S, typename T, int V>
__global__ __launch_bounds__(kSmallThreads) void k_small_cp(DG g, WT<T> w, SmallPlan sp, SmallCpArgs<T> a) {
    __shared__ double sm[16];
    __shared__ int sh_abort;
    if (threadIdx.x == 0) sh_abort = 0;
    const int L = small_logical_id(sp);
    if (L >= sp.nblocks) return;
    const int dep = small_dependency(g, sp, L, (int)threadIdx.x);
    const unsigned e0 = small_epoch_base(a.flags);
    const int vb0 = L * sp.per_block, vb1 = (vb0 + sp.per_block < sp.nvb) ? vb0 + sp.per_block : sp.nvb;
    const CohMem mx = CohMem::make(a.x, a.x_bytes), mq = CohMem::make(a.q, a.q_bytes);
    const CpDualCoh<S, T, V> dual{a.q, a.sigma_D, a.inv_lambda, mq};
    const SrcCoh<T, V> src{a.q, mq};
    for (int it = 0; it < a.n_iter; ++it) {
        TV_SMALL_MARK(0);
        // ---- dual: q <- proj(q + sigma D x), TV(x) (README.md:149-151)
        double acc = 0.0;
        for (int vb = vb0; vb < vb1; ++vb) {
            const Coord c = small_coord<V>(g, sp, vb);
            acc += d_site<S, T, V>(g, w, (const T*)a.x, (const T*)nullptr, (const T*)nullptr, 1, c, dual, mx);
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 0) * sp.nblocks + L] = acc;
        TV_SMALL_MARK(1);
        if (small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 1), &sh_abort)) break;
        TV_SMALL_MARK(2);
        // ---- primal: p <- (p + sigma_A (x - x0)) / (1 + sigma_A); x <- x - tau p - tau D^T q; 1/2 |x - x0|^2 (README.md:148,154,157)
        acc = 0.0;
        for (int vb = vb0; vb < vb1; ++vb) {
            const Coord c = small_coord<V>(g, sp, vb);
            if (c.ok) {
                long long inpl;
                const Vec<T, V> r = dt_site<S, T, V>(g, w, src, c, inpl);
                const long long off = (long long)c.zl * g.s_z + inpl;
                const Vec<T, V> xv = mx.template ld<T, V>(a.x + off), x0v = vload<T, V>(a.x0 + off), pv = vload<T, V>(a.p + off);
                Vec<T, V> pn, xn;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    pn.v[i] = (pv.v[i] + a.sigma_A * (xv.v[i] - x0v.v[i])) * a.inv_1p_sigma_A;
                    xn.v[i] = (xv.v[i] - a.tau * pn.v[i]) - a.tau * r.v[i];
                    const double e = (double)xn.v[i] - (double)x0v.v[i];
                    acc += 0.5 * e * e;
                }
                vstore<T, V>(a.p + off, pn);
                mx.template st<T, V>(a.x + off, xn);
            }
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 1) * sp.nblocks + L] = acc;
        TV_SMALL_MARK(3);
        if (it + 1 < a.n_iter && small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 2), &sh_abort)) break;
        TV_SMALL_MARK(4);
    }
}

// =================================================================================================================================
// REGISTER-RESIDENT form (volumes whose site-vectors fit the launch one per thread: <= blocks x 256 x V voxels, ~1 Mvoxel fp32).
// The generic kernel above spends its phases in four or five DEPENDENT memory round trips (x, then the dual channels of the in-plane axes,
// then z, then t: profiles/r6_small_profile_v1.txt -- 2 us per phase at 64 kvoxel, all latency).  Here a thread owns ONE site-vector for the
// whole launch: x, x0, p and the Nd dual channels of its site live in registers from the first iteration to the last; per phase it issues
// every neighbour load at once (raw buffer loads whose offset is out of range where the neighbour does not exist: no branch around a load,
// one round trip), computes, and stores only what neighbours read (x; the dual channels).  Sites are numbered FLAT inside a frame (row-major
// site-vectors, blockDim.x per block: no lanes wasted on frames narrower than a block), frames and planes as above.  Arithmetic: d_slots + the
// CpDual / dt_site / CpPrimal expressions in their order.  Not with a per-voxel weight volume (generic kernel).
constexpr unsigned kOOB = 0x80000000u;          // every array of these kernels is below 2^31 bytes (small_check)
// Block size of the register-resident kernels: chosen per volume (small_plan_flat) so that every CU gets the SAME number of waves.  With a
// fixed 256 threads the (20,4,100,100) volume makes 800 blocks on 256 CUs -- 3 or 4 per CU, and the blocks of the CUs that hold 4 run every
// phase 1.5 x slower and pace all the others through the neighbour waits; one-wave blocks spread evenly but multiply the flags and the
// dependency chains (13.1 -> 15.8 us per iteration): profiles/r6_small_profile_v3.txt.
constexpr int kRegMaxThreads = 1024;

template <typename T, int V> __device__ __forceinline__ Vec<T, V> coh_ldv(const CohMem& m, unsigned byte_off) {
    if constexpr (V == 1) {
        Vec<T, V> o;
        if constexpr (sizeof(T) == 4) o.v[0] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(m.r, (int)byte_off, 0, CohMem::AUX));
        else { typedef int v2i __attribute__((ext_vector_type(2))); o.v[0] = __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(m.r, (int)byte_off, 0, CohMem::AUX)); }
        return o;
    } else {
        typedef int v4i __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(Vec<T, V>, __builtin_amdgcn_raw_buffer_load_b128(m.r, (int)byte_off, 0, CohMem::AUX));
    }
}
template <typename T> __device__ __forceinline__ T coh_ld1(const CohMem& m, unsigned byte_off) { return coh_ldv<T, 1>(m, byte_off).v[0]; }
template <typename T, int V> __device__ __forceinline__ void coh_stv(const CohMem& m, unsigned byte_off, const Vec<T, V>& v) {
    if constexpr (V == 1 && sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v.v[0]), m.r, (int)byte_off, 0, CohMem::AUX);
    } else if constexpr (V == 1) {
        typedef int v2i __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, v.v[0]), m.r, (int)byte_off, 0, CohMem::AUX);
    } else {
        typedef int v4i __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), m.r, (int)byte_off, 0, CohMem::AUX);
    }
}

// flat numbering: block J = tile * (m nz) + plane * m + frame (TILE-MAJOR: consecutive J -- one XCD -- hold all frames and planes of a band of
// rows, so the frame / plane neighbours of a site share its L2 and only the rows at a band's edge look into another XCD's; with planes
// outermost 80 % of the blocks of a 20-plane volume read a whole neighbour plane from another XCD); thread = site-vector tile * blockDim.x + tid of
// its frame.  Which blocks own neighbouring sites?  In-frame shifts of 1 (columns), nxv (rows), 2 nxv (central sub-gradient) site-vectors
// reach the tiles floor((lo -+ d) / B), floor((hi -+ d) / B), B = blockDim.x; frames / planes one or two away are 1, 2 resp. m, 2 m blocks away.
// Symmetric (see small_dependency).
struct FlatId { int tile, t, zl; };
__device__ __forceinline__ FlatId small_flat_id(const DG& g, int J) {
    const int per_tile = g.m * g.nz, rest = J % per_tile;
    return FlatId{J / per_tile, rest % g.m, rest / g.m};
}
__device__ __forceinline__ int small_dependency_flat(const DG& g, const SmallPlan& sp, int J, int nxv, int i) {
    const int bs = (int)blockDim.x * sp.per_block;          // site-vectors per tile (per_block: site-vectors per thread, 1 in the resident kernels)
    const FlatId id = small_flat_id(g, J);
    if (i < 12) {
        const int which = i >> 2, sign = (i & 2) ? -1 : 1, end = i & 1;
        const long long d = (which == 0) ? 1 : (which == 1) ? nxv : 2ll * nxv;
        const long long s = (long long)id.tile * bs + (end ? bs - 1 : 0) + sign * d;
        if (s < 0) return -1;
        const long long tl = s / bs;
        if (tl >= sp.T || tl == id.tile) return -1;
        return (int)(J + (tl - id.tile) * g.m * g.nz);
    }
    if (i < 16) {
        const int k = i - 12, dt = (k & 1) ? -((k >> 1) + 1) : ((k >> 1) + 1);          // +1, -1, +2, -2 frames
        const int tt = id.t + dt;
        return (g.ta && tt >= 0 && tt < g.m) ? J + dt : -1;
    }
    if (i < 20) {
        const int k = i - 16, dz = (k & 1) ? -((k >> 1) + 1) : ((k >> 1) + 1);
        const int zz = id.zl + dz;
        return (g.za && zz >= 0 && zz < g.nz) ? J + dz * g.m : -1;
    }
    return -1;
}

template <int S, typename T, int V>
__global__ __launch_bounds__(kRegMaxThreads) void k_small_cp_reg(DG g, WT<T> w, SmallPlan sp, SmallCpArgs<T> a) {
    __shared__ double sm[16];
    __shared__ int sh_abort;
    if (threadIdx.x == 0) sh_abort = 0;
    const int L = small_logical_id(sp);
    if (L >= sp.nblocks) return;
    constexpr int NS = (S == HYBRID) ? 8 : 4;
    constexpr unsigned EB = sizeof(T);
    const int nxv = (g.nx + V - 1) / V;
    const int dep = small_dependency_flat(g, sp, L, nxv, (int)threadIdx.x);
    const unsigned e0 = small_epoch_base(a.flags);
    const FlatId fid = small_flat_id(g, L);
    const int tile = fid.tile, t = fid.t, zl = fid.zl;
    const int sidx = tile * (int)blockDim.x + (int)threadIdx.x;
    const bool ok = sidx < g.ny * nxv;
    const int y = ok ? sidx / nxv : 0, col0 = ok ? (sidx % nxv) * V : 0;
    const long long inpl = (long long)t * g.s_t + (long long)y * g.rp + col0;
    const long long offx = (long long)zl * g.s_z + inpl, offq = (long long)zl * g.s_dz + inpl;
    const CohMem mx = CohMem::make(a.x, a.x_bytes), mq = CohMem::make(a.q, a.q_bytes);
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    auto bo = [&](bool valid, long long elem) -> unsigned { return valid ? (unsigned)(elem * EB) : kOOB; };
    // ---- the neighbourhood, once
    XN<T, V> n;
    n.col0 = col0;
    n.h_nr = ok && (y + 1 < g.ny);
    n.h_pr = ok && (y > 0);
    n.h_nz = ok && g.za && (zl + 1 < g.nz);
    n.h_pz = ok && g.za && (zl > 0);
    n.h_nt = ok && g.ta && (t + 1 < g.m);
    n.h_pt = ok && g.ta && (t > 0);
    const bool has_tail = ok && (col0 + V < g.nx), has_head = ok && (col0 > 0);
    constexpr bool NEXT = (S != DOWNWIND), PREV = (S != UPWIND);            // what the forward operator reads (load_xn)
    // Byte offsets are NOT kept per access (25 registers for hybrid: the 4-D hybrid instantiation then spills them and reloads each one behind
    // its own s_waitcnt vmcnt(0) -- the loads of a phase go out one by one again).  Kept: the byte offset of the site in an image (bx0) and in
    // the dual array (bq0), wave-uniform deltas (SGPRs), the validity of every neighbour (lane masks in SGPRs); an access's offset is
    // valid ? base + delta : kOOB, two VALU instructions at the point of use (the bases pass through an empty asm every iteration so that the
    // sums are not hoisted back into registers).
    const unsigned bx0 = (unsigned)(offx * EB), bq0 = (unsigned)(offq * EB);
    const int d_row = g.rp * (int)EB, d_frame = (int)(g.s_t * EB), d_plane = (int)(g.s_z * EB), d_qplane = (int)(g.s_dz * EB);
    // dual channels: byte delta of slot k from channel 0 at my site, < 0 for inactive slots (their loads give 0, their stores are dropped)
    int dq[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) dq[k] = -1;
    for_each_channel<S>(g, [&](auto slot, int ch) {
        if constexpr (decltype(slot)::value < NS) dq[decltype(slot)::value] = (int)((long long)ch * g.s_z * EB);
    });
    auto at = [&](bool valid, unsigned base, int delta) -> unsigned { return valid ? base + (unsigned)delta : kOOB; };
    // neighbours the adjoint reads: slot `up` one step back, slot `down` one step ahead along each axis (the same slot unless hybrid)
    constexpr int U_R = 0, U_C = 1, U_Z = (S == HYBRID) ? 4 : 2, U_T = (S == HYBRID) ? 6 : 3;
    constexpr int D_R = (S == HYBRID) ? 2 : 0, D_C = (S == HYBRID) ? 3 : 1, D_Z = (S == HYBRID) ? 5 : 2, D_T = (S == HYBRID) ? 7 : 3;
    constexpr bool LO = (S != DOWNWIND), HI = (S != UPWIND);
    // ---- the state, once
    Vec<T, V> x = ok ? vload<T, V>(a.x + offx) : zero;
    const Vec<T, V> x0 = ok ? vload<T, V>(a.x0 + offx) : zero;
    Vec<T, V> p = ok ? vload<T, V>(a.p + offx) : zero;
    Vec<T, V> q[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) q[k] = coh_ldv<T, V>(mq, at(ok && dq[k] >= 0, bq0, dq[k]));
    const Vec<T, V> mf = (ok && g.ta) ? mask_factor<T, V>(g, w.sf, y, col0) : vsplat<T, V>(T(1));
    const int gz = zl;          // unsharded: z0 == 0

    for (int it = 0; it < a.n_iter; ++it) {
        TV_SMALL_MARK(0);
        // ---- dual: q <- proj(q + sigma D x), TV(x) (README.md:149-151): every neighbour of x at once
        unsigned bx = bx0, bq = bq0;
        asm volatile("" : "+v"(bx), "+v"(bq));
        n.c = x;
        n.nr = coh_ldv<T, V>(mx, at(NEXT && n.h_nr, bx, d_row)); n.pr = coh_ldv<T, V>(mx, at(PREV && n.h_pr, bx, -d_row));
        n.nz = coh_ldv<T, V>(mx, at(NEXT && n.h_nz, bx, d_plane)); n.pz = coh_ldv<T, V>(mx, at(PREV && n.h_pz, bx, -d_plane));
        n.nt = coh_ldv<T, V>(mx, at(NEXT && n.h_nt, bx, d_frame)); n.pt = coh_ldv<T, V>(mx, at(PREV && n.h_pt, bx, -d_frame));
        const T x_tail = coh_ld1<T>(mx, at(NEXT && has_tail, bx, V * (int)EB)), x_head = coh_ld1<T>(mx, at(PREV && has_head, bx, -(int)EB));
        n.nc = NEXT ? shift_left<T, V>(x, x_tail) : zero;
        n.pc = PREV ? shift_right<T, V>(x, x_head) : zero;
        double acc = 0.0;
        {
            Vec<T, V> o[8];
            d_slots<S, T, V>(g, w, n, mf, o);
            Vec<T, V> vs = zero;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                q[k] = q[k] + a.sigma_D * o[k];
                vs = vs + q[k] * q[k];
            }
            const Vec<T, V> ds = sumsq_slots<T, V>(o);
            Vec<T, V> scale;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                acc += (double)tsqrt(ds.v[i]);
                scale.v[i] = T(1) / tmax(T(1), tsqrt(vs.v[i]) * a.inv_lambda);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                q[k] = q[k] * scale;
                coh_stv<T, V>(mq, at(ok && dq[k] >= 0, bq, dq[k]), q[k]);
            }
            if (!ok) acc = 0.0;
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 0) * sp.nblocks + L] = acc;
        TV_SMALL_MARK(1);
        if (small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 1), &sh_abort)) break;
        TV_SMALL_MARK(2);
        // ---- primal: p <- (p + sigma_A (x - x0)) / (1 + sigma_A); x <- x - tau p - tau D^T q (README.md:148,154): every neighbour of q at once
        asm volatile("" : "+v"(bx), "+v"(bq));
        const Vec<T, V> lo_r = coh_ldv<T, V>(mq, at(LO && n.h_pr, bq, dq[U_R] - d_row)), hi_r = coh_ldv<T, V>(mq, at(HI && n.h_nr, bq, dq[D_R] + d_row));
        const Vec<T, V> lo_z = coh_ldv<T, V>(mq, at(LO && n.h_pz, bq, dq[U_Z] - d_qplane)), hi_z = coh_ldv<T, V>(mq, at(HI && n.h_nz, bq, dq[D_Z] + d_qplane));
        const Vec<T, V> lo_t = coh_ldv<T, V>(mq, at(LO && n.h_pt, bq, dq[U_T] - d_frame)), hi_t = coh_ldv<T, V>(mq, at(HI && n.h_nt, bq, dq[D_T] + d_frame));
        const T q_head = coh_ld1<T>(mq, at(LO && has_head, bq, dq[U_C] - (int)EB)), q_tail = coh_ld1<T>(mq, at(HI && has_tail, bq, dq[D_C] + V * (int)EB));
        Vec<T, V> r = zero, rt = zero;
        auto rows = [&](auto mode, const Vec<T, V>& ce_q) {
            constexpr int M = decltype(mode)::value;
            r = r + adj_axis<M, T, V>(y, g.ny, (M != 1) ? lo_r : zero, (M != 2) ? ce_q : zero, (M != 0) ? hi_r : zero);
        };
        auto cols = [&](auto mode, const Vec<T, V>& ce) {
            constexpr int M = decltype(mode)::value;
            const Vec<T, V> lo = shift_right<T, V>(ce, q_head), hi = shift_left<T, V>(ce, q_tail);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int col = col0 + i;
                T u, v;
                if (M == 0) { u = (col >= 1) ? lo.v[i] : T(0); v = (col <= g.nx - 2) ? ce.v[i] : T(0); }
                else if (M == 1) { u = (col >= 1) ? ce.v[i] : T(0); v = (col <= g.nx - 2) ? hi.v[i] : T(0); }
                else { u = (col >= 2) ? lo.v[i] : T(0); v = (col <= g.nx - 3) ? hi.v[i] : T(0); }
                r.v[i] += u - v;
            }
        };
        auto zax = [&](auto mode, const Vec<T, V>& ce_q) {
            constexpr int M = decltype(mode)::value;
            r = r + w.wz * adj_axis<M, T, V>(gz, g.nzg, (M != 1) ? lo_z : zero, (M != 2) ? ce_q : zero, (M != 0) ? hi_z : zero);
        };
        auto tax = [&](auto mode, const Vec<T, V>& ce_q) {
            constexpr int M = decltype(mode)::value;
            rt = rt + w.wt * adj_axis<M, T, V>(t, g.m, (M != 1) ? lo_t : zero, (M != 2) ? ce_q : zero, (M != 0) ? hi_t : zero);
        };
        if constexpr (S == UPWIND) {
            rows(IC<0>{}, q[0]); cols(IC<0>{}, q[1]);
            if (g.za) zax(IC<0>{}, q[2]);
            if (g.ta) tax(IC<0>{}, q[3]);
        } else if constexpr (S == DOWNWIND) {
            rows(IC<1>{}, q[0]); cols(IC<1>{}, q[1]);
            if (g.za) zax(IC<1>{}, q[2]);
            if (g.ta) tax(IC<1>{}, q[3]);
        } else if constexpr (S == CENTRAL) {
            rows(IC<2>{}, q[0]); cols(IC<2>{}, q[1]);
            if (g.za) { if (g.z_two) zax(IC<0>{}, q[2]); else zax(IC<2>{}, q[2]); }
            if (g.ta) { if (g.t_two) tax(IC<0>{}, q[3]); else tax(IC<2>{}, q[3]); }
        } else {
            // hybrid: the up slot of an axis is read one step back, the down slot one step ahead; the column scalars belong to slots 1 / 3
            rows(IC<0>{}, q[0]);
            { const Vec<T, V> ce = q[1], lo = shift_right<T, V>(ce, q_head);
#pragma unroll
              for (int i = 0; i < V; ++i) { const int col = col0 + i; r.v[i] += ((col >= 1) ? lo.v[i] : T(0)) - ((col <= g.nx - 2) ? ce.v[i] : T(0)); } }
            rows(IC<1>{}, q[2]);
            { const Vec<T, V> ce = q[3], hi = shift_left<T, V>(ce, q_tail);
#pragma unroll
              for (int i = 0; i < V; ++i) { const int col = col0 + i; r.v[i] += ((col >= 1) ? ce.v[i] : T(0)) - ((col <= g.nx - 2) ? hi.v[i] : T(0)); } }
            if (g.za) { zax(IC<0>{}, q[4]); zax(IC<1>{}, q[5]); }
            if (g.ta) { tax(IC<0>{}, q[6]); tax(IC<1>{}, q[7]); }
        }
        if (g.ta) r = r + rt * mf;
        if (S == HYBRID) r = Consts<T>::inv_sqrt2() * r;
        if (S == CENTRAL) r = T(0.5) * r;
        zero_pad_cols<T, V>(g, col0, r);
        acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            p.v[i] = (p.v[i] + a.sigma_A * (x.v[i] - x0.v[i])) * a.inv_1p_sigma_A;
            x.v[i] = (x.v[i] - a.tau * p.v[i]) - a.tau * r.v[i];
            const double e = (double)x.v[i] - (double)x0.v[i];
            acc += 0.5 * e * e;
        }
        if (!ok) acc = 0.0;
        coh_stv<T, V>(mx, at(ok, bx, 0), x);
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 1) * sp.nblocks + L] = acc;
        TV_SMALL_MARK(3);
        if (it + 1 < a.n_iter && small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 2), &sh_abort)) break;
        TV_SMALL_MARK(4);
    }
    if (ok) vstore<T, V>(a.p + offx, p);
}

template <typename T> struct SmallSgArgs {
    T* xa;                   // iterate on entry; even iterations read xa and write xb, odd ones the other way round
    T* xb;
    const T* x0;
    T* norms_ext;            // (nz + 2) planes: 1 / |D x|
    T step, lambda;
    int n_iter;
    unsigned* flags;
    double* partials;
    long long x_bytes, n_bytes;
};

// register-resident descent step: x and x0 of the site stay
// in registers; pass 1 loads the eight neighbours of x at once and stores 1 / |D x|; pass 2 loads the eight neighbours of 1 / |D x| at once
// (the neighbours of x are still in registers), forms G with subgrad_site and stores the new x into the OTHER image buffer for the
// neighbours (ping-pong: the blocks around me may still be reading the old one).
template <int S, typename T, int V>
__global__ __launch_bounds__(kRegMaxThreads) void k_small_sg_reg(DG g, WT<T> w, SmallPlan sp, SmallSgArgs<T> a) {
    __shared__ double sm[16];
    __shared__ int sh_abort;
    if (threadIdx.x == 0) sh_abort = 0;
    const int L = small_logical_id(sp);
    if (L >= sp.nblocks) return;
    constexpr unsigned EB = sizeof(T);
    const int nxv = (g.nx + V - 1) / V;
    const int dep = small_dependency_flat(g, sp, L, nxv, (int)threadIdx.x);
    const unsigned e0 = small_epoch_base(a.flags);
    const FlatId fid = small_flat_id(g, L);
    const int t = fid.t, zl = fid.zl;
    const int sidx = fid.tile * (int)blockDim.x + (int)threadIdx.x;
    const bool ok = sidx < g.ny * nxv;
    const int y = ok ? sidx / nxv : 0, col0 = ok ? (sidx % nxv) * V : 0;
    const long long offx = (long long)zl * g.s_z + (long long)t * g.s_t + (long long)y * g.rp + col0;
    const CohMem mxa = CohMem::make((const T*)a.xa, a.x_bytes), mxb = CohMem::make((const T*)a.xb, a.x_bytes);
    const CohMem mn = CohMem::make((const T*)a.norms_ext + g.s_z, a.n_bytes - g.s_z * (long long)EB);       // (local plane 0 of the extended array)
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    XN<T, V> n, ns;
    n.col0 = ns.col0 = col0;
    n.h_nr = ok && (y + 1 < g.ny);
    n.h_pr = ok && (y > 0);
    n.h_nz = ok && g.za && (zl + 1 < g.nz);
    n.h_pz = ok && g.za && (zl > 0);
    n.h_nt = ok && g.ta && (t + 1 < g.m);
    n.h_pt = ok && g.ta && (t > 0);
    const bool has_tail = ok && (col0 + V < g.nx), has_head = ok && (col0 > 0);
    const unsigned b0 = (unsigned)(offx * EB);
    const int d_row = g.rp * (int)EB, d_frame = (int)(g.s_t * EB), d_plane = (int)(g.s_z * EB);
    auto at = [&](bool valid, unsigned base, int delta) -> unsigned { return valid ? base + (unsigned)delta : kOOB; };
    Vec<T, V> x = ok ? vload<T, V>(a.xa + offx) : zero;
    const Vec<T, V> x0 = ok ? vload<T, V>(a.x0 + offx) : zero;
    const Vec<T, V> mf = (ok && g.ta) ? mask_factor<T, V>(g, w.sf, y, col0) : vsplat<T, V>(T(1));
    constexpr bool NEXT = true, PREV = true;          // the gather of pass 2 reads both sides whatever the scheme (sg_site)
    for (int it = 0; it < a.n_iter; ++it) {
        const CohMem& mxc = (it & 1) ? mxb : mxa;
        const CohMem& mxo = (it & 1) ? mxa : mxb;
        unsigned b = b0;
        asm volatile("" : "+v"(b));
        // ---- pass 1: 1 / |D x|, TV(x) (pytv/tv_GPU.py:84-88)
        n.c = x;
        n.nr = coh_ldv<T, V>(mxc, at(NEXT && n.h_nr, b, d_row)); n.pr = coh_ldv<T, V>(mxc, at(PREV && n.h_pr, b, -d_row));
        n.nz = coh_ldv<T, V>(mxc, at(NEXT && n.h_nz, b, d_plane)); n.pz = coh_ldv<T, V>(mxc, at(PREV && n.h_pz, b, -d_plane));
        n.nt = coh_ldv<T, V>(mxc, at(NEXT && n.h_nt, b, d_frame)); n.pt = coh_ldv<T, V>(mxc, at(PREV && n.h_pt, b, -d_frame));
        const T x_tail = coh_ld1<T>(mxc, at(has_tail, b, V * (int)EB)), x_head = coh_ld1<T>(mxc, at(has_head, b, -(int)EB));
        n.nc = shift_left<T, V>(x, x_tail);
        n.pc = shift_right<T, V>(x, x_head);
        double acc = 0.0;
        Vec<T, V> nv;
        {
            Vec<T, V> o[8];
            d_slots<S, T, V>(g, w, n, mf, o);
            const Vec<T, V> ssq = sumsq_slots<T, V>(o);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const T r = tsqrt(ssq.v[i]);
                acc += (double)r;
                nv.v[i] = (ssq.v[i] >= tiny_sumsq<T>()) ? T(1) / r : T(0);
            }
            if (!ok) acc = 0.0;
        }
        coh_stv<T, V>(mn, at(ok, b, 0), nv);
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 0) * sp.nblocks + L] = acc;
        if (small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 1), &sh_abort)) break;
        // ---- pass 2: G from x and 1 / |D x| (pytv/tv_GPU.py:91-124), x <- x - step ((x - x0) + lambda G) (README.md:122-123)
        asm volatile("" : "+v"(b));
        ns.c = nv;
        ns.nr = coh_ldv<T, V>(mn, at(n.h_nr, b, d_row)); ns.pr = coh_ldv<T, V>(mn, at(n.h_pr, b, -d_row));
        ns.nz = coh_ldv<T, V>(mn, at(n.h_nz, b, d_plane)); ns.pz = coh_ldv<T, V>(mn, at(n.h_pz, b, -d_plane));
        ns.nt = coh_ldv<T, V>(mn, at(n.h_nt, b, d_frame)); ns.pt = coh_ldv<T, V>(mn, at(n.h_pt, b, -d_frame));
        const T n_tail = coh_ld1<T>(mn, at(has_tail, b, V * (int)EB)), n_head = coh_ld1<T>(mn, at(has_head, b, -(int)EB));
        ns.nc = shift_left<T, V>(nv, n_tail);
        ns.pc = shift_right<T, V>(nv, n_head);
        Vec<T, V> G;
        if constexpr (S != CENTRAL) {
            G = subgrad_site<S, T, V>(g, w, n, ns, mf);
        } else {
            // central: G(p) = 1/2 sum_a [ g_a(p-e) - g_a(p+e) ], g_a(q) = 1/2 w_a (x(q+e) - x(q-e)) / |D x|(q) at interior q: x two steps away, 1/|Dx| one
            // step away (the expressions of sg_site_central, tv_site.h); two-point z / t axes: the forward stencil on the neighbours of pass 1
            const T hh = T(0.5);
            const Vec<T, V> xm2r = coh_ldv<T, V>(mxc, at(ok && y >= 2, b, -2 * d_row)), xp2r = coh_ldv<T, V>(mxc, at(ok && y + 2 < g.ny, b, 2 * d_row));
            const Vec<T, V> xm2z = coh_ldv<T, V>(mxc, at(ok && g.za && zl >= 2, b, -2 * d_plane)), xp2z = coh_ldv<T, V>(mxc, at(ok && g.za && zl + 2 < g.nz, b, 2 * d_plane));
            const Vec<T, V> xm2t = coh_ldv<T, V>(mxc, at(ok && g.ta && t >= 2, b, -2 * d_frame)), xp2t = coh_ldv<T, V>(mxc, at(ok && g.ta && t + 2 < g.m, b, 2 * d_frame));
            const T x_head2 = coh_ld1<T>(mxc, at(ok && col0 >= 2, b, -2 * (int)EB)), x_tail2 = coh_ld1<T>(mxc, at(ok && col0 + V + 1 < g.nx, b, (V + 1) * (int)EB));
            const Vec<T, V>& xc = n.c;
            Vec<T, V> r = zero;
            auto cen = [&](int pos, int cnt, const Vec<T, V>& xm2, const Vec<T, V>& xp2, const Vec<T, V>& nm1, const Vec<T, V>& np1, T wa, bool weighted, bool timeax) {
                if (pos - 1 > 0 && pos - 1 < cnt - 1) {
                    Vec<T, V> d = xc - xm2;
                    if (weighted) d = wa * d;
                    if (timeax) d = d * mf;
                    d = hh * d;
#pragma unroll
                    for (int i = 0; i < V; ++i) r.v[i] += d.v[i] * nm1.v[i];
                }
                if (pos + 1 > 0 && pos + 1 < cnt - 1) {
                    Vec<T, V> d = xp2 - xc;
                    if (weighted) d = wa * d;
                    if (timeax) d = d * mf;
                    d = hh * d;
#pragma unroll
                    for (int i = 0; i < V; ++i) r.v[i] -= d.v[i] * np1.v[i];
                }
            };
            auto fwd = [&](int pos, int cnt, const Vec<T, V>& xm1, const Vec<T, V>& xp1, const Vec<T, V>& nm1, const Vec<T, V>& n0, T wa, bool timeax) {
                if (pos >= 1) {
                    Vec<T, V> d = wa * (xc - xm1);
                    if (timeax) d = d * mf;
                    d = hh * d;
#pragma unroll
                    for (int i = 0; i < V; ++i) r.v[i] += d.v[i] * nm1.v[i];
                }
                if (pos <= cnt - 2) {
                    Vec<T, V> d = wa * (xp1 - xc);
                    if (timeax) d = d * mf;
                    d = hh * d;
#pragma unroll
                    for (int i = 0; i < V; ++i) r.v[i] -= d.v[i] * n0.v[i];
                }
            };
            cen(y, g.ny, xm2r, xp2r, ns.pr, ns.nr, T(1), false, false);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const int col = col0 + i;
                const T xl2 = (i >= 2) ? xc.v[(i >= 2) ? i - 2 : 0] : ((i == 1) ? x_head : x_head2);          // x(col - 2)
                const T xr2 = (i + 2 < V) ? xc.v[(i + 2 < V) ? i + 2 : 0] : ((i + 2 == V) ? x_tail : x_tail2);   // x(col + 2)
                const T nl = (i >= 1) ? nv.v[(i >= 1) ? i - 1 : 0] : n_head, nr_ = (i + 1 < V) ? nv.v[(i + 1 < V) ? i + 1 : 0] : n_tail;
                if (col - 1 > 0 && col - 1 < g.nx - 1) r.v[i] += (hh * (xc.v[i] - xl2)) * nl;
                if (col + 1 > 0 && col + 1 < g.nx - 1) r.v[i] -= (hh * (xr2 - xc.v[i])) * nr_;
            }
            if (g.za) {
                if (g.z_two) fwd(zl, g.nzg, n.pz, n.nz, ns.pz, nv, w.wz, false);
                else cen(zl, g.nzg, xm2z, xp2z, ns.pz, ns.nz, w.wz, true, false);
            }
            if (g.ta) {
                if (g.t_two) fwd(t, g.m, n.pt, n.nt, ns.pt, nv, w.wt, true);
                else cen(t, g.m, xm2t, xp2t, ns.pt, ns.nt, w.wt, true, true);
            }
            G = hh * r;
        }
        zero_pad_cols<T, V>(g, col0, G);
        acc = 0.0;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            x.v[i] = x.v[i] - a.step * ((x.v[i] - x0.v[i]) + a.lambda * G.v[i]);
            const double e = (double)x.v[i] - (double)x0.v[i];
            acc += 0.5 * e * e;
        }
        zero_pad_cols<T, V>(g, col0, x);
        if (!ok) acc = 0.0;
        coh_stv<T, V>(mxo, at(ok, b, 0), x);
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 1) * sp.nblocks + L] = acc;
        if (it + 1 < a.n_iter && small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 2), &sh_abort)) break;
    }
}

// =================================================================================================================================
// STREAMED flat form: the layout, the flags and the "every neighbour load of a phase at once" structure of the register-resident kernels, but a
// thread walks over sp.per_block (2 .. 4) site-vectors and re-loads a site's own state in each phase instead of keeping it: for volumes with
// more site-vectors than 16 waves per CU hold -- fp64 at the README shape (20,4,100,100: its native dtype), fp32 between 1 and 4 Mvoxel --
// where the generic form (the per-site bodies of the one-site kernels) walks through four to five dependent memory round trips per phase.
constexpr int kFlatMaxSites = 4;

template <int S, typename T, int V>
__global__ __launch_bounds__(kRegMaxThreads) void k_small_cp_flat(DG g, WT<T> w, SmallPlan sp, SmallCpArgs<T> a) {
    __shared__ double sm[16];
    __shared__ int sh_abort;
    if (threadIdx.x == 0) sh_abort = 0;
    const int L = small_logical_id(sp);
    if (L >= sp.nblocks) return;
    constexpr int NS = (S == HYBRID) ? 8 : 4;
    constexpr unsigned EB = sizeof(T);
    const int nxv = (g.nx + V - 1) / V, ns = sp.per_block;
    const int dep = small_dependency_flat(g, sp, L, nxv, (int)threadIdx.x);
    const unsigned e0 = small_epoch_base(a.flags);
    const FlatId fid = small_flat_id(g, L);
    const int t = fid.t, zl = fid.zl;
    const CohMem mx = CohMem::make(a.x, a.x_bytes), mq = CohMem::make(a.q, a.q_bytes);
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    const int d_row = g.rp * (int)EB, d_frame = (int)(g.s_t * EB), d_plane = (int)(g.s_z * EB), d_qplane = (int)(g.s_dz * EB);
    int dq[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) dq[k] = -1;
    for_each_channel<S>(g, [&](auto slot, int ch) {
        if constexpr (decltype(slot)::value < NS) dq[decltype(slot)::value] = (int)((long long)ch * g.s_z * EB);
    });
    auto at = [&](bool valid, unsigned base, int delta) -> unsigned { return valid ? base + (unsigned)delta : kOOB; };
    constexpr int U_R = 0, U_C = 1, U_Z = (S == HYBRID) ? 4 : 2, U_T = (S == HYBRID) ? 6 : 3;
    constexpr int D_R = (S == HYBRID) ? 2 : 0, D_C = (S == HYBRID) ? 3 : 1, D_Z = (S == HYBRID) ? 5 : 2, D_T = (S == HYBRID) ? 7 : 3;
    constexpr bool LO = (S != DOWNWIND), HI = (S != UPWIND), NEXT = (S != DOWNWIND), PREV = (S != UPWIND);
    // per-pixel factor of the time channels of my sites, once (a mask load per phase would be a dependent round trip)
    Vec<T, V> mfs[kFlatMaxSites];
#pragma unroll
    for (int j = 0; j < kFlatMaxSites; ++j) {
        mfs[j] = vsplat<T, V>(T(1));
        const int sidx = (fid.tile * ns + j) * (int)blockDim.x + (int)threadIdx.x;
        if (j < ns && g.ta && sidx < g.ny * nxv) mfs[j] = mask_factor<T, V>(g, w.sf, sidx / nxv, (sidx % nxv) * V);
    }
    // what a phase needs to know about site j
    struct Site { bool ok, has_tail, has_head; int y, col0; unsigned bx, bq; };
    auto site = [&](int j, XN<T, V>& n) -> Site {
        Site s;
        const int sidx = (fid.tile * ns + j) * (int)blockDim.x + (int)threadIdx.x;
        s.ok = sidx < g.ny * nxv;
        s.y = s.ok ? sidx / nxv : 0;
        s.col0 = s.ok ? (sidx % nxv) * V : 0;
        const long long inpl = (long long)t * g.s_t + (long long)s.y * g.rp + s.col0;
        s.bx = (unsigned)(((long long)zl * g.s_z + inpl) * EB);
        s.bq = (unsigned)(((long long)zl * g.s_dz + inpl) * EB);
        n.col0 = s.col0;
        n.h_nr = s.ok && (s.y + 1 < g.ny);
        n.h_pr = s.ok && (s.y > 0);
        n.h_nz = s.ok && g.za && (zl + 1 < g.nz);
        n.h_pz = s.ok && g.za && (zl > 0);
        n.h_nt = s.ok && g.ta && (t + 1 < g.m);
        n.h_pt = s.ok && g.ta && (t > 0);
        s.has_tail = s.ok && (s.col0 + V < g.nx);
        s.has_head = s.ok && (s.col0 > 0);
        return s;
    };
    const int gz = zl;
    for (int it = 0; it < a.n_iter; ++it) {
        // ---- dual
        double acc = 0.0;
        for (int j = 0; j < ns; ++j) {
            XN<T, V> n;
            const Site s = site(j, n);
            const Vec<T, V> mf = mfs[j < kFlatMaxSites ? j : 0];
            n.c = coh_ldv<T, V>(mx, at(s.ok, s.bx, 0));
            n.nr = coh_ldv<T, V>(mx, at(NEXT && n.h_nr, s.bx, d_row)); n.pr = coh_ldv<T, V>(mx, at(PREV && n.h_pr, s.bx, -d_row));
            n.nz = coh_ldv<T, V>(mx, at(NEXT && n.h_nz, s.bx, d_plane)); n.pz = coh_ldv<T, V>(mx, at(PREV && n.h_pz, s.bx, -d_plane));
            n.nt = coh_ldv<T, V>(mx, at(NEXT && n.h_nt, s.bx, d_frame)); n.pt = coh_ldv<T, V>(mx, at(PREV && n.h_pt, s.bx, -d_frame));
            const T x_tail = coh_ld1<T>(mx, at(NEXT && s.has_tail, s.bx, V * (int)EB)), x_head = coh_ld1<T>(mx, at(PREV && s.has_head, s.bx, -(int)EB));
            Vec<T, V> q[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) q[k] = coh_ldv<T, V>(mq, at(s.ok && dq[k] >= 0, s.bq, dq[k]));
            n.nc = NEXT ? shift_left<T, V>(n.c, x_tail) : zero;
            n.pc = PREV ? shift_right<T, V>(n.c, x_head) : zero;
            Vec<T, V> o[8];
            d_slots<S, T, V>(g, w, n, mf, o);
            Vec<T, V> vs = zero;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                q[k] = q[k] + a.sigma_D * o[k];
                vs = vs + q[k] * q[k];
            }
            const Vec<T, V> ds = sumsq_slots<T, V>(o);
            Vec<T, V> scale;
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                part += (double)tsqrt(ds.v[i]);
                scale.v[i] = T(1) / tmax(T(1), tsqrt(vs.v[i]) * a.inv_lambda);
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) coh_stv<T, V>(mq, at(s.ok && dq[k] >= 0, s.bq, dq[k]), q[k] * scale);
            if (s.ok) acc += part;
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 0) * sp.nblocks + L] = acc;
        if (small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 1), &sh_abort)) break;
        // ---- primal
        acc = 0.0;
        for (int j = 0; j < ns; ++j) {
            XN<T, V> n;
            const Site s = site(j, n);
            const Vec<T, V> mf = mfs[j < kFlatMaxSites ? j : 0];
            const int y = s.y, col0 = s.col0;
            Vec<T, V> q[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) q[k] = coh_ldv<T, V>(mq, at(s.ok && dq[k] >= 0, s.bq, dq[k]));
            const Vec<T, V> lo_r = coh_ldv<T, V>(mq, at(LO && n.h_pr, s.bq, dq[U_R] - d_row)), hi_r = coh_ldv<T, V>(mq, at(HI && n.h_nr, s.bq, dq[D_R] + d_row));
            const Vec<T, V> lo_z = coh_ldv<T, V>(mq, at(LO && n.h_pz, s.bq, dq[U_Z] - d_qplane)), hi_z = coh_ldv<T, V>(mq, at(HI && n.h_nz, s.bq, dq[D_Z] + d_qplane));
            const Vec<T, V> lo_t = coh_ldv<T, V>(mq, at(LO && n.h_pt, s.bq, dq[U_T] - d_frame)), hi_t = coh_ldv<T, V>(mq, at(HI && n.h_nt, s.bq, dq[D_T] + d_frame));
            const T q_head = coh_ld1<T>(mq, at(LO && s.has_head, s.bq, dq[U_C] - (int)EB)), q_tail = coh_ld1<T>(mq, at(HI && s.has_tail, s.bq, dq[D_C] + V * (int)EB));
            Vec<T, V> x = coh_ldv<T, V>(mx, at(s.ok, s.bx, 0));
            const long long offx = (long long)(s.bx / EB);
            const Vec<T, V> x0 = s.ok ? vload<T, V>(a.x0 + offx) : zero;
            Vec<T, V> p = s.ok ? vload<T, V>(a.p + offx) : zero;
            Vec<T, V> r = zero, rt = zero;
            auto rows = [&](auto mode, const Vec<T, V>& ce_q) {
                constexpr int M = decltype(mode)::value;
                r = r + adj_axis<M, T, V>(y, g.ny, (M != 1) ? lo_r : zero, (M != 2) ? ce_q : zero, (M != 0) ? hi_r : zero);
            };
            auto cols = [&](auto mode, const Vec<T, V>& ce) {
                constexpr int M = decltype(mode)::value;
                const Vec<T, V> lo = shift_right<T, V>(ce, q_head), hi = shift_left<T, V>(ce, q_tail);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const int col = col0 + i;
                    T u, v;
                    if (M == 0) { u = (col >= 1) ? lo.v[i] : T(0); v = (col <= g.nx - 2) ? ce.v[i] : T(0); }
                    else if (M == 1) { u = (col >= 1) ? ce.v[i] : T(0); v = (col <= g.nx - 2) ? hi.v[i] : T(0); }
                    else { u = (col >= 2) ? lo.v[i] : T(0); v = (col <= g.nx - 3) ? hi.v[i] : T(0); }
                    r.v[i] += u - v;
                }
            };
            auto zax = [&](auto mode, const Vec<T, V>& ce_q) {
                constexpr int M = decltype(mode)::value;
                r = r + w.wz * adj_axis<M, T, V>(gz, g.nzg, (M != 1) ? lo_z : zero, (M != 2) ? ce_q : zero, (M != 0) ? hi_z : zero);
            };
            auto tax = [&](auto mode, const Vec<T, V>& ce_q) {
                constexpr int M = decltype(mode)::value;
                rt = rt + w.wt * adj_axis<M, T, V>(t, g.m, (M != 1) ? lo_t : zero, (M != 2) ? ce_q : zero, (M != 0) ? hi_t : zero);
            };
            if constexpr (S == UPWIND) {
                rows(IC<0>{}, q[0]); cols(IC<0>{}, q[1]);
                if (g.za) zax(IC<0>{}, q[2]);
                if (g.ta) tax(IC<0>{}, q[3]);
            } else if constexpr (S == DOWNWIND) {
                rows(IC<1>{}, q[0]); cols(IC<1>{}, q[1]);
                if (g.za) zax(IC<1>{}, q[2]);
                if (g.ta) tax(IC<1>{}, q[3]);
            } else if constexpr (S == CENTRAL) {
                rows(IC<2>{}, q[0]); cols(IC<2>{}, q[1]);
                if (g.za) { if (g.z_two) zax(IC<0>{}, q[2]); else zax(IC<2>{}, q[2]); }
                if (g.ta) { if (g.t_two) tax(IC<0>{}, q[3]); else tax(IC<2>{}, q[3]); }
            } else {
                rows(IC<0>{}, q[0]);
                { const Vec<T, V> ce = q[1], lo = shift_right<T, V>(ce, q_head);
#pragma unroll
                  for (int i = 0; i < V; ++i) { const int col = col0 + i; r.v[i] += ((col >= 1) ? lo.v[i] : T(0)) - ((col <= g.nx - 2) ? ce.v[i] : T(0)); } }
                rows(IC<1>{}, q[2]);
                { const Vec<T, V> ce = q[3], hi = shift_left<T, V>(ce, q_tail);
#pragma unroll
                  for (int i = 0; i < V; ++i) { const int col = col0 + i; r.v[i] += ((col >= 1) ? ce.v[i] : T(0)) - ((col <= g.nx - 2) ? hi.v[i] : T(0)); } }
                if (g.za) { zax(IC<0>{}, q[4]); zax(IC<1>{}, q[5]); }
                if (g.ta) { tax(IC<0>{}, q[6]); tax(IC<1>{}, q[7]); }
            }
            if (g.ta) r = r + rt * mf;
            if (S == HYBRID) r = Consts<T>::inv_sqrt2() * r;
            if (S == CENTRAL) r = T(0.5) * r;
            zero_pad_cols<T, V>(g, col0, r);
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                p.v[i] = (p.v[i] + a.sigma_A * (x.v[i] - x0.v[i])) * a.inv_1p_sigma_A;
                x.v[i] = (x.v[i] - a.tau * p.v[i]) - a.tau * r.v[i];
                const double e = (double)x.v[i] - (double)x0.v[i];
                part += 0.5 * e * e;
            }
            coh_stv<T, V>(mx, at(s.ok, s.bx, 0), x);
            if (s.ok) { vstore<T, V>(a.p + offx, p); acc += part; }
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 1) * sp.nblocks + L] = acc;
        if (it + 1 < a.n_iter && small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 2), &sh_abort)) break;
    }
}

template <int S, typename T, int V>
__global__ __launch_bounds__(kRegMaxThreads) void k_small_sg_flat(DG g, WT<T> w, SmallPlan sp, SmallSgArgs<T> a) {
    __shared__ double sm[16];
    __shared__ int sh_abort;
    if (threadIdx.x == 0) sh_abort = 0;
    const int L = small_logical_id(sp);
    if (L >= sp.nblocks) return;
    constexpr unsigned EB = sizeof(T);
    const int nxv = (g.nx + V - 1) / V, ns = sp.per_block;
    const int dep = small_dependency_flat(g, sp, L, nxv, (int)threadIdx.x);
    const unsigned e0 = small_epoch_base(a.flags);
    const FlatId fid = small_flat_id(g, L);
    const int t = fid.t, zl = fid.zl;
    const CohMem mxa = CohMem::make((const T*)a.xa, a.x_bytes), mxb = CohMem::make((const T*)a.xb, a.x_bytes);
    const CohMem mn = CohMem::make((const T*)a.norms_ext + g.s_z, a.n_bytes - g.s_z * (long long)EB);
    const Vec<T, V> zero = vsplat<T, V>(T(0));
    const int d_row = g.rp * (int)EB, d_frame = (int)(g.s_t * EB), d_plane = (int)(g.s_z * EB);
    auto at = [&](bool valid, unsigned base, int delta) -> unsigned { return valid ? base + (unsigned)delta : kOOB; };
    Vec<T, V> mfs[kFlatMaxSites];
#pragma unroll
    for (int j = 0; j < kFlatMaxSites; ++j) {
        mfs[j] = vsplat<T, V>(T(1));
        const int sidx = (fid.tile * ns + j) * (int)blockDim.x + (int)threadIdx.x;
        if (j < ns && g.ta && sidx < g.ny * nxv) mfs[j] = mask_factor<T, V>(g, w.sf, sidx / nxv, (sidx % nxv) * V);
    }
    struct Site { bool ok, has_tail, has_head; int y, col0; unsigned b; };
    auto site = [&](int j, XN<T, V>& n) -> Site {
        Site s;
        const int sidx = (fid.tile * ns + j) * (int)blockDim.x + (int)threadIdx.x;
        s.ok = sidx < g.ny * nxv;
        s.y = s.ok ? sidx / nxv : 0;
        s.col0 = s.ok ? (sidx % nxv) * V : 0;
        s.b = (unsigned)(((long long)zl * g.s_z + (long long)t * g.s_t + (long long)s.y * g.rp + s.col0) * EB);
        n.col0 = s.col0;
        n.h_nr = s.ok && (s.y + 1 < g.ny);
        n.h_pr = s.ok && (s.y > 0);
        n.h_nz = s.ok && g.za && (zl + 1 < g.nz);
        n.h_pz = s.ok && g.za && (zl > 0);
        n.h_nt = s.ok && g.ta && (t + 1 < g.m);
        n.h_pt = s.ok && g.ta && (t > 0);
        s.has_tail = s.ok && (s.col0 + V < g.nx);
        s.has_head = s.ok && (s.col0 > 0);
        return s;
    };
    // the eight neighbours of an image-like array around site s (absent ones: out of range -> 0)
    auto neighbours = [&](const CohMem& m, const Site& s, XN<T, V>& n, const Vec<T, V>& c, T& head, T& tail) {
        n.c = c;
        n.nr = coh_ldv<T, V>(m, at(n.h_nr, s.b, d_row)); n.pr = coh_ldv<T, V>(m, at(n.h_pr, s.b, -d_row));
        n.nz = coh_ldv<T, V>(m, at(n.h_nz, s.b, d_plane)); n.pz = coh_ldv<T, V>(m, at(n.h_pz, s.b, -d_plane));
        n.nt = coh_ldv<T, V>(m, at(n.h_nt, s.b, d_frame)); n.pt = coh_ldv<T, V>(m, at(n.h_pt, s.b, -d_frame));
        tail = coh_ld1<T>(m, at(s.has_tail, s.b, V * (int)EB));
        head = coh_ld1<T>(m, at(s.has_head, s.b, -(int)EB));
    };
    for (int it = 0; it < a.n_iter; ++it) {
        const CohMem& mxc = (it & 1) ? mxb : mxa;
        const CohMem& mxo = (it & 1) ? mxa : mxb;
        // ---- pass 1
        double acc = 0.0;
        for (int j = 0; j < ns; ++j) {
            XN<T, V> n;
            const Site s = site(j, n);
            const Vec<T, V> mf = mfs[j < kFlatMaxSites ? j : 0];
            const Vec<T, V> x = coh_ldv<T, V>(mxc, at(s.ok, s.b, 0));
            T x_head, x_tail;
            neighbours(mxc, s, n, x, x_head, x_tail);
            n.nc = shift_left<T, V>(x, x_tail);
            n.pc = shift_right<T, V>(x, x_head);
            Vec<T, V> o[8], nv;
            d_slots<S, T, V>(g, w, n, mf, o);
            const Vec<T, V> ssq = sumsq_slots<T, V>(o);
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const T r = tsqrt(ssq.v[i]);
                part += (double)r;
                nv.v[i] = (ssq.v[i] >= tiny_sumsq<T>()) ? T(1) / r : T(0);
            }
            coh_stv<T, V>(mn, at(s.ok, s.b, 0), nv);
            if (s.ok) acc += part;
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 0) * sp.nblocks + L] = acc;
        if (small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 1), &sh_abort)) break;
        // ---- pass 2
        acc = 0.0;
        for (int j = 0; j < ns; ++j) {
            XN<T, V> n, nn;
            const Site s = site(j, n);
            nn = n;
            const Vec<T, V> mf = mfs[j < kFlatMaxSites ? j : 0];
            const int y = s.y, col0 = s.col0;
            const unsigned b = s.b;
            const bool ok = s.ok;
            Vec<T, V> x = coh_ldv<T, V>(mxc, at(ok, b, 0));
            const Vec<T, V> nv = coh_ldv<T, V>(mn, at(ok, b, 0));
            T x_head, x_tail, n_head, n_tail;
            neighbours(mxc, s, n, x, x_head, x_tail);
            neighbours(mn, s, nn, nv, n_head, n_tail);
            const Vec<T, V> x0 = ok ? vload<T, V>(a.x0 + (long long)(b / EB)) : zero;
            n.nc = shift_left<T, V>(x, x_tail);
            n.pc = shift_right<T, V>(x, x_head);
            nn.nc = shift_left<T, V>(nv, n_tail);
            nn.pc = shift_right<T, V>(nv, n_head);
            const XN<T, V>& ns_ = nn;
            Vec<T, V> G;
            if constexpr (S != CENTRAL) {
                G = subgrad_site<S, T, V>(g, w, n, ns_, mf);
            } else {
                const T hh = T(0.5);
                const Vec<T, V> xm2r = coh_ldv<T, V>(mxc, at(ok && y >= 2, b, -2 * d_row)), xp2r = coh_ldv<T, V>(mxc, at(ok && y + 2 < g.ny, b, 2 * d_row));
                const Vec<T, V> xm2z = coh_ldv<T, V>(mxc, at(ok && g.za && zl >= 2, b, -2 * d_plane)), xp2z = coh_ldv<T, V>(mxc, at(ok && g.za && zl + 2 < g.nz, b, 2 * d_plane));
                const Vec<T, V> xm2t = coh_ldv<T, V>(mxc, at(ok && g.ta && t >= 2, b, -2 * d_frame)), xp2t = coh_ldv<T, V>(mxc, at(ok && g.ta && t + 2 < g.m, b, 2 * d_frame));
                const T x_head2 = coh_ld1<T>(mxc, at(ok && col0 >= 2, b, -2 * (int)EB)), x_tail2 = coh_ld1<T>(mxc, at(ok && col0 + V + 1 < g.nx, b, (V + 1) * (int)EB));
                const Vec<T, V>& xc = n.c;
                Vec<T, V> r = zero;
                auto cen = [&](int pos, int cnt, const Vec<T, V>& xm2, const Vec<T, V>& xp2, const Vec<T, V>& nm1, const Vec<T, V>& np1, T wa, bool weighted, bool timeax) {
                    if (pos - 1 > 0 && pos - 1 < cnt - 1) {
                        Vec<T, V> d = xc - xm2;
                        if (weighted) d = wa * d;
                        if (timeax) d = d * mf;
                        d = hh * d;
#pragma unroll
                        for (int i = 0; i < V; ++i) r.v[i] += d.v[i] * nm1.v[i];
                    }
                    if (pos + 1 > 0 && pos + 1 < cnt - 1) {
                        Vec<T, V> d = xp2 - xc;
                        if (weighted) d = wa * d;
                        if (timeax) d = d * mf;
                        d = hh * d;
#pragma unroll
                        for (int i = 0; i < V; ++i) r.v[i] -= d.v[i] * np1.v[i];
                    }
                };
                auto fwd = [&](int pos, int cnt, const Vec<T, V>& xm1, const Vec<T, V>& xp1, const Vec<T, V>& nm1, const Vec<T, V>& n0, T wa, bool timeax) {
                    if (pos >= 1) {
                        Vec<T, V> d = wa * (xc - xm1);
                        if (timeax) d = d * mf;
                        d = hh * d;
#pragma unroll
                        for (int i = 0; i < V; ++i) r.v[i] += d.v[i] * nm1.v[i];
                    }
                    if (pos <= cnt - 2) {
                        Vec<T, V> d = wa * (xp1 - xc);
                        if (timeax) d = d * mf;
                        d = hh * d;
#pragma unroll
                        for (int i = 0; i < V; ++i) r.v[i] -= d.v[i] * n0.v[i];
                    }
                };
                cen(y, g.ny, xm2r, xp2r, ns_.pr, ns_.nr, T(1), false, false);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const int col = col0 + i;
                    const T xl2 = (i >= 2) ? xc.v[(i >= 2) ? i - 2 : 0] : ((i == 1) ? x_head : x_head2);
                    const T xr2 = (i + 2 < V) ? xc.v[(i + 2 < V) ? i + 2 : 0] : ((i + 2 == V) ? x_tail : x_tail2);
                    const T nl = (i >= 1) ? nv.v[(i >= 1) ? i - 1 : 0] : n_head, nr_ = (i + 1 < V) ? nv.v[(i + 1 < V) ? i + 1 : 0] : n_tail;
                    if (col - 1 > 0 && col - 1 < g.nx - 1) r.v[i] += (hh * (xc.v[i] - xl2)) * nl;
                    if (col + 1 > 0 && col + 1 < g.nx - 1) r.v[i] -= (hh * (xr2 - xc.v[i])) * nr_;
                }
                if (g.za) {
                    if (g.z_two) fwd(zl, g.nzg, n.pz, n.nz, ns_.pz, nv, w.wz, false);
                    else cen(zl, g.nzg, xm2z, xp2z, ns_.pz, ns_.nz, w.wz, true, false);
                }
                if (g.ta) {
                    if (g.t_two) fwd(t, g.m, n.pt, n.nt, ns_.pt, nv, w.wt, true);
                    else cen(t, g.m, xm2t, xp2t, ns_.pt, ns_.nt, w.wt, true, true);
                }
                G = hh * r;
            }
            zero_pad_cols<T, V>(g, col0, G);
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                x.v[i] = x.v[i] - a.step * ((x.v[i] - x0.v[i]) + a.lambda * G.v[i]);
                const double e = (double)x.v[i] - (double)x0.v[i];
                part += 0.5 * e * e;
            }
            zero_pad_cols<T, V>(g, col0, x);
            coh_stv<T, V>(mxo, at(ok, b, 0), x);
            if (ok) acc += part;
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 1) * sp.nblocks + L] = acc;
        if (it + 1 < a.n_iter && small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 2), &sh_abort)) break;
    }
}

template <int S, typename T, int V>
__global__ __launch_bounds__(kSmallThreads) void k_small_sg(DG g, WT<T> w, SmallPlan sp, SmallSgArgs<T> a) {
    __shared__ double sm[16];
    __shared__ int sh_abort;
    if (threadIdx.x == 0) sh_abort = 0;
    const int L = small_logical_id(sp);
    if (L >= sp.nblocks) return;
    const int dep = small_dependency(g, sp, L, (int)threadIdx.x);
    const unsigned e0 = small_epoch_base(a.flags);
    const int vb0 = L * sp.per_block, vb1 = (vb0 + sp.per_block < sp.nvb) ? vb0 + sp.per_block : sp.nvb;
    const CohMem mn = CohMem::make(a.norms_ext, a.n_bytes);
    const NormEpiCoh<S, T, V> nepi{a.norms_ext, mn};
    for (int it = 0; it < a.n_iter; ++it) {
        const T* xc = (it & 1) ? a.xb : a.xa;
        T* xo = (it & 1) ? a.xa : a.xb;
        const CohMem mxc = CohMem::make(xc, a.x_bytes), mxo = CohMem::make((const T*)xo, a.x_bytes);
        // ---- pass 1: 1 / |D x| per voxel, TV(x) (pytv/tv_GPU.py:84-88)
        double acc = 0.0;
        for (int vb = vb0; vb < vb1; ++vb) {
            const Coord c = small_coord<V>(g, sp, vb);
            acc += d_site<S, T, V>(g, w, xc, (const T*)nullptr, (const T*)nullptr, 2, c, nepi, mxc);
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 0) * sp.nblocks + L] = acc;
        if (small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 1), &sh_abort)) break;
        // ---- pass 2: G from x and 1 / |D x| (pytv/tv_GPU.py:91-124), x <- x - step ((x - x0) + lambda G) (README.md:122-123)
        acc = 0.0;
        for (int vb = vb0; vb < vb1; ++vb) {
            const Coord c = small_coord<V>(g, sp, vb);
            if (c.ok) {
                Vec<T, V> G;
                if constexpr (S == CENTRAL) G = sg_site_central<T, V>(g, w, xc, (const T*)nullptr, (const T*)nullptr, (const T*)a.norms_ext, c, mxc, mn);
                else G = sg_site<S, T, V>(g, w, xc, (const T*)nullptr, (const T*)nullptr, (const T*)a.norms_ext, c, mxc, mn);
                const long long off = (long long)c.zl * g.s_z + (long long)c.t * g.s_t + (long long)c.y * g.rp + c.col0;
                const Vec<T, V> xv = mxc.template ld<T, V>(xc + off), x0v = vload<T, V>(a.x0 + off);
                Vec<T, V> xn;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    xn.v[i] = xv.v[i] - a.step * ((xv.v[i] - x0v.v[i]) + a.lambda * G.v[i]);
                    const double e = (double)xn.v[i] - (double)x0v.v[i];
                    acc += 0.5 * e * e;
                }
                zero_pad_cols<T, V>(g, c.col0, xn);
                mxo.template st<T, V>(xo + off, xn);
            }
        }
        acc = block_sum(acc, sm);
        if (threadIdx.x == 0) a.partials[((long long)it * 2 + 1) * sp.nblocks + L] = acc;
        if (it + 1 < a.n_iter && small_sync(a.flags, L, dep, e0 + (unsigned)(2 * it + 2), &sh_abort)) break;
    }
}

// hist[(row / 2) * stride + (row & 1 ? fid_offset : 0)] = sum over the blocks of partials[row][*] (fixed order: deterministic); the first
// block also advances the workspace's epoch past this call's phases
__global__ __launch_bounds__(256) void k_small_reduce(const double* partials, int nblocks, double* hist, long long stride, long long fid_offset,
                                                      unsigned* flags, unsigned advance) {
    __shared__ double sm[16];
    const double* p = partials + (long long)blockIdx.x * nblocks;
    double acc = 0.0;
    for (int i = (int)threadIdx.x; i < nblocks; i += 256) acc += p[i];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) {
        if (*small_abort_word(flags) != 0) acc = __builtin_nan("");          // the launch was abandoned (small_sync): no number is valid
        hist[(long long)(blockIdx.x >> 1) * stride + ((blockIdx.x & 1) ? fid_offset : 0)] = acc;
        if (blockIdx.x == 0) flags[(long long)kMaxSmallBlocks * kFlagStride] += advance;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
static int small_capacity(const void* kernel) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kSmallThreads, 0) != hipSuccess) return 0;
    // at most 4 blocks of 256 threads per CU = 16 waves, half of a CU's wave slots (the register-resident plans keep the same bound): the occupancy
    // query allows 7 - 8 for the scalar-lane instantiations and hipLaunchCooperativeKernel accepts them, but 7 per CU were observed NOT to be resident
    // together (1764 blocks: the launch hung; 6 per CU ran) -- and more blocks per CU buy nothing here (profiles/r6_small_volume_v2: 8 = 4)
    const int cap_opt = env_int("TV_SMALL_BLOCKS_PER_CU", 4);
    if (per_cu > cap_opt) per_cu = cap_opt;
    return per_cu * cus;
}

static SmallPlan small_plan(const DG& d, int V, int capacity) {
    SmallPlan sp{};
    const LC lc = launch_cfg(d, V, d.nz);
    sp.bx = (int)lc.block.x;
    sp.by = (int)lc.block.y;
    const int nxv = (d.nx + V - 1) / V;
    sp.tiles_x = (nxv + sp.bx - 1) / sp.bx;
    sp.tiles_y = (d.ny + sp.by - 1) / sp.by;
    sp.T = sp.tiles_x * sp.tiles_y;
    sp.nvb = sp.T * d.m * d.nz;
    int cap = capacity / 8 * 8;
    if (cap < 8) cap = 8;
    sp.per_block = (sp.nvb + cap - 1) / cap;
    sp.nblocks = (sp.nvb + sp.per_block - 1) / sp.per_block;
    sp.grid = (sp.nblocks + 7) / 8 * 8;
    return sp;
}

// the register-resident kernels: flat site numbering, ONE block per 256 site-vectors of a frame; usable iff the launch can hold them all
static bool small_plan_flat(const DG& d, int V, const void* kernel, SmallPlan& sp, int& threads, int sites = 1) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) return false;
    if (d.wv != nullptr || env_int("TV_SMALL_GENERIC", 0)) return false;
    const long long nxv = (d.nx + V - 1) / V, sf = (long long)d.ny * nxv, frames = (long long)d.m * d.nz;
    // T tiles per frame, blocks of roundup64(ceil(sf / (T sites))) threads, `sites` site-vectors per thread (1: the register-resident kernels).
    // Cost of a choice = waves on the most loaded CU (<= 16: 128 registers per thread, and the residency bound of small_capacity) + blocks on it
    // (every co-resident block adds flags and dependency chains).  Measured on (20,4,100,100) hybrid, us per iteration: T = 3 (240 blocks x 896
    // threads) 12.0, T = 5 (400 x 512) 12.7, T = 10 (800 x 256) 13.1, T = 40 (3200 x 64) 15.8 -- the cost orders them the same way (15, 18, 20,
    // 26): profiles/r6_small_volume_tiles.txt.  Ties: the smaller block.
    const int force = env_int("TV_SMALL_TILES", 0);
    long long best_T = 0, best_cost = 1ll << 60, best_bs = 0;
    for (long long T = 1; T <= sf; ++T) {
        long long bs = ((sf + T * sites - 1) / (T * sites) + 63) / 64 * 64;
        if (bs < 64) bs = 64;
        if (bs > kRegMaxThreads || (force > 0 && T != force)) { if (bs == 64) break; continue; }
        const long long nb = T * frames, per_cu = (nb + cus - 1) / cus, waves = per_cu * (bs / 64);
        if (waves <= 16 && nb <= kMaxSmallBlocks) {
            const long long cost = waves + per_cu;
            if (cost <= best_cost) { best_cost = cost; best_T = T; best_bs = bs; }
        }
        if (bs == 64) break;
    }
    if (best_T == 0) return false;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, (int)best_bs, 0) != hipSuccess) return false;
    const long long nvb = best_T * frames;
    if (nvb > (long long)per_cu * cus) return false;
    sp = SmallPlan{};
    sp.bx = (int)best_bs; sp.by = 1;
    sp.tiles_x = (int)best_T; sp.tiles_y = 1;
    sp.T = (int)best_T;
    sp.nvb = (int)nvb;
    sp.per_block = sites;            // flat kernels: site-vectors per thread
    sp.nblocks = (int)nvb;
    sp.grid = (sp.nblocks + 7) / 8 * 8;
    threads = (int)best_bs;
    return true;
}

static long long small_max_voxels() { return 1024ll * env_int("TV_SMALL_MAX_KVOXELS", 4096); }

static int small_check(const tv_geom* g, DG& d, const char* who) {
    if (int rc = make_dg(g, d, true)) return rc;
    if (g->z0 != 0 || g->nz != g->nz_global) return fail(TV_E_ARG, "the persistent small-volume kernels take an unsharded volume (nz == nz_global)");
    const long long eb = (g->dtype == TV_F32) ? 4 : 8;
    if (d.s_z * d.nz > small_max_voxels() || d.s_dz * (d.nz + 2) * eb >= (1ll << 31))
        return fail(TV_E_ARG, who);
    return 0;
}

}  // namespace tv

using namespace tv;

extern "C" {

int tv_small_supported(const tv_geom* g) {
    DG d;
    if (make_dg(g, d, true)) return 0;
    if (g->z0 != 0 || g->nz != g->nz_global) return 0;
    const long long eb = (g->dtype == TV_F32) ? 4 : 8;
    if (d.s_z * d.nz > small_max_voxels() || d.s_dz * (d.nz + 2) * eb >= (1ll << 31)) return 0;
    if (env_int("TV_NO_SMALL", 0)) return 0;
    return 1;
}

size_t tv_small_workspace_bytes(const tv_geom* g, int64_t n_iter) {
    DG d;
    if (make_dg(g, d, true) || n_iter < 1) return 0;
    // flags: one 128-byte line per launched block; partials: n_iter x 2 x blocks doubles
    const size_t blocks = kMaxSmallBlocks;
    const size_t flag_bytes = (blocks + 1) * kFlagStride * sizeof(unsigned);      // (+ the line that holds the epoch)
#ifdef TV_SMALL_PROFILE
    return flag_bytes + (size_t)n_iter * 7 * blocks * sizeof(double) + 256;
#else
    return flag_bytes + (size_t)n_iter * 2 * blocks * sizeof(double) + 256;
#endif
}

int tv_small_cp(const tv_geom* g, void* x, const void* x0, void* p, void* q, double sigma_D, double lambda, double tau, double sigma_A,
                int64_t n_iter, double* hist, int64_t hist_stride, int64_t hist_fid_offset, void* ws, void* stream) {
    DG d;
    if (int rc = small_check(g, d, "tv_small_cp: volume too large for the persistent kernel (tv_small_supported)")) return rc;
    if (x == nullptr || x0 == nullptr || p == nullptr || q == nullptr || hist == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (!(lambda > 0.0)) return fail(TV_E_ARG, "lambda must be > 0");
    if (n_iter < 1 || n_iter > (1 << 20)) return fail(TV_E_ARG, "n_iter out of range");
    if (hist_stride < 1 || hist_fid_offset == 0 || hist_fid_offset >= hist_stride || hist_fid_offset < 0) return fail(TV_E_ARG, "hist_stride / hist_fid_offset: 0 < fid_offset < stride");
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x0, p, q, d.wv});
    hipStream_t st = (hipStream_t)stream;
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        const void* kern = (const void*)k_small_cp_reg<S, T, V>;
        SmallPlan sp;
        int threads = kSmallThreads;
        // TV_SMALL_SITES: 0 = resident form first, then the streamed form with 2 .. 4 site-vectors per thread; 1 = resident only; 2 .. 4 = streamed from there
        const int s0 = env_int("TV_SMALL_SITES", 0);
        bool flat = s0 <= 1 && small_plan_flat(d, V, kern, sp, threads);
        for (int sites = s0 > 2 ? s0 : 2; !flat && sites <= kFlatMaxSites && s0 != 1; ++sites) {
            kern = (const void*)k_small_cp_flat<S, T, V>;
            flat = small_plan_flat(d, V, kern, sp, threads, sites);
        }
        if (!flat) {
            kern = (const void*)k_small_cp<S, T, V>;
            threads = kSmallThreads;
            const int cap = small_capacity(kern);
            if (cap < 8) return fail(TV_E_ARG, "tv_small_cp: no HIP device / occupancy query failed");
            sp = small_plan(d, V, cap);
        }
        if (sp.grid > kMaxSmallBlocks) return fail(TV_E_ARG, "internal: more blocks than the workspace holds");
        unsigned* flags = (unsigned*)ws;
        double* partials = (double*)((char*)ws + (size_t)(kMaxSmallBlocks + 1) * kFlagStride * sizeof(unsigned));
        DG dd = d;
        WT<T> w = make_w<T>(g);
        SmallCpArgs<T> a{(T*)x, (const T*)x0, (T*)p, (T*)q, (T)sigma_D, (T)(1.0 / lambda), (T)tau, (T)sigma_A, (T)(1.0 / (1.0 + sigma_A)), (int)n_iter,
                         flags, partials, d.s_z * d.nz * (long long)sizeof(T), d.s_dz * d.nz * (long long)sizeof(T)};
        void* args[] = {&dd, &w, &sp, &a};
        HIP_TRY(hipLaunchCooperativeKernel(kern, dim3((unsigned)sp.grid), dim3((unsigned)threads), args, 0, st));
        hipLaunchKernelGGL(k_small_reduce, dim3((unsigned)(2 * n_iter)), dim3(256), 0, st, (const double*)partials, sp.nblocks, hist, (long long)hist_stride,
                           (long long)hist_fid_offset, flags, (unsigned)(2 * n_iter));
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

int tv_small_subgrad_descent(const tv_geom* g, void* x, void* x_alt, const void* x0, void* norms_ext, double step, double lambda, int64_t n_iter,
                             double* hist, int64_t hist_stride, int64_t hist_fid_offset, void* ws, void* stream) {
    DG d;
    if (int rc = small_check(g, d, "tv_small_subgrad_descent: volume too large for the persistent kernel (tv_small_supported)")) return rc;
    if (x == nullptr || x_alt == nullptr || x0 == nullptr || norms_ext == nullptr || hist == nullptr || ws == nullptr) return fail(TV_E_ARG, "NULL array");
    if (x == x_alt) return fail(TV_E_ARG, "x and x_alt must be different arrays (the iterate is ping-ponged)");
    if (n_iter < 1 || n_iter > (1 << 20)) return fail(TV_E_ARG, "n_iter out of range");
    if (hist_stride < 1 || hist_fid_offset == 0 || hist_fid_offset >= hist_stride || hist_fid_offset < 0) return fail(TV_E_ARG, "hist_stride / hist_fid_offset: 0 < fid_offset < stride");
    const bool vec = rows_vectorisable(g, d) && aligned16({x, x_alt, x0, norms_ext, d.wv});
    hipStream_t st = (hipStream_t)stream;
    return dispatch(g->scheme, g->dtype, vec, [&]<int S, typename T, int V>() -> int {
        const void* kern = nullptr;
        SmallPlan sp;
        int threads = kSmallThreads;
        bool flat = false;
        kern = (const void*)k_small_sg_reg<S, T, V>;
        const int s0 = env_int("TV_SMALL_SITES", 0);
        flat = s0 <= 1 && small_plan_flat(d, V, kern, sp, threads);
        for (int sites = s0 > 2 ? s0 : 2; !flat && sites <= kFlatMaxSites && s0 != 1; ++sites) {
            kern = (const void*)k_small_sg_flat<S, T, V>;
            flat = small_plan_flat(d, V, kern, sp, threads, sites);
        }
        if (!flat) {
            kern = (const void*)k_small_sg<S, T, V>;
            threads = kSmallThreads;
            const int cap = small_capacity(kern);
            if (cap < 8) return fail(TV_E_ARG, "tv_small_subgrad_descent: no HIP device / occupancy query failed");
            sp = small_plan(d, V, cap);
        }
        if (sp.grid > kMaxSmallBlocks) return fail(TV_E_ARG, "internal: more blocks than the workspace holds");
        unsigned* flags = (unsigned*)ws;
        double* partials = (double*)((char*)ws + (size_t)(kMaxSmallBlocks + 1) * kFlagStride * sizeof(unsigned));
        DG dd = d;
        WT<T> w = make_w<T>(g);
        SmallSgArgs<T> a{(T*)x, (T*)x_alt, (const T*)x0, (T*)norms_ext, (T)step, (T)lambda, (int)n_iter, flags, partials,
                         d.s_z * d.nz * (long long)sizeof(T), d.s_z * (d.nz + 2) * (long long)sizeof(T)};
        void* args[] = {&dd, &w, &sp, &a};
        HIP_TRY(hipLaunchCooperativeKernel(kern, dim3((unsigned)sp.grid), dim3((unsigned)threads), args, 0, st));
        hipLaunchKernelGGL(k_small_reduce, dim3((unsigned)(2 * n_iter)), dim3(256), 0, st, (const double*)partials, sp.nblocks, hist, (long long)hist_stride,
                           (long long)hist_fid_offset, flags, (unsigned)(2 * n_iter));
        HIP_TRY(hipGetLastError());
        return 0;
    });
}

}  // extern "C"
