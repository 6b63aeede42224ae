// tv_subgrad2.h -- ONE-PASS TV value + sub-gradient (pytv/tv_GPU.py:47-375 of the reference), round-3 kernel.
//
// Same mathematics as the round-1 kernel (retired, in the history; the scatter form: a site hands the PRODUCTS d * 1/|Dx| of its gradient channels to
// its neighbours, 1/|Dx| never leaves the chip, the result is a pure function of x), rebuilt around what an instruction
// costs on a gfx950 SIMD (tools/archive/issue_bench.hip, profiles/r3_issue_bench.txt): a plain fp32 VALU op 2 cycles, DPP / compare
// / select / packed / fp64 4, v_rsq 6 - 8, and ds_bpermute_b32 -- what __shfl_up/down compile to -- 18 per dword.  The
// round-1 kernel (a lane = 1 row x 4 columns, wave = 4 rows x 16 lanes) spent a third of its issue time in the 19
// bpermutes per frame that move row neighbours between lanes.  Here
//
//   a lane = R ROWS x 1 column (a column strip), a wave = R rows x 64 columns, NW waves stacked in y form the block;
//   * row neighbours are registers of the same thread;
//   * column neighbours are ONE DPP move each (wave_shr:1 / wave_shl:1 cross the 16-lane rows on gfx9);
//   * only the first / last row of a wave's strip talks to another wave: one float per lane through LDS, double
//     buffered so that ONE barrier per plane suffices (x of plane z+1 is published while plane z is computed);
//   * the tile-overlap ring costs one or two COLUMNS on either side (2 - 4 of 64 lanes) instead of one 4-column lane (2 of 16);
//   * a block whose tile lies strictly inside the frame runs a variant without any border multiplier / select
//     (block-uniform branch), the zero-gradient rule is a wave-uniform branch around the selects;
//   * z marching as before: x(z-1), x(z) and the accumulators G(z-1), G(z) of all M frames in registers, x(z+1) arrives
//     through a short ring of loads issued two frames ahead.
// Global accesses are 4 bytes per lane, 256 contiguous bytes per row and wave.
#pragma once
#include "tv_device.h"
#include "tv_stencil.h"
#include "tv_fused.h"

namespace tv {

template <typename T, int R> struct Col { T v[R]; };

// lane i gets the value of lane i-1 / i+1 of the WAVE (0 at the wave's ends: those are ring lanes)
__device__ __forceinline__ float from_left(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float from_right(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ double from_left(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double from_right(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x130, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x130, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// the same with a value for the lane that has no source (lane 0 of a shift to the right, lane 63 of a shift to the left): the
// dpp "old" operand with bound_ctrl off -- how the ALIGNED tiles (AL, round 5) put the ring column's value next to the edge lane
__device__ __forceinline__ float from_left(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float from_right(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x130, 0xF, 0xF, false));
}
__device__ __forceinline__ double from_left(double v, double edge) {
    const long long b = __double_as_longlong(v), e = __double_as_longlong(edge);
    const int lo = __builtin_amdgcn_update_dpp((int)e, (int)b, 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(b >> 32), 0x138, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double from_right(double v, double edge) {
    const long long b = __double_as_longlong(v), e = __double_as_longlong(edge);
    const int lo = __builtin_amdgcn_update_dpp((int)e, (int)b, 0x130, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(e >> 32), (int)(b >> 32), 0x130, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ float rsq_fast(float v) { return __builtin_amdgcn_rsqf(v); }
// fp64: v_rsq_f64 is good to ~2^-26; two Newton steps y <- y (1.5 - 0.5 v y^2) bring it to ~1 ulp
__device__ __forceinline__ double rsq_fast(double v) {
    double y = __builtin_amdgcn_rsq(v);
    const double h = 0.5 * v;
    y = y * (1.5 - h * y * y);
    y = y * (1.5 - h * y * y);
    return y;
}
// ---- buffer addressing ------------------------------------------------------------------------------------------------
// Every global access of the kernel is a raw buffer load / store: a wave-uniform descriptor (base = one frame of one plane,
// num_records = bytes of a frame) + a per-lane byte offset.  The hardware's range check does the predication: a load whose
// offset lies beyond num_records returns 0, such a store is dropped.  Sites outside the frame, ring rows / lanes that must not
// be stored, planes that do not exist (descriptor with num_records = 0) therefore need NO branch and NO exec masking -- which
// matters beyond the instruction count: gfx950 has one in-order vmcnt for loads and stores, and with control flow around
// memory operations the compiler can only wait with vmcnt(0), i.e. for every outstanding store and look-ahead load, once
// per frame (the first version of this kernel did: 53 % of its wave cycles waiting).  Straight-line code gets counted waits.
// (Rsrc: tv_fused.h)
constexpr unsigned SG2_OOB = 0x80000000u;         // beyond any frame (frames are < 2^31 bytes: sg2_supported)
typedef int sg2_v2i __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ Rsrc sg2_rsrc(const T* base, bool valid, int nbytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, valid ? nbytes : 0, 0x00020000);
}
__device__ __forceinline__ float sg2_ld(Rsrc r, unsigned off, float) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); }
__device__ __forceinline__ double sg2_ld(Rsrc r, unsigned off, double) {
    const sg2_v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    return __longlong_as_double(((long long)v.y << 32) | (unsigned)v.x);
}
#ifndef TV_SG2_NT
#define TV_SG2_NT 0            // 1: EXPERIMENT stores (G / x_out / norms) and x0 loads non-temporal (aux bit 1): 5 - 10 % SLOWER with 4-byte lanes (hybrid loop 7.5 -> 8.1 ms)
#endif
constexpr int SG2_AUX_S = (TV_SG2_NT == 1) ? 2 : 0;      // stores
constexpr int SG2_AUX_L = (TV_SG2_NT != 0) ? 2 : 0;      // x0 loads (TV_SG2_NT=2: loads only -- no measurable difference, 6.9 - 7.1 ms either way)
__device__ __forceinline__ float sg2_ld_s(Rsrc r, unsigned off, float) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, SG2_AUX_L)); }
__device__ __forceinline__ double sg2_ld_s(Rsrc r, unsigned off, double) {
    const sg2_v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, SG2_AUX_L);
    return __longlong_as_double(((long long)v.y << 32) | (unsigned)v.x);
}
__device__ __forceinline__ void sg2_st(Rsrc r, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), r, (int)off, 0, SG2_AUX_S); }
__device__ __forceinline__ void sg2_st(Rsrc r, unsigned off, double v) {
    const long long b = __double_as_longlong(v);
    sg2_v2i w;
    w.x = (int)b;
    w.y = (int)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, (int)off, 0, SG2_AUX_S);
}
// A frame of a plane as the kernel addresses it: a descriptor per frame (so = 0), or -- TV_SG2_PLANE_DESC, a round-6 experiment that
// removed 70 scalar instructions per plane step and gained nothing -- ONE descriptor per plane and stream + the frame's byte offset as the
// instruction's scalar offset (the descriptor spans the whole plane, so it does not matter whether the hardware counts the scalar offset in
// its range check; a lane that must not take part still has the offset 2^31 >= num_records; time windows keep a descriptor per frame).
struct SgFrame { Rsrc r; int so; };
__device__ __forceinline__ float sg2_ld(const SgFrame& f, unsigned off, float) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(f.r, (int)off, f.so, 0)); }
__device__ __forceinline__ double sg2_ld(const SgFrame& f, unsigned off, double) {
    const sg2_v2i v = __builtin_amdgcn_raw_buffer_load_b64(f.r, (int)off, f.so, 0);
    return __longlong_as_double(((long long)v.y << 32) | (unsigned)v.x);
}
__device__ __forceinline__ float sg2_ld_s(const SgFrame& f, unsigned off, float) { return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(f.r, (int)off, f.so, SG2_AUX_L)); }
__device__ __forceinline__ double sg2_ld_s(const SgFrame& f, unsigned off, double) {
    const sg2_v2i v = __builtin_amdgcn_raw_buffer_load_b64(f.r, (int)off, f.so, SG2_AUX_L);
    return __longlong_as_double(((long long)v.y << 32) | (unsigned)v.x);
}
__device__ __forceinline__ void sg2_st(const SgFrame& f, unsigned off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), f.r, (int)off, f.so, SG2_AUX_S); }
__device__ __forceinline__ void sg2_st(const SgFrame& f, unsigned off, double v) {
    const long long b = __double_as_longlong(v);
    sg2_v2i w;
    w.x = (int)b;
    w.y = (int)(b >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(w, f.r, (int)off, f.so, SG2_AUX_S);
}
template <typename T> __device__ __forceinline__ void pin1(T& a);
template <> __device__ __forceinline__ void pin1<float>(float& a) { asm volatile("" : "+v"(a)); }
template <> __device__ __forceinline__ void pin1<double>(double& a) { asm volatile("" : "+v"(a)); }

template <typename T> struct SgArgs2 {
    const T* x0;
    T* x_out;
    T step, lambda;
    double* part_fid;
    T* norms;
};

constexpr int SG2_TWN = 8, SG2_TWU = SG2_TWN - 2;      // time windows for M > 8: 8 frames computed, 6 stored
#ifndef TV_SG2_D
#define TV_SG2_D 2
#endif
#ifndef TV_SG2_COPYONLY
#define TV_SG2_COPYONLY 0      // 1: EXPERIMENT -- store x instead of G and drop the arithmetic: what the memory pattern alone costs
#endif
#ifndef TV_SG2_STALIGN
#define TV_SG2_STALIGN 0       // 1: EXPERIMENT -- stores at line-aligned positions (results wrong): the price of partial-line stores
#endif
#ifndef TV_SG2_LDALIGN
#define TV_SG2_LDALIGN 0       // 1: EXPERIMENT -- loads at line-aligned positions (results wrong)
#endif
#ifndef TV_SG2_WIDE
#define TV_SG2_WIDE 0          // 1: EXPERIMENT (with COPYONLY) -- the waves of a block side by side: 4 rows x 256 columns, no ring
#endif
#ifndef TV_SG2_X0_AHEAD
#define TV_SG2_X0_AHEAD 1
#endif
#ifndef TV_SG2_PLANE_DESC
#define TV_SG2_PLANE_DESC 0    // 1: EXPERIMENT (round 6) -- one buffer descriptor per plane and stream, the frame as the scalar offset: 70 scalar instructions fewer per
                               // plane step (1806 -> 1736) and NO gain (hybrid 1.418 -> 1.428 ms, profiles/r6_sg_plane_desc_ab.txt): scalar instructions are not what the SIMD waits for
#endif
#ifndef TV_SG2_PRIO
#define TV_SG2_PRIO 0          // 1 .. 3: EXPERIMENT (round 6) -- alternate the issue priority of the two waves of a SIMD frame by frame
#endif
#ifndef TV_SG2_SCHED_BARRIER
#define TV_SG2_SCHED_BARRIER 1
#endif
constexpr int SG2_D = TV_SG2_D;                        // depth of the x(z+1) load ring (frames ahead)

// MODE 0: G stored.  MODE 1: the descent step x_out = x - step ((x - x0) + lambda G), 1/2 |x_out - x0|^2 reduced (README.md:118-124).
// MODE 2: G and the per-voxel norms |Dx| (zeros -> +inf: the reference's grad_norms, pytv/tv_GPU.py:88,135-139).
// XLD: the rows just above / below a wave's strip come from MEMORY (two more loads per frame, L1 / L2 hits: the neighbouring
// wave requests the same lines) instead of the LDS hand-off xe -- half the LDS, which is what lets the fp64 instantiation
// (R = 2 rows of doubles: the same registers as R = 4 floats) keep two blocks per CU.
// NWX: wave columns per block.  The memory pattern, not the arithmetic, bounds this kernel (a build that stores x instead of G and
// drops all the mathematics takes the same 1.55 ms at 64x8x1024x1024, profiles/r3_subgrad_col_pattern.txt): 256-byte row
// segments at a 4 KiB stride are what HBM likes least.  NWX = 2 puts two 64-column wave tiles side by side in ONE block (8 waves,
// one block per CU: the same two waves per SIMD): their rows are 496 contiguous bytes, requested within a frame of each other.
// AL (round 5): ALIGNED tiles.  A wave owns 64 columns that start on a multiple of 64 (whole 128-byte lines) and STORES all of them; the
// column ring is not a pair of idle lanes per side any more but a RING SLOT: once per plane the 64 lanes of the wave are re-read as
// (side, row of the strip, frame) and compute, from a few 4-byte gathers, x and 1 / |Dx| (or the finished product) of the column just
// left / right of the strip for all R rows and M frames at once; the values reach lane 0 / lane 63 of the frame loop through LDS
// and the "old" operand of the DPP moves.  Why: profiles/r4_sgpattern.txt (a partly written line costs 2.7 whole ones; the tile
// without its column ring runs in 0.99 instead of 1.77 ms) and profiles/r5_sgpattern_ring_slot.txt (the same with the gathers: 1.14 ms).
template <int S, typename T, int M, int R, int NW, int MODE, bool TWIN, bool XLD = false, int NWX = 1, bool AL = false>
struct SgCol {
    static constexpr bool CEN = (S == CENTRAL);
    static constexpr bool UP = (S == UPWIND || S == HYBRID), DN = (S == DOWNWIND || S == HYBRID);
    static constexpr bool HALO = (S == HYBRID || CEN);     // the norm of a ring row looks at the row outside the tile
    // ring of the tile: one row on either side (the first / last wave reads the row outside the tile from memory for the
    // schemes whose norm looks both ways); columns: a lane's norm needs x of BOTH neighbouring lanes for hybrid / central, so
    // x is valid on lanes 0..63, 1/|Dx| on 1..62 and G on 2..61; upwind / downwind look one way only: G on 1..62
    static constexpr int RING = HALO ? 2 : 1;
    static constexpr int RINGL = AL ? 0 : RING;                // ring LANES per side (AL: none)
    static constexpr int RB = R * NW, UR = RB - 2, UC = 64 - 2 * RINGL;
    static_assert(!AL || (!TWIN && 2 * R * M <= 64 && R >= 2), "ring slot: lane = (side, row, frame)");
    static constexpr int QS = 2 * 2 * R;                       // ring-slot hand-off, elements per frame: [side][quantity][row]
    static constexpr int H = (SG2_D < M) ? SG2_D : M;          // frames of the next plane requested before a step starts
    static constexpr bool HEADS = (M % SG2_D != 0);            // M a multiple of the ring depth: the ring slot of a frame never changes, no staging registers
    static constexpr int ROWS = 2 * NW + 1, ZR = 2 * NW;       // hand-off rows per (parity, frame): two per wave + one row of zeros
    using C = Col<T, R>;

    struct Shared {
        // [parity][frame][row][lane]; row 2 w = first row of wave w's strip, 2 w + 1 = its last row, row ZR = zeros (what the
        // first / last wave of the block reads instead of a neighbour: no select, no multiplier)
        T xe[NWX][XLD ? 1 : 2][XLD ? 1 : M][XLD ? 1 : ROWS][64];       // [wave column] x of plane z (parity z & 1); XLD: unused
        T ye[NWX][2][M][ROWS][64];  // row products of step z: [2 w] for the wave above, [2 w + 1] for the wave below
        // AL: what the ring slot of a wave hands to its frame loop: x of the ring column (quantity 0) and its product / 1 / |Dx|
        // (quantity 1), per frame, side and row; qz: zeros of the same shape (what lanes 1 .. 62 read), qdump: where the idle ring lanes write
        alignas(16) T qb[AL ? NW * NWX : 1][AL ? M : 1][2][2][R];
        alignas(16) T qz[AL ? M : 1][2][2][R];
        T qdump[AL ? 64 + R : 1];
        double sm[16];
    };

    // FAST: the tile (ring included) lies strictly inside the frame and there is no per-pixel time factor: no border
    // multipliers.  Everything else takes the generic variant (same memory operations, masks on the differences).
    // BK (round 3, late): 0 = FAST; 1 = COLUMN border -- the tile's rows (ring and halo rows included) lie strictly inside the frame,
    // its columns do not, and there is no per-pixel / per-voxel time factor: only the column multipliers survive (one value per
    // lane, the same for all R rows), the time channel runs as in FAST.  2 = generic.  On a 1024 x 1024 frame 148 of the 162 outline
    // blocks are column-border blocks; the generic variant costs 27 % more than FAST (TV_SPARE=7 forces it everywhere: 8.0 ->
    // 10.2 ms per descent step on 256x8x1024x1024), which made the outline 9 % of the kernel.
    template <int BK>
    static __device__ __forceinline__ void run(const DG& g, const WT<T>& w, const T* __restrict__ x, const T* __restrict__ xp,
                                               const T* __restrict__ xn, T* __restrict__ G, int zchunk, int chunk, int tile_x, int tile_y,
                                               int win, long long lid, double* __restrict__ partials, const SgArgs2<T>& sa, Shared& sh) {
        constexpr bool FAST = (BK == 0);         // no border multipliers at all
        constexpr bool ROWS_IN = (BK <= 1);      // every row this thread touches exists and has both neighbours
        constexpr bool TFAST = (BK <= 1);        // uniform time factor: the weighted time difference is carried from frame to frame
        const int lane = (int)threadIdx.x, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.y);    // the wave index is uniform: keep it scalar
        const int wv = wid % NW, wx = wid / NW;                 // position in the block: row strip, wave column
        tile_x = tile_x * NWX + wx;
        auto& xe = sh.xe[wx];
        auto& ye = sh.ye[wx];
        const int Mg = TWIN ? g.m : M;
        const int t0 = TWIN ? win * SG2_TWU - 1 : 0;
        auto fvalid = [&](int t) { return !TWIN || (t0 + t >= 0 && t0 + t < Mg); };
        auto fstore = [&](int t) { return !TWIN || (t0 + t >= win * SG2_TWU && t0 + t < win * SG2_TWU + SG2_TWU && t0 + t < Mg); };
        auto foff_t = [&](int t) { return (long long)(t0 + t) * g.s_t; };     // uniform
        const int fbytes = (int)(g.s_t * (long long)sizeof(T));
#if TV_SG2_WIDE
        const int cx = (tile_x * NW + wv) * 64 + lane;
        const int yb = tile_y * R;
#else
        const int cx = tile_x * UC - RINGL + lane;
        const int yb = tile_y * UR - 1 + wv * R;               // first row of this wave's strip
#endif
        const bool in_x = FAST || (cx >= 0 && cx < g.nx);
        // per-lane byte offsets inside a frame: roff = where this thread's sites are (out of range if outside the frame: loads
        // give 0), soff = where it STORES (only sites the tile owns: not the ring, not outside the frame)
        unsigned roff[R], soff[R];
        C mi, mfr, mbr, mfc, mbc, mft, cm;
        const bool lane_ok = in_x && (TV_SG2_WIDE || (lane >= RINGL && lane <= 63 - RINGL));

#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int y = yb + i;
            const bool in = in_x && (ROWS_IN || (y >= 0 && y < g.ny));
            const bool own = in && lane_ok && (TV_SG2_WIDE || (!(wv == 0 && i == 0) && !(wv == NW - 1 && i == R - 1)));
            const unsigned off = (unsigned)(((long long)y * g.rp + cx) * (long long)sizeof(T));
            roff[i] = in ? off : SG2_OOB;
            soff[i] = own ? off : SG2_OOB;
#if TV_SG2_STALIGN        // EXPERIMENT (wrong results): every lane stores, at 64-column-aligned positions -- what whole-line stores would be worth
            {
                const int ca = tile_x * 64 + lane;
                soff[i] = (y >= 0 && y < g.ny && ca < g.rp) ? (unsigned)(((long long)y * g.rp + ca) * (long long)sizeof(T)) : SG2_OOB;
            }
#endif
#if TV_SG2_LDALIGN        // EXPERIMENT (wrong results): the loads at 64-column-aligned positions
            {
                const int ca = tile_x * 64 + lane;
                roff[i] = (y >= 0 && y < g.ny && ca < g.rp) ? (unsigned)(((long long)y * g.rp + ca) * (long long)sizeof(T)) : SG2_OOB;
            }
#endif
            // AL: every lane of the strip owns its columns, and a lane outside the frame holds zeros (sum of squares 0 -> norm 0 by the
            // zero-gradient rule): the mask of the TV sum is the wave-uniform "not a ring row" (scalar registers instead of R vector ones)
            cm.v[i] = AL ? ((!(wv == 0 && i == 0) && !(wv == NW - 1 && i == R - 1)) ? T(1) : T(0)) : (own ? T(1) : T(0));
            // border multipliers of the generic variant: mi: site exists; mfr / mbr: it has a next / previous row; mfc / mbc:
            // a next / previous column (central: both, for a difference to exist)
            const bool hn = in && (ROWS_IN || y + 1 < g.ny), hp = in && (ROWS_IN || y > 0), cn = in && (cx + 1 < g.nx), cp = in && (cx > 0);
            mi.v[i] = in ? T(1) : T(0);
            mfr.v[i] = (CEN ? (hn && hp) : hn) ? T(1) : T(0);
            mbr.v[i] = (CEN ? (hn && hp) : hp) ? T(1) : T(0);
            mfc.v[i] = (CEN ? (cn && cp) : cn) ? T(1) : T(0);
            mbc.v[i] = (CEN ? (cn && cp) : cp) ? T(1) : T(0);
            mft.v[i] = T(0);
            if (!TFAST && g.ta && in) mft.v[i] = w.wt * mask_factor1<T>(g, w.sf, y, cx);
        }
        const int zs = chunk * zchunk;
        const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
        const T s = (S == HYBRID) ? Consts<T>::inv_sqrt2() : (CEN ? T(0.5) : T(1));
        const T thr = tiny_sumsq<T>() / (s * s);               // zero-gradient rule on the UNSCALED sum of squares: s^2 ss < tiny
        const T a_step = (MODE == 1) ? sa.step * sa.lambda * s : T(0);
        const T kap = (MODE == 1) ? -(T(1) - sa.step) / a_step : T(0);
        const T wt_u = g.ta ? w.wt : T(0);                     // FAST: uniform time weight
        const T wz_u = g.za ? w.wz : T(0);
        // the row just outside the tile (hybrid / central: the norm of a ring row needs it): read from memory by the first /
        // last wave; every other wave requests an out-of-range offset and gets 0
        unsigned hoff = SG2_OOB;
        if (HALO && in_x && !XLD) {
            if (wv == 0 && yb > 0) hoff = (unsigned)(((long long)(yb - 1) * g.rp + cx) * (long long)sizeof(T));
            if (wv == NW - 1 && yb + R < g.ny) hoff = (unsigned)(((long long)(yb + R) * g.rp + cx) * (long long)sizeof(T));
        }
        // XLD: EVERY wave reads the row above / below its strip (where the scheme needs it and the row exists: the block's first /
        // last wave only for the schemes whose ring norm looks outwards, exactly like the halo load above)
        unsigned uoff = SG2_OOB, doff = SG2_OOB;
        if (XLD && in_x) {
            if ((DN || CEN) && yb > 0 && (wv > 0 || HALO)) uoff = (unsigned)(((long long)(yb - 1) * g.rp + cx) * (long long)sizeof(T));
            if ((UP || CEN) && yb + R < g.ny && (wv < NW - 1 || HALO)) doff = (unsigned)(((long long)(yb + R) * g.rp + cx) * (long long)sizeof(T));
        }
        // ---- AL: the ring slot ----------------------------------------------------------------------------------------------------
        // lane = (side, row of the strip, frame): side 0 = the column LEFT of the strip (c0 - 1), side 1 = the column RIGHT of it
        // (c0 + 64); 8 lanes per row (frames), 32 per side.  What each scheme needs there: upwind the finished product f_c / |Dx| of
        // the left column (and x of the right one), downwind the mirror image, central the product on both sides, hybrid the left
        // product and 1 / |Dx| of the right column -- always x of the ring column itself.  Inputs: 4-byte gathers (one descriptor per
        // plane, the frame offset rides in the lane offset), requested a whole plane step ahead: r?n = plane zl + 1, r? = plane zl.
        constexpr bool R_NEED_O = AL && HALO;                    // the column beyond the ring column
        constexpr bool R_NEED_UP = AL && (DN || CEN), R_NEED_DN = AL && (UP || CEN);
        const int r_side = lane >> 5, r_i = (lane >> 3) & 3, r_t = lane & 7;
        const bool r_lane = AL && r_i < R && r_t < M;
        const int r_c0 = tile_x * 64, r_dir = r_side ? 1 : -1;
        const int r_c = r_side ? r_c0 + 64 : r_c0 - 1, r_y = yb + r_i;
        const bool r_in = r_lane && r_c >= 0 && r_c < g.nx && (ROWS_IN || (r_y >= 0 && r_y < g.ny));
        // the side whose 1 / |Dx| is needed (the other side only hands x over)
        const bool r_norm = r_in && (HALO || (UP && r_side == 0) || (DN && S != HYBRID && r_side == 1));
        unsigned go_c = SG2_OOB, go_o = SG2_OOB, go_i = SG2_OOB, go_h = SG2_OOB;
        T r_mi = T(0), r_mfr = T(0), r_mbr = T(0), r_mfc = T(0), r_mbc = T(0), r_mtn = T(0), r_mtp = T(0), r_wt = T(0);
        if constexpr (AL) {
            const unsigned fo = (unsigned)r_t * (unsigned)fbytes;
            auto off = [&](int y, int c) -> unsigned {
                return (c >= 0 && c < g.nx && y >= 0 && y < g.ny) ? fo + (unsigned)(((long long)y * g.rp + c) * (long long)sizeof(T)) : SG2_OOB;
            };
            if (r_in) go_c = off(r_y, r_c);
            if (r_norm) {
                if (R_NEED_O) go_o = off(r_y, r_c + r_dir);
                go_i = off(r_y, r_c - r_dir);
                if (R_NEED_UP && r_i == 0) go_h = off(r_y - 1, r_c);
                if (R_NEED_DN && r_i == R - 1) go_h = off(r_y + 1, r_c);
            }
            const bool hn = r_in && (ROWS_IN || r_y + 1 < g.ny), hp = r_in && (ROWS_IN || r_y > 0), cn = r_in && r_c + 1 < g.nx, cp = r_in && r_c > 0;
            r_mi = r_in ? T(1) : T(0);
            r_mfr = (CEN ? (hn && hp) : hn) ? T(1) : T(0);
            r_mbr = (CEN ? (hn && hp) : hp) ? T(1) : T(0);
            r_mfc = (CEN ? (cn && cp) : cn) ? T(1) : T(0);
            r_mbc = (CEN ? (cn && cp) : cp) ? T(1) : T(0);
            const bool tn = r_in && (r_t + 1 < M), tp = r_in && (r_t > 0);
            r_mtn = (CEN ? (tn && tp) : tn) ? T(1) : T(0);
            r_mtp = (CEN ? (tn && tp) : tp) ? T(1) : T(0);
            if (g.ta && r_in) r_wt = w.wt * ((g.mask != nullptr || g.tf != nullptr) ? mask_factor1<T>(g, w.sf, r_y, r_c) : T(1));
        }
        // state: rc = x(zl) of the ring site, rcn = x(zl + 1) (needed a plane early for the z difference), ro / ri / rh = the outer /
        // inner column and the halo row of plane zl -- requested one step ahead, together with the centre of the plane after
        T rc = T(0), ro = T(0), ri = T(0), rh = T(0), rcn = T(0), rpz = T(0);
        auto ring_ld_c = [&](const T* plane, T& c_) { c_ = sg2_ld(sg2_rsrc<T>(plane, plane != nullptr, fbytes * M), go_c, T(0)); };
        auto ring_ld_n = [&](const T* plane, T& o_, T& i_, T& h_) {
            const Rsrc rs = sg2_rsrc<T>(plane, plane != nullptr, fbytes * M);
            if (R_NEED_O) o_ = sg2_ld(rs, go_o, T(0));
            i_ = sg2_ld(rs, go_i, T(0));
            h_ = sg2_ld(rs, go_h, T(0));
        };
        // where the ring lanes put their results and where the lanes of the frame loop find them (lanes 1 .. 62: zeros)
        T* const qw = r_lane ? &sh.qb[AL ? wid : 0][AL ? r_t : 0][r_side][0][r_i] : &sh.qdump[AL ? lane : 0];
        const T* const qr = (lane == 0) ? &sh.qb[AL ? wid : 0][0][0][0][0] : ((lane == 63) ? &sh.qb[AL ? wid : 0][0][1][0][0] : &sh.qz[0][0][0][0]);
        if (AL && wid == 0) {
            for (int k = lane; k < M * QS; k += 64) (&sh.qz[0][0][0][0])[k] = T(0);
        }
        // LDS hand-off rows: own pair, the neighbour's row above / below (the zero row at the block's ends)
        const int r_own = 2 * wv, r_up = (wv > 0) ? 2 * (wv - 1) + 1 : ZR, r_dn = (wv < NW - 1) ? 2 * (wv + 1) : ZR;
        if (wv == 0) {
#pragma unroll
            for (int t = 0; t < M; ++t) {
                if (!XLD) xe[0][t][ZR][lane] = xe[1][t][ZR][lane] = T(0);
                ye[0][t][ZR][lane] = ye[1][t][ZR][lane] = T(0);
            }
        }
#pragma unroll
        for (int t = 0; t < M; ++t)        // the first step reads the "previous step's" products for a store that is dropped: keep them finite
            ye[0][t][r_own][lane] = ye[0][t][r_own + 1][lane] = ye[1][t][r_own][lane] = ye[1][t][r_own + 1][lane] = T(0);
        static_assert(NW >= 2, "one halo load per thread serves the first OR the last wave");
        const T m_hu = (wv == 0) ? T(1) : T(0), m_hd = (wv == NW - 1) ? T(1) : T(0);
        double acc = 0.0, acc_fid = 0.0;

        // frame t of the window inside the plane that starts at `plane0` (frame 0 of the VOLUME): see SgFrame
        auto fr = [&](const T* plane0, bool valid, int t) -> SgFrame {
            if constexpr (TWIN || !TV_SG2_PLANE_DESC) return SgFrame{sg2_rsrc<T>(plane0 + foff_t(t), valid, fbytes), 0};
            else return SgFrame{sg2_rsrc<T>(plane0, valid, fbytes * M), t * fbytes};
        };
        auto frame = [&](const T* plane, int t) { return fr(plane, plane != nullptr && fvalid(t), t); };
        auto load_rows = [&](const SgFrame& r, C& o) {
#pragma unroll
            for (int i = 0; i < R; ++i) o.v[i] = sg2_ld(r, roff[i], T(0));
        };

        // per-frame state carried from plane to plane: x(z); the accumulators of G(z-1) and G(z); and ONE more array only where
        // the scheme looks backwards in z -- central: x(z-1); downwind / hybrid: the weighted forward z difference of plane z-1,
        // which IS the backward difference of plane z (upwind carries nothing: 96 registers of state at M = 8 instead of 128)
        constexpr bool PZ = DN || CEN;
        C Cc[M], Pz[PZ ? M : 1], Gp[M], Gc[M], Nq[SG2_D], Nh[HEADS ? H : 1];
        const int z_lo = g.za ? zs - 1 : zs;
        {
            const T* pp = g.za ? zplane<T>(g, x, xp, xn, 2, z_lo - 1) : nullptr;
            const T* pc = zplane<T>(g, x, xp, xn, 2, z_lo);
            const int gz0 = g.z0 + z_lo;
            const bool zp0 = (gz0 > 0) && (gz0 < g.nzg) && g.za;       // planes z_lo - 1 and z_lo both exist
#pragma unroll
            for (int t = 0; t < M; ++t) {
                load_rows(frame(pc, t), Cc[t]);
                if (PZ) {
                    C pv;
                    load_rows(frame(pp, t), pv);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        if (CEN) Pz[t].v[i] = pv.v[i];
                        else Pz[t].v[i] = (zp0 ? wz_u : T(0)) * (Cc[t].v[i] - pv.v[i]) * (FAST ? T(1) : mi.v[i]);
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i) Gp[t].v[i] = Gc[t].v[i] = T(0);
                if (!XLD) {
                    xe[z_lo & 1][t][r_own][lane] = Cc[t].v[0];
                    xe[z_lo & 1][t][r_own + 1][lane] = Cc[t].v[R - 1];
                }
            }
        }
        auto next_plane = [&](int zl) -> const T* {                 // plane zl+1 if this chunk needs it
            const T* pn = zplane<T>(g, x, xp, xn, 2, zl + 1);
            return (pn != nullptr && (g.za || zl + 1 < ze)) ? pn : nullptr;
        };
        {
            const T* pn = next_plane(z_lo);
#pragma unroll
            for (int d = 0; d < H; ++d) load_rows(frame(pn, d), HEADS ? Nh[HEADS ? d : 0] : Nq[d]);
        }
        if constexpr (AL) {
            // plane z_lo (current), z_lo + 1 (next); the z carry of the ring column like Pz of the strip
            const T* pp = g.za ? zplane<T>(g, x, xp, xn, 2, z_lo - 1) : nullptr;
            const T* pc = zplane<T>(g, x, xp, xn, 2, z_lo);
            ring_ld_c(pc, rc);
            ring_ld_n(pc, ro, ri, rh);
            if (PZ) {
                const Rsrc rs = sg2_rsrc<T>(pp, pp != nullptr, fbytes * M);
                const T pv = sg2_ld(rs, go_c, T(0));
                const int gz0 = g.z0 + z_lo;
                const bool zp0 = (gz0 > 0) && (gz0 < g.nzg) && g.za;
                rpz = CEN ? pv : (zp0 ? wz_u : T(0)) * (rc - pv) * r_mi;
            }
            ring_ld_c(next_plane(z_lo), rcn);
        }
        __syncthreads();
        // plane zl+1 exists in memory for this call iff it is a local plane or a supplied halo plane: same answer as zplane(),
        // but for the CURRENT plane of the next step it was already computed as this step's "next" unless the chunk did not need it
        auto zplane_cheap = [&](int zl) -> const T* { return zplane<T>(g, x, xp, xn, 2, zl); };
        const T* pc_c = zplane<T>(g, x, xp, xn, 2, z_lo);
        const T* pn_c = next_plane(z_lo);

#ifdef TV_SG2_TIMELINE
        // variant builds only (tools/sg_timeline.py): lane 0 of every wave of the first 256 blocks records the shader clock at the top of
        // every plane step, before the LDS barrier that ends it and after that barrier -- behind the THIRD partial array of the workspace
        // (unused by this kernel): [block][wave][step][4]; slot 3: the 100 MHz wall clock at the top of the step
        unsigned long long* tl_base = nullptr;
        if (lid < 256 && lane == 0 && sa.part_fid != nullptr) tl_base = reinterpret_cast<unsigned long long*>(sa.part_fid) + ((lid * (NW * NWX) + wid) * 40ll) * 4;
        int tl_k = 0;
#define TV_SG2_MARK(j) do { if (tl_base != nullptr && tl_k < 40) { tl_base[tl_k * 4 + (j)] = (unsigned long long)clock64(); if ((j) == 0) tl_base[tl_k * 4 + 3] = (unsigned long long)wall_clock64(); } } while (0)
#else
#define TV_SG2_MARK(j) do { } while (0)
#endif
#ifdef TV_SG2_PRIO_STATIC
        if (wid >= (NW * NWX) / 2) __builtin_amdgcn_s_setprio(TV_SG2_PRIO_STATIC);       // EXPERIMENT: the younger wave of every SIMD pair always first
#endif
        for (int zl = z_lo; zl <= ze; ++zl) {
            TV_SG2_MARK(0);
            const int gz = g.z0 + zl, par = zl & 1;
            const bool plane_in = (gz >= 0) && (gz < g.nzg) && (zl < ze || g.za);
            const T* pc = pc_c;                                            // carried: one zplane() per step instead of three
            const T* pn = pn_c;
            const T* pn2 = (zl + 1 <= ze) ? next_plane(zl + 1) : nullptr;
            pc_c = zplane_cheap(zl + 1);
            pn_c = pn2;
            const bool z_prev = plane_in && (gz > 0), z_next = plane_in && (gz + 1 < g.nzg);
            const T wzn = (CEN ? (z_prev && z_next) : z_next) ? wz_u : T(0), wzp = (CEN ? (z_prev && z_next) : z_prev) ? wz_u : T(0);
            const T m_pl = plane_in ? T(1) : T(0);
            const bool count = plane_in && (zl >= zs) && (zl < ze);
            const bool store = (zl - 1 >= zs) && (zl - 1 < ze);
            T hq[2] = {T(0), T(0)}, hu[2] = {T(0), T(0)}, hd[2] = {T(0), T(0)};
            if (HALO && !XLD) {
                hq[0] = sg2_ld(frame(pc, 0), hoff, T(0));
                if (M > 1) hq[1] = sg2_ld(frame(pc, 1), hoff, T(0));
            }
            if (XLD) {
#pragma unroll
                for (int k = 0; k < 2 && k < M; ++k) {
                    if (DN || CEN) hu[k] = sg2_ld(frame(pc, k), uoff, T(0));
                    if (UP || CEN) hd[k] = sg2_ld(frame(pc, k), doff, T(0));
                }
            }
            C x0q;                                  // MODE 1: x0 of the frame about to be stored, requested a frame ahead
            if (MODE == 1 && TV_SG2_X0_AHEAD) {
                const SgFrame r0 = fr(sa.x0 + (long long)(zl - 1) * g.s_z, store && fstore(0), 0);
#pragma unroll
                for (int i = 0; i < R; ++i) x0q.v[i] = sg2_ld_s(r0, soff[i], T(0));
            }
            if (HEADS) {
#pragma unroll
                for (int d = 0; d < H; ++d) Nq[d] = Nh[HEADS ? d : 0];
            }
            if constexpr (AL) {
                // ---- ring slot of plane zl: the same differences, masks and zero-gradient rule as a site of the strip, one site per lane
                T x_up = T(0), x_dn = T(0);
                if (R_NEED_UP) { const T sh_ = __shfl_up(rc, 8, 64); x_up = (r_i == 0) ? rh : sh_; }
                if (R_NEED_DN) { const T sh_ = __shfl_down(rc, 8, 64); x_dn = (r_i == R - 1) ? rh : sh_; }
                const T x_l = r_side ? ri : ro, x_r = r_side ? ro : ri;           // x(r_c - 1), x(r_c + 1)
                const T xtn = from_right(rc), xtp = from_left(rc);                // frame t + 1 / t - 1: the neighbouring lanes
                const T wti = r_wt * m_pl;
                T ss_r, fc_r = T(0), bc_r = T(0), fz_r;
                if (CEN) {
                    const T fr = (x_dn - x_up) * (r_mfr * m_pl);
                    fc_r = (x_r - x_l) * (r_mfc * m_pl);
                    fz_r = wzn * (rcn - rpz) * r_mi;
                    const T ft = (r_mtn * wti) * (xtn - xtp);
                    T a_ = fr * fr + fc_r * fc_r;
                    a_ = a_ + fz_r * fz_r;
                    a_ = a_ + ft * ft;
                    ss_r = a_;
                } else {
                    const T fr = (x_dn - rc) * (r_mfr * m_pl), br = (rc - x_up) * (r_mbr * m_pl);
                    fc_r = (x_r - rc) * (r_mfc * m_pl);
                    bc_r = (rc - x_l) * (r_mbc * m_pl);
                    fz_r = wzn * (rcn - rc) * r_mi;
                    const T bz = DN ? rpz : T(0);
                    const T ft = (r_mtn * wti) * (xtn - rc), bt = (r_mtp * wti) * (rc - xtp);
                    T a_ = fr * fr + fc_r * fc_r;
                    a_ = a_ + fz_r * fz_r;
                    a_ = a_ + ft * ft;
                    T b_ = br * br + bc_r * bc_r;
                    b_ = b_ + bz * bz;
                    b_ = b_ + bt * bt;
                    ss_r = (S == HYBRID) ? a_ + b_ : ((S == DOWNWIND) ? b_ : a_);
                }
                const T nn = (ss_r >= thr) ? rsq_fast(ss_r) : T(0);
                // quantity 1: what the edge lane of the strip adds -- see the scatter code of each scheme below
                T q2;
                if (CEN) q2 = fc_r * nn;
                else if (S == HYBRID) q2 = r_side ? nn : fc_r * nn;
                else if (UP) q2 = r_side ? T(0) : fc_r * nn;
                else q2 = r_side ? bc_r * nn : T(0);
                qw[0] = rc;
                qw[R] = q2;
                // carry, rotate, request plane zl + 2
                if (CEN) rpz = rc; else if (DN) rpz = fz_r;
                rc = rcn;
                ring_ld_n(pn, ro, ri, rh);           // plane zl + 1: consumed at the top of the next step
                ring_ld_c(pn2, rcn);                 // plane zl + 2
            }
            T xu_n = T(0), xd_n = T(0), yu_n = T(0), yd_n = T(0);
            if ((DN || CEN) && !XLD) xu_n = xe[par][0][r_up][lane];
            if ((UP || CEN) && !XLD) xd_n = xe[par][0][r_dn][lane];
            if (UP || CEN) yu_n = ye[par ^ 1][0][r_up][lane];
            if (DN || CEN) yd_n = ye[par ^ 1][0][r_dn][lane];
            C q1_n, q2_n;                           // AL: ring values of the next frame (quantity 0: x, quantity 1: product / 1 / |Dx|)
            if constexpr (AL) {
                q1_n = *reinterpret_cast<const C*>(qr);
                q2_n = *reinterpret_cast<const C*>(qr + R);
            }
            C pf_t_prev, f_t_prev, c_old_prev;      // time-axis carries: product / forward difference / x of frame t-1
#pragma unroll
            for (int i = 0; i < R; ++i) pf_t_prev.v[i] = f_t_prev.v[i] = c_old_prev.v[i] = T(0);
#pragma unroll
            for (int t = 0; t < M; ++t) {
#if TV_SG2_PRIO
                // the two waves of a SIMD leapfrog: each has the higher issue priority in every other frame.  Without it the older wave of the
                // pair (waves 0 .. NW - 1) wins every arbitration, finishes a plane step ~3 k cycles early and sits in the barrier while the
                // younger one finishes alone at a single wave's issue rate (profiles/r6_sg_timeline.txt)
                if (((t + (wid >= (NW * NWX) / 2 ? 1 : 0)) & 1) != 0) __builtin_amdgcn_s_setprio(TV_SG2_PRIO);
                else __builtin_amdgcn_s_setprio(0);
#endif
                const C c = Cc[t];
                const C nx = Nq[t % SG2_D];          // x(zl+1, t)
                C q1, q2;
                if constexpr (AL) {
                    q1 = q1_n;
                    q2 = q2_n;
                    if (t + 1 < M) {
                        q1_n = *reinterpret_cast<const C*>(qr + (t + 1) * QS);
                        q2_n = *reinterpret_cast<const C*>(qr + (t + 1) * QS + R);
                    }
                }
                // ---- row neighbours across the strip's ends -----------------------------------------------------------
                T xu = xu_n, xd = xd_n;              // read from LDS one frame ahead (two waves per SIMD do not hide an LDS round trip)
                if (t + 1 < M && !XLD) {
                    if (DN || CEN) xu_n = xe[par][(t + 1 < M) ? t + 1 : t][r_up][lane];
                    if (UP || CEN) xd_n = xe[par][(t + 1 < M) ? t + 1 : t][r_dn][lane];
                }
                if (XLD) {                           // from memory, requested two frames ahead
                    xu = hu[t & 1];
                    xd = hd[t & 1];
                    if (t + 2 < M) {
                        if (DN || CEN) hu[t & 1] = sg2_ld(frame(pc, t + 2), uoff, T(0));
                        if (UP || CEN) hd[t & 1] = sg2_ld(frame(pc, t + 2), doff, T(0));
                    }
                }
                if (HALO && !XLD) {                          // first / last wave: LDS gave 0, the halo load the value; every other wave: the reverse
                    const T h = hq[t & 1];
                    if (t + 2 < M) hq[t & 1] = sg2_ld(frame(pc, t + 2), hoff, T(0));
                    xu += h * m_hu;
                    xd += h * m_hd;
                }
                C x0n;
                if (MODE == 1 && TV_SG2_X0_AHEAD && t + 1 < M) {
                    const SgFrame r0 = fr(sa.x0 + (long long)(zl - 1) * g.s_z, store && fstore(t + 1), t + 1);
#pragma unroll
                    for (int i = 0; i < R; ++i) x0n.v[i] = sg2_ld(r0, soff[i], T(0));
                }
                // publish the strip ends of plane zl+1 for the next step
                if (!XLD) {
                    xe[par ^ 1][t][r_own][lane] = nx.v[0];
                    xe[par ^ 1][t][r_own + 1][lane] = nx.v[R - 1];
                }
                // ---- raw differences --------------------------------------------------------------------------------------
                // dr[i] = x(row i+1) - x(row i) for i = -1 .. R-1 (index shifted by one): the forward row differences of the
                // strip and of the row above it
                T dr[R + 1];
                dr[0] = c.v[0] - xu;
#pragma unroll
                for (int i = 0; i + 1 < R; ++i) dr[i + 1] = c.v[i + 1] - c.v[i];
                dr[R] = xd - c.v[R - 1];
                C xr, xl;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if constexpr (AL) {
                        xr.v[i] = from_right(c.v[i], q1.v[i]);      // lane 63: x of the column right of the strip
                        xl.v[i] = from_left(c.v[i], q1.v[i]);       // lane 0: x of the column left of it
                    } else {
                        xr.v[i] = from_right(c.v[i]);
                        xl.v[i] = from_left(c.v[i]);
                    }
                }
                C f_r, b_r, f_c, b_c, f_z, b_z, f_t, b_t;     // weighted channels (central: f_* only)
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if (CEN) {
                        f_r.v[i] = ((i + 1 < R) ? c.v[(i + 1 < R) ? i + 1 : i] : xd) - ((i > 0) ? c.v[(i > 0) ? i - 1 : 0] : xu);   // x(i+1) - x(i-1), one rounding
                        f_c.v[i] = xr.v[i] - xl.v[i];
                        f_z.v[i] = wzn * (nx.v[i] - Pz[t].v[i]);
                        if (!FAST) { f_r.v[i] *= mfr.v[i] * m_pl; f_c.v[i] *= mfc.v[i] * m_pl; f_z.v[i] *= mi.v[i]; }
                        b_r.v[i] = b_c.v[i] = b_z.v[i] = T(0);
                    } else {
                        f_r.v[i] = dr[i + 1];
                        b_r.v[i] = dr[i];
                        f_c.v[i] = xr.v[i] - c.v[i];
                        b_c.v[i] = c.v[i] - xl.v[i];
                        f_z.v[i] = wzn * (nx.v[i] - c.v[i]);
                        b_z.v[i] = DN ? Pz[DN ? t : 0].v[i] : T(0);          // = the forward difference of plane zl-1, weights and masks included
                        if (!FAST) {
                            f_r.v[i] *= mfr.v[i] * m_pl; b_r.v[i] *= mbr.v[i] * m_pl;
                            f_c.v[i] *= mfc.v[i] * m_pl; b_c.v[i] *= mbc.v[i] * m_pl;
                            f_z.v[i] *= mi.v[i];
                        }
                    }
                }
                // time axis: forward difference to frame t+1 (of the OLD plane zl: Cc[t+1] has not been rotated yet)
                bool has_tn, has_tp;        // frame t+1 / t-1 exists in the volume
                if (!TWIN) { has_tn = (t + 1 < M); has_tp = (t > 0); }
                else { has_tn = fvalid(t) && (t0 + t + 1 < Mg); has_tp = fvalid(t) && (t0 + t > 0); }
                C xtn, xtp = c_old_prev;    // x(zl, t+1), x(zl, t-1)
                if (t + 1 < M) xtn = Cc[(t + 1 < M) ? t + 1 : t];
                else if (TWIN) load_rows(SgFrame{sg2_rsrc<T>(pc + foff_t(t + 1), pc != nullptr && has_tn, fbytes), 0}, xtn);
                else xtn = c;
                if (TWIN && t == 0 && (CEN || DN)) load_rows(SgFrame{sg2_rsrc<T>(pc + foff_t(-1), pc != nullptr && has_tp, fbytes), 0}, xtp);
                // per-VOXEL weight of the time channels (tv_geom::time_weight_vol; generic variant only -- with a weight volume no
                // tile is FAST): the factor of THIS voxel scales its forward AND its backward channel, so the backward one is no
                // longer the forward one of frame t-1 and the carry is the unweighted difference.  No volume: a descriptor with
                // num_records = 0, the loads cost nothing and the factor is 1.
                C wv_t;
                if (!TFAST) {
                    const T* wpl = vol_plane<T>(g, zl, t0 + t);          // uniform: the weight frame of (zl, t), ghost planes included
                    const Rsrc rw = sg2_rsrc<T>(wpl, wpl != nullptr, fbytes);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const T v = sg2_ld(rw, roff[i], T(0));
                        wv_t.v[i] = (wpl != nullptr) ? v : T(1);
                    }
                }
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const T wti = FAST ? wt_u : (TFAST ? wt_u * (mi.v[i] * m_pl) : (mft.v[i] * m_pl) * wv_t.v[i]);
                    if (CEN) {
                        f_t.v[i] = (has_tn && has_tp) ? wti * (xtn.v[i] - xtp.v[i]) : T(0);
                        b_t.v[i] = T(0);
                    } else if (TFAST) {
                        f_t.v[i] = has_tn ? wti * (xtn.v[i] - c.v[i]) : T(0);
                        b_t.v[i] = f_t_prev.v[i];
                        if (TWIN && t == 0 && DN) b_t.v[i] = has_tp ? wti * (c.v[i] - xtp.v[i]) : T(0);   // backward difference of the window's first frame
                    } else {
                        const T dt = has_tn ? (xtn.v[i] - c.v[i]) : T(0);
                        f_t.v[i] = wti * dt;
                        b_t.v[i] = wti * f_t_prev.v[i];                   // f_t_prev carries the RAW difference here
                        if (TWIN && t == 0 && DN) b_t.v[i] = has_tp ? wti * (c.v[i] - xtp.v[i]) : T(0);
                        f_t_prev.v[i] = dt;
                    }
                }
                if (TFAST || CEN)
                f_t_prev = f_t;
                c_old_prev = c;
                // ---- 1 / |Dx| -------------------------------------------------------------------------------------------
                C ss, n;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    T a = f_r.v[i] * f_r.v[i] + f_c.v[i] * f_c.v[i];
                    a = a + f_z.v[i] * f_z.v[i];
                    a = a + f_t.v[i] * f_t.v[i];
                    if (S == HYBRID || S == DOWNWIND) {
                        T b = b_r.v[i] * b_r.v[i] + b_c.v[i] * b_c.v[i];
                        b = b + b_z.v[i] * b_z.v[i];
                        b = b + b_t.v[i] * b_t.v[i];
                        a = (S == HYBRID) ? a + b : b;
                    }
                    ss.v[i] = a;
                }
                {
                    T mn = ss.v[0];
#pragma unroll
                    for (int i = 1; i < R; ++i) mn = (sizeof(T) == 4) ? (T)__builtin_fminf((float)mn, (float)ss.v[i]) : (T)__builtin_fmin((double)mn, (double)ss.v[i]);   // v_min: one instruction, not compare + select
                    // a wave that holds no vanishing gradient (the usual case) skips the selects
                    if (__builtin_expect(__any(!(mn >= thr)), 0)) {
#pragma unroll
                        for (int i = 0; i < R; ++i) n.v[i] = (ss.v[i] >= thr) ? rsq_fast(ss.v[i]) : T(0);
                    } else {
#pragma unroll
                        for (int i = 0; i < R; ++i) n.v[i] = rsq_fast(ss.v[i]);
                    }
                }
                // TV: |Dx| = s * ss * (1 / sqrt(ss)); sites the tile owns, planes this chunk owns (uniform multiplier, no branch)
                {
                    const bool cnt = count && fstore(t);
                    T sum = T(0);
                    SgFrame rn_rs{};
                    if (MODE == 2) rn_rs = fr(sa.norms + (long long)zl * g.s_z, cnt, t);
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const T rn = ss.v[i] * n.v[i];
                        sum += rn * cm.v[i];
                        if (MODE == 2) sg2_st(rn_rs, soff[i], (n.v[i] > T(0)) ? s * rn : (T)__builtin_inff());
                    }
                    if (!TV_SG2_COPYONLY) acc += (double)(sum * (cnt ? s : T(0)));
                    pin1<double>(acc);        // or LLVM sinks the sums of all M frames below the frame loop and keeps their operands alive
                }
                // ---- scatter the products -------------------------------------------------------------------------------
                C gc = Gc[t], gn;
                T to_up = T(0), to_dn = T(0);          // products handed to the wave above / below
                if (CEN) {
                    // one product per axis, +1/2 to the site after, -1/2 to the site before
#pragma unroll
                    for (int i = 0; i < R; ++i) {
                        const T p_r = f_r.v[i] * n.v[i];
                        if (i + 1 < R) gc.v[(i + 1 < R) ? i + 1 : i] += p_r; else to_dn = p_r;
                        if (i > 0) gc.v[(i > 0) ? i - 1 : 0] -= p_r; else to_up = p_r;
                        const T p_c = f_c.v[i] * n.v[i];
                        if constexpr (AL) gc.v[i] += from_left(p_c, q2.v[i]) - from_right(p_c, q2.v[i]);
                        else gc.v[i] += from_left(p_c) - from_right(p_c);
                        const T p_z = f_z.v[i] * n.v[i];
                        gn.v[i] = p_z;
                        Gp[t].v[i] -= p_z;
                        const T p_t = f_t.v[i] * n.v[i];
                        gc.v[i] += pf_t_prev.v[i];
                        pf_t_prev.v[i] = p_t;
                        if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i] -= p_t; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i]); }   // G(zl, t-1): rotated into Gp at the end of frame t-1
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < R; ++i) gn.v[i] = T(0);
                    if constexpr (S == HYBRID) {
                        // hybrid, FLUX form.  The backward channel of a site IS the forward channel of its predecessor, so an
                        // edge (v, v + e) with difference d hands d n(v) + d n(v + e) = d (n(v) + n(v + e)) to v + e and takes the
                        // same from v: one product per edge instead of two, and half the accumulations.  Used where both ends
                        // of the edge are at hand in the same frame step: the rows inside the strip, the columns (1 / |Dx| of the
                        // right neighbour: one DPP move) and -- FAST variant, where the time factor is the same for both ends --
                        // the frames (the edge t-1 -> t closes at frame t).  The strip's end rows and the planes stay scatters.
#pragma unroll
                        for (int i = 0; i + 1 < R; ++i) {
                            const T ph = f_r.v[i] * (n.v[i] + n.v[(i + 1 < R) ? i + 1 : i]);
                            gc.v[i] -= ph;
                            gc.v[(i + 1 < R) ? i + 1 : i] += ph;
                        }
                        to_up = b_r.v[0] * n.v[0];
                        gc.v[0] += to_up;
                        to_dn = f_r.v[R - 1] * n.v[R - 1];
                        gc.v[R - 1] -= to_dn;
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            if constexpr (AL) {
                                // lane 63: 1 / |Dx| of the column right of the strip; lane 0: the edge from the left column = its product
                                // f_c / |Dx| (ring slot) + the same difference over this lane's own norm
                                const T pc_ = f_c.v[i] * (n.v[i] + from_right(n.v[i], q2.v[i]));
                                const T e0 = b_c.v[i] * n.v[i] + q2.v[i];
                                gc.v[i] += from_left(pc_, e0) - pc_;
                            } else {
                            const T pc_ = f_c.v[i] * (n.v[i] + from_right(n.v[i]));
                            gc.v[i] += from_left(pc_) - pc_;
                            }
                            const T pz_f = f_z.v[i] * n.v[i], pz_b = b_z.v[i] * n.v[i];
                            gc.v[i] += pz_b - pz_f;
                            gn.v[i] = pz_f;
                            Gp[t].v[i] -= pz_b;
                            if (TFAST) {
                                T pt_;
                                if (TWIN && t == 0) pt_ = b_t.v[i] * n.v[i];          // the edge into the window: its other end is not ours
                                else pt_ = b_t.v[i] * (pf_t_prev.v[i] + n.v[i]);      // pf_t_prev carries 1 / |Dx| of frame t-1 here
                                gc.v[i] += pt_;
                                if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i] -= pt_; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i]); }
                                pf_t_prev.v[i] = n.v[i];
                            } else {
                                const T p_t = f_t.v[i] * n.v[i], q_t = b_t.v[i] * n.v[i];
                                gc.v[i] += (pf_t_prev.v[i] - p_t) + q_t;
                                pf_t_prev.v[i] = p_t;
                                if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i] -= q_t; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i]); }
                            }
                        }
                    } else
                    if (UP) {       // forward channels: - to the site itself, + to the next site
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const T p_r = f_r.v[i] * n.v[i];
                            gc.v[i] -= p_r;
                            if (i + 1 < R) gc.v[(i + 1 < R) ? i + 1 : i] += p_r; else to_dn = p_r;
                            const T p_c = f_c.v[i] * n.v[i];
                            if constexpr (AL) gc.v[i] += from_left(p_c, q2.v[i]) - p_c;
                            else gc.v[i] += from_left(p_c) - p_c;
                            const T p_z = f_z.v[i] * n.v[i];
                            gc.v[i] -= p_z;
                            gn.v[i] = p_z;
                            const T p_t = f_t.v[i] * n.v[i];
                            gc.v[i] += pf_t_prev.v[i] - p_t;
                            pf_t_prev.v[i] = p_t;
                        }
                    }
                    if (DN && S != HYBRID) {       // backward channels: + to the site itself, - to the previous site
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const T p_r = b_r.v[i] * n.v[i];
                            gc.v[i] += p_r;
                            if (i > 0) gc.v[(i > 0) ? i - 1 : 0] -= p_r; else to_up = p_r;
                            const T p_c = b_c.v[i] * n.v[i];
                            if constexpr (AL) gc.v[i] += p_c - from_right(p_c, q2.v[i]);
                            else gc.v[i] += p_c - from_right(p_c);
                            const T p_z = b_z.v[i] * n.v[i];
                            gc.v[i] += p_z;
                            Gp[t].v[i] -= p_z;
                            const T p_t = b_t.v[i] * n.v[i];
                            gc.v[i] += p_t;
                            if (t > 0) { Gp[(t > 0) ? t - 1 : 0].v[i] -= p_t; pin1<T>(Gp[(t > 0) ? t - 1 : 0].v[i]); }   // G(zl, t-1): rotated into Gp at the end of frame t-1
                        }
                    }
                }
                if (DN || CEN) ye[par][t][r_own][lane] = to_up;
                if (UP || CEN) ye[par][t][r_own + 1][lane] = to_dn;
#pragma unroll
                for (int i = 0; i < R; ++i) pin1<T>(gc.v[i]);
                if (MODE == 1) {
                    // the descent step needs x(zl) when G(zl) is complete, one step from now: instead of carrying x for a plane
                    // (32 registers at M = 8) its share of the step rides in the accumulator: x_out = step x0 - a (G + kap x),
                    // a = step lambda s, kap = -(1 - step) / a.  The accumulator then holds |kap x| ~ 10^3 next to G's O(1) terms:
                    // x_out is good to a few ulp of x, like the direct form (host: only for step * lambda >= 1e-6)
#pragma unroll
                    for (int i = 0; i < R; ++i) gc.v[i] += kap * c.v[i];
                }
                // ---- plane zl-1 is complete: store (sites the tile owns; everything else has an out-of-range offset) -----------
                {
                    const bool st = store && fstore(t);
                    const long long foff = (long long)(zl - 1) * g.s_z + foff_t(t);      // uniform
                    // the row products the neighbouring waves published in the previous step (zero row at the block's ends)
                    if (UP || CEN) Gp[t].v[0] += yu_n;
                    if (DN || CEN) Gp[t].v[R - 1] -= yd_n;
                    if (t + 1 < M) {                 // next frame's, a frame ahead
                        if (UP || CEN) yu_n = ye[par ^ 1][(t + 1 < M) ? t + 1 : t][r_up][lane];
                        if (DN || CEN) yd_n = ye[par ^ 1][(t + 1 < M) ? t + 1 : t][r_dn][lane];
                    }
                    if (MODE != 1) {
                        const SgFrame rg = fr(G + (long long)(zl - 1) * g.s_z, st, t);
#pragma unroll
                        for (int i = 0; i < R; ++i) sg2_st(rg, soff[i], TV_SG2_COPYONLY ? c.v[i] : s * Gp[t].v[i]);
                    } else {
                        const SgFrame ro = fr(sa.x_out + (long long)(zl - 1) * g.s_z, st, t);
                        if (!TV_SG2_X0_AHEAD) {
                            const SgFrame r0 = fr(sa.x0 + (long long)(zl - 1) * g.s_z, st, t);
#pragma unroll
                            for (int i = 0; i < R; ++i) x0q.v[i] = sg2_ld_s(r0, soff[i], T(0));
                        }
                        T e2 = T(0);
#pragma unroll
                        for (int i = 0; i < R; ++i) {
                            const T x0v = x0q.v[i];
                            const T xo = sa.step * x0v - a_step * Gp[t].v[i];
                            const T e = xo - x0v;
                            e2 += (e * e) * cm.v[i];
                            sg2_st(ro, soff[i], xo);
                        }
                        acc_fid += (double)(e2 * (st ? T(0.5) : T(0)));
                        pin1<double>(acc_fid);
                        if (TV_SG2_X0_AHEAD && t + 1 < M) x0q = x0n;
                    }
                }
                // rotate this frame: the finished slot carries the start of G(zl+1); x(zl) -> x(zl-1), x(zl+1) -> x(zl)
#pragma unroll
                for (int i = 0; i < R; ++i) pin1<T>(gn.v[i]);
                Gp[t] = gc;          // G(zl, t): still misses its z+1 term, the time term of frame t+1 and the cross-wave row terms
                Gc[t] = gn;          // the start of G(zl+1, t)
                if (CEN) Pz[CEN ? t : 0] = c;
                else if (DN) Pz[DN ? t : 0] = f_z;
                Cc[t] = nx;
                // next load of the ring: frame t + D of plane zl+1, or -- at the end of the step -- frame t + D - M of plane zl+2
                if (t + SG2_D < M) load_rows(frame(pn, t + SG2_D), Nq[t % SG2_D]);
                if (t >= M - H) load_rows(frame(pn2, t - (M - H)), HEADS ? Nh[(HEADS && t >= M - H) ? t - (M - H) : 0] : Nq[t % SG2_D]);
#if TV_SG2_SCHED_BARRIER
                __builtin_amdgcn_sched_barrier(0);   // keep the frames apart: interleaving them costs registers (scratch) for nothing
#endif
            }
            // LDS-only barrier: the hand-off rows must be visible, nothing else.  __syncthreads() would also drain vmcnt -- the
            // stores of this plane and the look-ahead loads of the next one, one exposed memory round trip per plane
            TV_SG2_MARK(1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            TV_SG2_MARK(2);
#ifdef TV_SG2_TIMELINE
            ++tl_k;
#endif
        }
        acc = block_sum(acc, sh.sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) partials[lid] = acc;
        if (MODE == 1) {
            acc_fid = block_sum(acc_fid, sh.sm);
            if (threadIdx.x == 0 && threadIdx.y == 0) sa.part_fid[lid] = acc_fid;
        }
    }
};

// Which tiles a launch works on.  The tiles whose 16 x 64 sites (ring included) lie strictly inside the frame form a rectangle
// [ix0, ix1] x [iy0, iy1] of the tile grid: they run the FAST instantiation (no border multipliers), everything else -- the
// frame's outline, or every tile when a per-pixel time factor is set -- the generic one.
struct SgTiles {
    int tx, ty;                  // tile grid
    int ix0, ix1, iy0, iy1;      // the interior rectangle (empty if ix1 < ix0 or iy1 < iy0)
    long long nfast, nborder;    // tiles inside / outside the rectangle
};
__device__ __forceinline__ void sg2_tile(const SgTiles& tm, bool fast, long long k, int& bx, int& by) {
    const int nxi = tm.ix1 - tm.ix0 + 1, nyi = tm.iy1 - tm.iy0 + 1;
    if (fast) {
        bx = tm.ix0 + (int)(k % nxi);
        by = tm.iy0 + (int)(k / nxi);
        return;
    }
    if (nxi <= 0 || nyi <= 0) {          // no interior: all tiles
        bx = (int)(k % tm.tx);
        by = (int)(k / tm.tx);
        return;
    }
    const long long top = (long long)tm.iy0 * tm.tx, wmid = tm.tx - nxi, mid = (long long)nyi * wmid;
    if (k < top) {
        bx = (int)(k % tm.tx);
        by = (int)(k / tm.tx);
    } else if (k < top + mid) {
        k -= top;
        const int c = (int)(k % wmid);
        by = tm.iy0 + (int)(k / wmid);
        bx = (c < tm.ix0) ? c : c + nxi;
    } else {
        k -= top + mid;
        bx = (int)(k % tm.tx);
        by = tm.iy1 + 1 + (int)(k / tm.tx);
    }
}

template <int S, typename T, int M, int R, int NW, int MODE, bool TWIN, bool XLD = false, int NWX = 1, bool AL = false>
__global__ __launch_bounds__(64 * NW * NWX, (NW * NWX >= 16) ? 1 : 2) void k_subgrad_col(DG g, WT<T> w, const T* __restrict__ x, const T* __restrict__ xp,
                                                                            const T* __restrict__ xn, T* __restrict__ G, int zchunk, int nchunks,
                                                                            double* __restrict__ partials, SgArgs2<T> sa, SgTiles tm) {
    using K = SgCol<S, T, M, R, NW, MODE, TWIN, XLD, NWX, AL>;
    __shared__ typename K::Shared sh;
    const int Mg = TWIN ? g.m : M;
    const int nwin = TWIN ? (Mg + SG2_TWU - 1) / SG2_TWU : 1;
    // ONE launch for both variants (two launches would serialise, and the outline tiles alone do not fill the GPU): logical
    // ids [0, nb_border) are the generic tiles -- first, they take longer -- then the interior ones.  Inside each range the
    // XCD-aware order (consecutive workgroup ids go round-robin to the 8 XCDs; neighbouring tiles share ring rows / columns):
    // logical id = (id % 8) * per_xcd + id / 8
    const long long per_tile = (long long)nchunks * nwin;
    const long long nb_border = tm.nborder * per_tile, nb_fast = tm.nfast * per_tile;
    const long long pb = (nb_border + 7) / 8 * 8;                  // grid: [border range padded to 8][interior range padded to 8]
    const bool fast = (long long)blockIdx.x >= pb;
    const long long id = fast ? (long long)blockIdx.x - pb : (long long)blockIdx.x;
    const long long total = fast ? nb_fast : nb_border, ntiles = fast ? tm.nfast : tm.nborder, per_xcd = (total + 7) / 8;
    const long long lid = (id % 8) * per_xcd + id / 8;
    if (lid >= total) return;
    const int win = (int)(lid / (ntiles * nchunks));
    const int chunk = (int)((lid / ntiles) % nchunks);
    int bx, by;
    sg2_tile(tm, fast, lid % ntiles, bx, by);
    const long long slot = (fast ? nb_border : 0) + lid;
    // outline blocks in the rows of the interior rectangle are column-border blocks (an empty rectangle -- small frames, a time
    // factor -- has iy1 < iy0: none)
    const bool colb = !fast && by >= tm.iy0 && by <= tm.iy1;
    if (fast) K::template run<0>(g, w, x, xp, xn, G, zchunk, chunk, bx, by, win, slot, partials, sa, sh);
    else if (colb) K::template run<1>(g, w, x, xp, xn, G, zchunk, chunk, bx, by, win, slot, partials, sa, sh);
    else K::template run<2>(g, w, x, xp, xn, G, zchunk, chunk, bx, by, win, slot, partials, sa, sh);
}

}  // namespace tv
