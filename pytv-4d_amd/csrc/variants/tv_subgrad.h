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
#ifndef SG_BREAK_COND
#define SG_BREAK_COND (MODE == 1 && S == HYBRID)
#endif

namespace tv {

// Opaque use of a vector: stops LLVM from SINKING the accumulation chains of a frame below the barrier that follows the
// frame loop (it does: the sums are only consumed there, and the ~14 product vectors of every frame then stay live
// to the end of the loop -- 70 VGPRs per frame instead of 20).
__device__ __forceinline__ void pin(F4& a) { asm volatile("" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3])); }

// value of the lane one to the left / right inside the 16-lane row (= the neighbouring 4-column vector of the same
// image row): a DPP-modified move, no LDS traffic.  Row-edge lanes get 0 (they are ring lanes: never used).
__device__ __forceinline__ float from_left_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111 /* row_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float from_right_lane(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101 /* row_shl:1 */, 0xF, 0xF, true));
}

// 1/|Dx| from |Dx|^2 with one v_rsq_f32 (1 ulp).  A |Dx|^2 that is not a normal fp32 number (|Dx| < 1.1e-19) counts
// as a zero gradient: v_rsq_f32 flushes denormal inputs, and fp32 squares carry no information there anyway
// (the reference's own float32 path loses |Dx| in the same range).
__device__ __forceinline__ float inv_norm(float ss, bool site_ok) {
    return (ss >= tiny_sumsq<float>() && site_ok) ? __builtin_amdgcn_rsqf(ss) : 0.f;
}

// MODE 1: instead of G the kernel writes the sub-gradient DESCENT step of the README loop (README.md:122-123)
//     x_out = x - step * ((x - x0) + lambda * G(x)),     fid partial = 1/2 |x_out - x0|^2
// (x is ping-ponged: neighbouring tiles still read the old image), which saves writing and re-reading G.
// MODE 2: G as in MODE 0 plus the per-voxel norms |Dx| (zeros replaced by +inf: the reference's grad_norms,
//     pytv/tv_GPU.py:88,135-139), 3 words per voxel instead of the >= 5 of the two-pass form.
struct SgStepArgs {
    const float* x0;
    float* x_out;
    float step, lambda;
    double* part_fid;
    float* norms;
};

// The kernel is issue-bound (hybrid: ~200 vector instructions per site-vector and frame), so the per-site
// arithmetic is kept to the minimum: border masks, the 1/sqrt(2) (hybrid) or 1/2 (central) of the scheme and the
// axis weights are folded into per-lane multipliers outside the loops; the backward time difference is the forward
// one of the previous frame and is carried, not recomputed; 1/|Dx| is one v_rsq_f32.
// central: ONE channel per axis, d = 1/2 w (x(+e) - x(-e)) on interior points, whose product goes to BOTH neighbours
// (G(v) = 1/2 sum_a P_a(v-e) - P_a(v+e)); two-point z / t axes (forward stencil) are left to the two-pass path.
// TWIN (M == SG_TWN == 8 only): volumes with more than 8 frames.  A block then works on a WINDOW of 8 consecutive frames
// [t0, t0 + 8) with t0 = 6 w - 1: it computes 1/|Dx| and the products on all 8 (the x frame on either side of the window
// is read for the time differences of the first / last one) but stores G for the inner 6 only -- the same overlap trick
// as in y and x, now along time: 8 / 6 of the work, still one pass and no fix-up.
constexpr int SG_TWN = 8, SG_TWU = SG_TWN - 2;

template <int S, int M, int NW, int MODE, bool TWIN = false>
__global__ __launch_bounds__(64 * NW, 2) void k_subgrad_one(DG g, WT<float> w, const float* __restrict__ x,
                                                            const float* __restrict__ xp, const float* __restrict__ xn,
                                                            float* __restrict__ G, int zchunk, int nchunks, double* __restrict__ partials,
                                                            SgStepArgs sa) {
    constexpr bool CEN = (S == CENTRAL);
    constexpr bool UP = (S == UPWIND || S == HYBRID), DN = (S == DOWNWIND || S == HYBRID);
    constexpr bool HALO = (S == HYBRID || CEN);   // the norm of a ring row looks at the row outside the tile
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
    // XCD-aware tile order: the hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its
    // own L2).  Neighbouring tiles share their ring rows / columns, so every XCD gets a CONTIGUOUS run of tiles:
    // logical id = (id % 8) * per_xcd + id / 8  (grid padded to 8 * per_xcd; the padding blocks leave at once)
    const int tiles_y = (g.ny + UR - 1) / UR;
    const int Mg = TWIN ? g.m : M;                                        // frames of the volume
    const int nwin = TWIN ? (Mg + SG_TWU - 1) / SG_TWU : 1;
    const long long ntiles = (long long)tiles_x * tiles_y, total = ntiles * nchunks * nwin, per_xcd = (total + 7) / 8;
    const long long lid = (long long)(blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (lid >= total) return;
    const int win = (int)(lid / (ntiles * nchunks));
    const int chunk = (int)((lid / ntiles) % nchunks), tile = (int)(lid % ntiles);
    const int t0 = TWIN ? win * SG_TWU - 1 : 0;                           // volume frame of this block's frame 0
    // is frame t of the block a frame of the volume / one whose G this block stores?
    auto fvalid = [&](int t) { return !TWIN || (t0 + t >= 0 && t0 + t < Mg); };
    auto fstore = [&](int t) { return !TWIN || (t0 + t >= win * SG_TWU && t0 + t < win * SG_TWU + SG_TWU && t0 + t < Mg); };
    auto foff_t = [&](int t) { return (long long)(t0 + t) * g.s_t; };     // uniform
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int cv = bx * UC - 1 + lx, y = by * UR - 1 + ry, col0 = cv * 4;
    const bool in = (cv >= 0) && (cv < nxv) && (y >= 0) && (y < g.ny);               // site inside the frame
    const bool useful = in && (ry >= 1) && (ry <= RB - 2) && (lx >= 1) && (lx <= 14);
    const unsigned voff = in ? (unsigned)(((long long)y * g.nx + col0) * 4) : 0u;    // byte offset inside a frame (< 2^32: host)
    const int zs = chunk * zchunk;
    const int ze = (zs + zchunk < g.nz) ? zs + zchunk : g.nz;
    const F4 zero = vsplat<float, 4>(0.f);
    const float s = (S == HYBRID) ? Consts<float>::inv_sqrt2() : (CEN ? 0.5f : 1.f);
    // frame-border masks as multipliers, with the scheme's scale folded in (straight-line code: every `if` around
    // a vector costs registers here).  central: a difference exists on interior points only
    const bool has_pr = in && (y > 0), has_nr = in && (y + 1 < g.ny);
    const float ms_pr = (CEN ? (has_pr && has_nr) : has_pr) ? s : 0.f, ms_nr = (CEN ? (has_pr && has_nr) : has_nr) ? s : 0.f;
    const float ms_c0 = (in && col0 > 0) ? s : 0.f, ms_c3 = (in && col0 + 3 < g.nx - 1) ? s : 0.f;
    const float ms_in = in ? s : 0.f;
    F4 mft = zero;                        // time-axis multiplier: s * sqrt(reg_time) * mask factor
    if (g.ta && in) mft = (s * w.wt) * mask_factor<float, 4>(g, w.sf, y, col0);
    // ring rows of a hybrid / central tile read the row just outside the tile from memory
    const bool ring_up = HALO && (ry == 0) && in && (y > 0), ring_dn = HALO && (ry == RB - 1) && in && (y + 1 < g.ny);
    const unsigned hoff = ring_up ? voff - (unsigned)g.nx * 4u : voff + (unsigned)g.nx * 4u;
    // cross-wave row neighbours: slot of the wave above / below (clamped), and whether this lane takes them
    const int w_up = (wv > 0) ? wv - 1 : 0, w_dn = (wv < NW - 1) ? wv + 1 : NW - 1;
    const bool take_up = (rr == 0), take_dn = (rr == 3);
    const float m_xup = (rr == 0 && wv > 0) ? 1.f : 0.f, m_xdn = (rr == 3 && wv < NW - 1) ? 1.f : 0.f;
    const float m_iup = (rr > 0) ? 1.f : 0.f, m_idn = (rr < 3) ? 1.f : 0.f;
    const float wzs = g.za ? s * w.wz : 0.f;
    double acc = 0.0, acc_fid = 0.0;

    F4 C[M], Gp[M], Gc[M];
    {
        // centre plane of the first step and the plane below it
        const int z_c = g.za ? zs - 1 : zs;
        const float* pp = g.za ? zplane<float>(g, x, xp, xn, 2, z_c - 1) : nullptr;
        const float* pc = zplane<float>(g, x, xp, xn, 2, z_c);
#pragma unroll
        for (int t = 0; t < M; ++t) {
            lds_P[t][tid] = (in && pp && fvalid(t)) ? ldu(pp + foff_t(t), voff) : zero;
            C[t] = (in && pc && fvalid(t)) ? ldu(pc + foff_t(t), voff) : zero;
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
        for (int t = 0; t < M; ++t) N[t] = (want_next && fvalid(t)) ? ldu(pn + foff_t(t), voff) : zero;
#pragma unroll
        for (int t = 0; t < M; ++t)
            if (rr == 0 || rr == 3) xe[t][wv][rr == 3 ? 1 : 0][lx] = C[t];
        __syncthreads();
        // uniform per step: z weights (0 when the neighbour plane does not exist), validity of this plane
        const bool z_prev = plane_in && (gz > 0), z_next = plane_in && (gz + 1 < g.nzg);
        const float wzn = (CEN ? (z_prev && z_next) : z_next) ? wzs : 0.f, wzp = z_prev ? wzs : 0.f;
        const bool site_ok = in && plane_in;                          // 1/|Dx| of a site that does not exist is 0
        const float m_row = plane_in ? 1.f : 0.f;
        const bool count = useful && plane_in && (zl >= zs) && (zl < ze);
        const bool store = useful && (zl - 1 >= zs) && (zl - 1 < ze);
        const bool halo_here = (ring_up || ring_dn) && (pc != nullptr);
        // halo rows: two frames ahead of their use (one exposed latency per frame would stall the whole block at
        // the barrier: only the first and the last wave have these loads)
        F4 hq[2] = {zero, zero};
        if (HALO && halo_here) {
            if (fvalid(0)) hq[0] = ldu(pc + foff_t(0), hoff);
            if (M > 1 && fvalid(1)) hq[1] = ldu(pc + foff_t(1), hoff);
        }
        F4 pf_t_prev = zero;                 // product of the time channel of frame t-1 (added to frame t)
        F4 f_t_prev = zero;                  // forward time difference of frame t-1 == backward one of frame t
#pragma unroll
        for (int t = 0; t < M; ++t) {
            // never taken (g.m == M), but it keeps the compiler from interleaving the frames: the hybrid descent-step
            // instantiation then needs 254 VGPRs and no scratch instead of 256 + 72 B/lane (69 -> 78 it/s on the north-star
            // volume); the G-storing instantiations are faster without it (measured)
            if (SG_BREAK_COND && t >= g.m) break;
            const F4 c = C[t];
            // ---- neighbourhood of x(zl, t) ------------------------------------------------------------
            // row neighbours: inside the wave by a 16-lane shuffle; first / last row of the wave from the neighbouring
            // wave's LDS slot (the block's first / last row: from the halo load)
            F4 xu = zero, xd = zero;
            F4 h = zero;
            if (HALO) {
                h = hq[t & 1];
                if (halo_here && t + 2 < M && fvalid(t + 2)) hq[t & 1] = ldu(pc + foff_t(t + 2), hoff);
            }
            if (DN || CEN) {
                xu = shfl_up16(c);
                F4 eu = xe[t][w_up][1][lx];
                if (HALO && wv == 0) eu = h;
#pragma unroll
                for (int i = 0; i < 4; ++i) xu.v[i] = take_up ? eu.v[i] : xu.v[i];
            }
            if (UP || CEN) {
                xd = shfl_down16(c);
                F4 ed = xe[t][w_dn][0][lx];
                if (HALO && wv == NW - 1) ed = h;
#pragma unroll
                for (int i = 0; i < 4; ++i) xd.v[i] = take_dn ? ed.v[i] : xd.v[i];
            }
            const float xl = from_left_lane(c.v[3]), xr = from_right_lane(c.v[0]);      // ring lanes: don't care
            // ---- gradient channels ---------------------------------------------------------------------
            // f_*: forward (upwind / hybrid-up) or central channel; b_*: backward channel
            F4 f_r, b_r = zero, f_c, b_c = zero, f_z, b_z = zero, f_t = zero, b_t = zero;
            if (CEN) {
                f_r = (ms_nr * m_row) * (xd - xu);
                f_c.v[0] = (ms_c0 * m_row) * (c.v[1] - xl); f_c.v[1] = (ms_in * m_row) * (c.v[2] - c.v[0]);
                f_c.v[2] = (ms_in * m_row) * (c.v[3] - c.v[1]); f_c.v[3] = (ms_c3 * m_row) * (xr - c.v[2]);
                f_z = wzn * (N[t] - lds_P[t][tid]);
                if (!TWIN) {
                    if (t > 0 && t + 1 < M) f_t = (m_row * mft) * (C[(t + 1 < M) ? t + 1 : t] - C[(t > 0) ? t - 1 : 0]);
                } else if (t0 + t > 0 && t0 + t + 1 < Mg) {      // interior frame of the volume; window ends: one more x frame
                    const F4 xn_t = (t + 1 < M) ? C[(t + 1 < M) ? t + 1 : t] : ((in && pc) ? ldu(pc + foff_t(t + 1), voff) : zero);
                    const F4 xp_t = (t > 0) ? C[(t > 0) ? t - 1 : 0] : ((in && pc) ? ldu(pc + foff_t(t - 1), voff) : zero);
                    f_t = (m_row * mft) * (xn_t - xp_t);
                }
            } else {
                f_r = (ms_nr * m_row) * (xd - c);
                b_r = (ms_pr * m_row) * (c - xu);
                const float e0 = (ms_in * m_row) * (c.v[1] - c.v[0]), e1 = (ms_in * m_row) * (c.v[2] - c.v[1]),
                            e2 = (ms_in * m_row) * (c.v[3] - c.v[2]);
                f_c.v[0] = e0; f_c.v[1] = e1; f_c.v[2] = e2; f_c.v[3] = (ms_c3 * m_row) * (xr - c.v[3]);
                b_c.v[0] = (ms_c0 * m_row) * (c.v[0] - xl); b_c.v[1] = e0; b_c.v[2] = e1; b_c.v[3] = e2;
                f_z = wzn * (N[t] - c);
                if (DN) b_z = wzp * (c - lds_P[t][tid]);
                if (!TWIN) {
                    if (t + 1 < M) f_t = (m_row * mft) * (C[(t + 1 < M) ? t + 1 : t] - c);
                    b_t = f_t_prev;
                } else {
                    if (fvalid(t) && t0 + t + 1 < Mg) {
                        const F4 xn_t = (t + 1 < M) ? C[(t + 1 < M) ? t + 1 : t] : ((in && pc) ? ldu(pc + foff_t(t + 1), voff) : zero);
                        f_t = (m_row * mft) * (xn_t - c);
                    }
                    b_t = f_t_prev;
                    if (t == 0 && DN && t0 >= 1)                 // backward difference of the window's first frame
                        b_t = (m_row * mft) * (c - ((in && pc) ? ldu(pc + foff_t(-1), voff) : zero));
                }
                f_t_prev = f_t;
            }
            F4 ss = zero;
            if (S == HYBRID) ss = ((((((f_r * f_r + f_c * f_c) + b_r * b_r) + b_c * b_c) + f_z * f_z) + b_z * b_z) + f_t * f_t) + b_t * b_t;
            if (S == UPWIND || CEN) ss = ((f_r * f_r + f_c * f_c) + f_z * f_z) + f_t * f_t;
            if (S == DOWNWIND) ss = ((b_r * b_r + b_c * b_c) + b_z * b_z) + b_t * b_t;
            F4 n, rn;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                n.v[i] = inv_norm(ss.v[i], site_ok && fvalid(t));
                rn.v[i] = ss.v[i] * n.v[i];
            }
            // four norms in fp32 (each carries its own 2^-24 already), then fp64 across frames / planes / threads
            if (count && fstore(t)) {
                acc += (double)((rn.v[0] + rn.v[1]) + (rn.v[2] + rn.v[3]));
                if (MODE == 2) {
                    F4 nv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) nv.v[i] = (n.v[i] > 0.f) ? rn.v[i] : __builtin_inff();
                    stu(sa.norms + (long long)zl * g.s_z + foff_t(t), voff, nv);
                }
            }
            // ---- scatter the products ---------------------------------------------------------------------
            F4 gc = Gc[t], gn = zero;
            if (CEN) {          // one product per axis, to the neighbours on both sides
                const F4 pf_r = f_r * n;
                const F4 from_up = shfl_up16(pf_r), from_dn = shfl_down16(pf_r);
                if (rr == 3) ye[t][wv][1][lx] = pf_r;
                if (rr == 0) ye[t][wv][0][lx] = pf_r;
                gc = gc + (m_iup * from_up - m_idn * from_dn);
                const F4 pf_c = f_c * n;
                gc = gc + (shift_right<float, 4>(pf_c, from_left_lane(pf_c.v[3])) - shift_left<float, 4>(pf_c, from_right_lane(pf_c.v[0])));
                const F4 pf_z = f_z * n;
                gn = pf_z;
                Gp[t] = Gp[t] - pf_z;
                const F4 pf_t = f_t * n;
                gc = gc + pf_t_prev;
                pf_t_prev = pf_t;
                if (t > 0) { Gc[t - 1] = Gc[t - 1] - pf_t; pin(Gc[t - 1]); }
            }
            if (UP) {           // forward channels: to the next site (+) and to the site itself (-)
                const F4 pf_r = f_r * n;
                const F4 from_up = shfl_up16(pf_r);                       // PF_r of the row above (same wave)
                if (rr == 3) ye[t][wv][1][lx] = pf_r;
                gc = gc + (m_iup * from_up - pf_r);
                const F4 pf_c = f_c * n;
                gc = gc + (shift_right<float, 4>(pf_c, from_left_lane(pf_c.v[3])) - pf_c);
                const F4 pf_z = f_z * n;
                gc = gc - pf_z;
                gn = pf_z;
                const F4 pf_t = f_t * n;
                gc = gc + (pf_t_prev - pf_t);
                pf_t_prev = pf_t;
            }
            if (DN) {           // backward channels: to the site itself (+) and to the previous site (-)
                const F4 pb_r = b_r * n;
                const F4 from_dn = shfl_down16(pb_r);                     // PB_r of the row below (same wave)
                if (rr == 0) ye[t][wv][0][lx] = pb_r;
                gc = gc + (pb_r - m_idn * from_dn);
                const F4 pb_c = b_c * n;
                gc = gc + (pb_c - shift_left<float, 4>(pb_c, from_right_lane(pb_c.v[0])));
                const F4 pb_z = b_z * n;
                gc = gc + pb_z;
                Gp[t] = Gp[t] - pb_z;
                const F4 pb_t = b_t * n;
                gc = gc + pb_t;
                if (t > 0) { Gc[t - 1] = Gc[t - 1] - pb_t; pin(Gc[t - 1]); }
            }
            pin(gc);
            Gc[t] = gc;
            // ---- plane zl-1 is complete ------------------------------------------------------------------
            if (store && fstore(t)) {
                F4 o = Gp[t];
                if (S == HYBRID || CEN) o = s * o;
                const long long foff = (long long)(zl - 1) * g.s_z + foff_t(t);      // uniform
                if (MODE != 1) {
                    stu(G + foff, voff, o);
                } else {
                    const F4 x0v = ldu(sa.x0 + foff, voff), p = lds_P[t][tid];      // p = x(zl-1)
                    F4 xo;
                    float e2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        xo.v[i] = p.v[i] - sa.step * ((p.v[i] - x0v.v[i]) + sa.lambda * o.v[i]);
                        const float e = xo.v[i] - x0v.v[i];
                        e2 += e * e;
                    }
                    stu(sa.x_out + foff, voff, xo);
                    acc_fid += 0.5 * (double)e2;
                }
            }
            pin(gn);
            Gp[t] = gn;                      // the slot of the finished plane now carries the start of G(zl+1)
            lds_P[t][tid] = c;
        }
        __syncthreads();
        // cross-wave row products, rotation of the planes
#pragma unroll
        for (int t = 0; t < M; ++t) {
            F4 gc = Gc[t];
            if (UP || CEN) gc = gc + m_xup * ye[t][w_up][1][lx];
            if (DN || CEN) gc = gc - m_xdn * ye[t][w_dn][0][lx];
            Gc[t] = Gp[t];
            Gp[t] = gc;
            C[t] = N[t];
        }
    }
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0 && threadIdx.y == 0) partials[lid] = acc;
    if (MODE == 1) {
        acc_fid = block_sum(acc_fid, sm);
        if (threadIdx.x == 0 && threadIdx.y == 0) sa.part_fid[lid] = acc_fid;
    }
}

}  // namespace tv
