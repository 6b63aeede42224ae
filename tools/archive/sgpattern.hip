// tools/archive/sgpattern.hip -- round 4, verdict item 5: what does the MEMORY PATTERN of a z-marching, ring-overlapped tile kernel cost?
//
// profiles/r3_subgrad_col_pattern.txt showed that k_subgrad_col (one-pass TV + sub-gradient) costs what its loads and stores cost: a
// build without arithmetic takes the same 1.55 ms at 64x8x1024x1024, the same bytes in ring-less 4 x 256 tiles 1.03 ms.  This program
// is that copy-only kernel with the GEOMETRY as a parameter, so that a tile can be chosen by measurement before a kernel is built
// around it: lane shape (rows x columns, 4 / 8 / 16-byte accesses), lanes per wave row, waves per block in y and x, ring rows / columns
// (loaded, not stored), halo rows (one more row above / below the block), the z partition (uniform chunks with overlap planes, or ONE
// balanced contiguous share of the (tile, plane) list per block), and the block order.  Like the real kernel: raw buffer accesses
// whose offsets the hardware range-checks (no branches), all M frames of a plane per step, a load ring D frames ahead, one block
// (8 or 16 waves) per CU.
//
// build: hipcc -O3 --offload-arch=gfx950 -o tools/sgpattern tools/archive/sgpattern.hip        run: tools/sgpattern [nz m ny nx] [config ...]
// config = LR,LC,WX,NWY,NWX,ry,rx,halo,zc,ovl,ovh,grid,xcd[,nt[,rl[,dd,dl,hd,ustride,xoff[,rg]]]]   (grid > 0: balanced mode with that many blocks, zc ignored;
// nt: the aux (cache policy) bits of the stores -- 1 sc0, 2 nt, 16 sc1 and their sums; rl: ring columns loaded per side, the remaining ring lanes idle -- an ALIGNED tile with a narrow ring)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)

using Rsrc = __amdgpu_buffer_rsrc_t;
constexpr unsigned OOB = 0x80000000u;
constexpr int M = 8;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int LC> struct Vec;
template <> struct Vec<1> { float v; };
template <> struct Vec<2> { v2f v; };
template <> struct Vec<4> { v4f v; };

template <int LC> __device__ __forceinline__ Vec<LC> ld(Rsrc r, unsigned off);
template <> __device__ __forceinline__ Vec<1> ld<1>(Rsrc r, unsigned off) { Vec<1> o; o.v = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0)); return o; }
template <> __device__ __forceinline__ Vec<2> ld<2>(Rsrc r, unsigned off) {
    const v2i a = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    Vec<2> o; o.v.x = __int_as_float(a.x); o.v.y = __int_as_float(a.y); return o;
}
template <> __device__ __forceinline__ Vec<4> ld<4>(Rsrc r, unsigned off) {
    const v4i a = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    Vec<4> o; o.v.x = __int_as_float(a.x); o.v.y = __int_as_float(a.y); o.v.z = __int_as_float(a.z); o.v.w = __int_as_float(a.w); return o;
}
template <int AUX> __device__ __forceinline__ void st1(Rsrc r, unsigned off, Vec<1> v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v.v), r, (int)off, 0, AUX); }

template <int AUX> __device__ __forceinline__ void st2(Rsrc r, unsigned off, Vec<2> v) {
    v2i a; a.x = __float_as_int(v.v.x); a.y = __float_as_int(v.v.y);
    __builtin_amdgcn_raw_buffer_store_b64(a, r, (int)off, 0, AUX);
}

template <int AUX> __device__ __forceinline__ void st4(Rsrc r, unsigned off, Vec<4> v) {
    v4i a; a.x = __float_as_int(v.v.x); a.y = __float_as_int(v.v.y); a.z = __float_as_int(v.v.z); a.w = __float_as_int(v.v.w);
    __builtin_amdgcn_raw_buffer_store_b128(a, r, (int)off, 0, AUX);
}
template <int LC, int AUX> __device__ __forceinline__ void st(Rsrc r, unsigned off, Vec<LC> v) {
    if constexpr (LC == 1) st1<AUX>(r, off, v); else if constexpr (LC == 2) st2<AUX>(r, off, v); else st4<AUX>(r, off, v);
}
template <int LC> __device__ __forceinline__ void addh(Vec<LC>& a, const Vec<LC>& h, float zero);
template <> __device__ __forceinline__ void addh<1>(Vec<1>& a, const Vec<1>& h, float zero) { a.v += h.v * zero; }
template <> __device__ __forceinline__ void addh<2>(Vec<2>& a, const Vec<2>& h, float zero) { a.v.x += h.v.x * zero; a.v.y += h.v.y * zero; }
template <> __device__ __forceinline__ void addh<4>(Vec<4>& a, const Vec<4>& h, float zero) { a.v.x += h.v.x * zero; a.v.y += h.v.y * zero; a.v.z += h.v.z * zero; a.v.w += h.v.w * zero; }

struct Geo {
    int nz, m, ny, nx;
    int wx, nwy, nwx;          // lanes per wave row; waves per block in y / x
    int ry, rx, halo;          // ring rows / columns per side (not stored), halo rows (0 / 1)
    int zc, ovl, ovh;          // uniform mode: planes per chunk; planes read before / after the stored range
    int grid, xcd, nt;         // balanced mode if grid > 0; XCD-aware block order; non-temporal stores
    int rl;                    // ring columns actually LOADED per side (<= rx: the other ring lanes idle, their offsets out of range)
    int hd;                    // depth of the halo-row ring (0: same as dd)
    int ustride, xoff;         // EXPERIMENT: column stride / origin of the tiles other than UC / -rx (leaves columns unwritten: alignment tests)
    int rg;                    // round 5: GATHERS per wave and plane step -- the ring columns of a wave's aligned 64-column strip fetched by a 64-lane
                               // 'mini slot' (lane = side, row, frame) instead of by ring lanes: 4-byte loads from the neighbouring tiles' lines
    int dd, dl;                // depth of the register load ring (frames ahead); dl > 0: the LDS-DMA variant with dl ring slots per wave
    int tx, ty, nchunks;       // derived: tile grid, chunks
    float zero;
};

template <int LR, int LC, int D, int NT, int HD = D, int RG = 0>
__global__ __launch_bounds__(1024, 1) void k_pattern(Geo g, const float* __restrict__ x, float* __restrict__ G) {
    extern __shared__ char lds_pad[];           // occupancy: one block per CU, like the register-bound real kernel
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x == 0) lds_pad[0] = 0;
    const int wy = 64 / g.wx;
    const int lx = lane % g.wx, ly = lane / g.wx;
    const int wyi = wid % g.nwy, wxi = wid / g.nwy;
    const int TB = g.nwy * wy * LR, TC = g.nwx * g.wx * LC;
    const int UR = TB - 2 * g.ry, UC = TC - 2 * g.rx;
    const long long s_t = (long long)g.ny * g.nx, s_z = s_t * g.m;
    const int fbytes = (int)(s_t * 4);
    const long long ntiles = (long long)g.tx * g.ty;

    long long id = blockIdx.x, total = gridDim.x;
    if (g.xcd) { const long long per = (total + 7) / 8; id = (id % 8) * per + id / 8; if (id >= total) return; }

    // the work of this block: [u0, u1) of the (tile, plane) list in balanced mode, one (tile, chunk) otherwise
    long long u0, u1;
    const long long units = ntiles * g.nz;
    if (g.grid > 0) { u0 = units * id / total; u1 = units * (id + 1) / total; }
    else { const long long tile = id % ntiles; const int ch = (int)(id / ntiles); u0 = tile * g.nz + (long long)ch * g.zc; u1 = tile * g.nz + std::min<long long>(g.nz, (long long)(ch + 1) * g.zc); }

    while (u0 < u1) {
        const long long tile = u0 / g.nz;
        const int zs = (int)(u0 % g.nz);
        const int ze = (int)std::min<long long>(g.nz, zs + (u1 - u0));
        u0 += ze - zs;
        const int bx = (int)(tile % g.tx), by = (int)(tile / g.tx);
        const int cx = bx * (g.ustride ? g.ustride : UC) + (g.ustride ? g.xoff : -g.rx) + (wxi * g.wx + lx) * LC;
        const int yb = by * UR - g.ry + (wyi * wy + ly) * LR;
        const int lc0 = (wxi * g.wx + lx) * LC;           // column of this lane inside the tile
        const bool in_x = cx >= 0 && cx + LC <= g.nx && lc0 + LC > g.rx - g.rl && lc0 < TC - g.rx + g.rl;
        const bool own_x = in_x && (wxi * g.wx + lx) * LC >= g.rx && (wxi * g.wx + lx) * LC + LC <= TC - g.rx;
        unsigned roff[LR], soff[LR];
#pragma unroll
        for (int i = 0; i < LR; ++i) {
            const int y = yb + i, yl = (wyi * wy + ly) * LR + i;
            const bool in = in_x && y >= 0 && y < g.ny;
            const bool own = in && own_x && yl >= g.ry && yl < TB - g.ry;
            const unsigned off = (unsigned)(((long long)y * g.nx + cx) * 4);
            roff[i] = in ? off : OOB;
            soff[i] = own ? off : OOB;
        }
        unsigned hoff = OOB;
        if (g.halo && in_x) {
            if (wyi == 0 && ly == 0 && yb > 0) hoff = (unsigned)(((long long)(yb - 1) * g.nx + cx) * 4);
            if (wyi == g.nwy - 1 && ly == wy - 1 && yb + LR < g.ny) hoff = (unsigned)(((long long)(yb + LR) * g.nx + cx) * 4);
        }
        auto frame = [&](const float* base, int z, int t, bool ok) {
            const bool v = ok && z >= 0 && z < g.nz;
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (v ? (long long)z * s_z + (long long)t * s_t : 0)), 0, v ? fbytes : 0, 0x00020000);
        };
        const int z_lo = zs - g.ovl, z_hi = ze - 1 + g.ovh;
        // ring-column mini slot: lane = (side, row i of the strip, frame t); gather k of a plane: 0 = the ring column itself, 1 = one
        // column further out, 2 / 3 = the rows above / below in the ring column, 4 = the strip's own edge column.  One descriptor per
        // plane (all M frames), the frame offset rides in the per-lane offset.
        unsigned goff[RG > 0 ? RG : 1];
        if (RG > 0) {
            const int side = lane >> 5, gi = (lane >> 3) & 3, gt = lane & 7;
            const int c_edge = cx - lx * LC + (side ? g.wx * LC - 1 : 0);      // the strip's first / last column
            const int dir = side ? 1 : -1;
#pragma unroll
            for (int k = 0; k < RG; ++k) {
                if (RG <= 2) {
                    // RG = 1 / 2: ONE 16-byte gather [ring - 1, ring, own edge, own edge + 1] (8-byte aligned, straddles the line
                    // boundary), + (RG = 2) the halo rows of the ring column: 32 active lanes (rows -1 / LR of the strip only)
                    int col = side ? c_edge - 1 : c_edge - 2, row = yb - ly * LR + gi;
                    bool ok = gi < LR && col >= 0 && col + 4 <= g.nx && row >= 0 && row < g.ny;
                    if (k == 1) {
                        col = c_edge + dir;
                        row = (gi == 0) ? row - 1 : row + 1;
                        ok = (gi == 0 || gi == LR - 1) && col >= 0 && col < g.nx && row >= 0 && row < g.ny;
                    }
                    goff[k] = ok ? (unsigned)((long long)gt * fbytes + ((long long)row * g.nx + col) * 4) : OOB;
                    continue;
                }
                int col = c_edge + dir, row = yb - ly * LR + gi;
                if (k == 1) col += dir;
                if (k == 2) row -= 1;
                if (k == 3) row += 1;
                if (k == 4) col = c_edge;
                const bool ok = gi < LR && col >= 0 && col < g.nx && row >= 0 && row < g.ny;
                goff[k] = ok ? (unsigned)((long long)gt * fbytes + ((long long)row * g.nx + col) * 4) : OOB;
            }
        }
        auto plane_rs = [&](int z) {
            const bool v = z >= 0 && z < g.nz && z <= z_hi;
            return __builtin_amdgcn_make_buffer_rsrc((void*)(x + (v ? (long long)z * s_z : 0)), 0, v ? fbytes * M : 0, 0x00020000);
        };
        float gq[RG > 0 ? RG : 1];
        if (RG > 0) {
            const Rsrc rp = plane_rs(z_lo);
#pragma unroll
            for (int k = 0; k < RG; ++k) gq[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, (int)goff[k], 0, 0));
        }
        v4i gq4 = {0, 0, 0, 0};
        if (RG > 0 && RG <= 2) gq4 = __builtin_amdgcn_raw_buffer_load_b128(plane_rs(z_lo), (int)goff[0], 0, 0);
        Vec<LC> cur[M][LR], nq[D][LR], hq[HD];
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const Rsrc r = frame(x, z_lo, t, true);
#pragma unroll
            for (int i = 0; i < LR; ++i) cur[t][i] = ld<LC>(r, roff[i]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const Rsrc r = frame(x, z_lo + 1, d, z_lo + 1 <= z_hi);
#pragma unroll
            for (int i = 0; i < LR; ++i) nq[d][i] = ld<LC>(r, roff[i]);
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) hq[d] = ld<LC>(frame(x, z_lo, d, true), hoff);
        for (int z = z_lo; z <= z_hi; ++z) {
            const bool store = z >= zs && z < ze;
            float gsum = 0.f;
            if (RG > 0) {           // consume the ring values of plane z, request those of plane z + 1 (a whole step ahead)
#pragma unroll
                for (int k = (RG <= 2) ? 1 : 0; k < RG; ++k) gsum += gq[k];
                if (RG <= 2) gsum += __int_as_float(gq4.x) + __int_as_float(gq4.y) + __int_as_float(gq4.z) + __int_as_float(gq4.w);
                gsum *= g.zero;
                const Rsrc rp = plane_rs(z + 1);
                if (RG <= 2) gq4 = __builtin_amdgcn_raw_buffer_load_b128(rp, (int)goff[0], 0, 0);
#pragma unroll
                for (int k = (RG <= 2) ? 1 : 0; k < RG; ++k) gq[k] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rp, (int)goff[k], 0, 0));
            }
            // the halo row of frame t of plane z, requested HD frames ahead (HD = 2: what k_subgrad_col did until round 4 --
            // with ONE in-order vmcnt the wait for it is a wait for everything older, whatever the depth of the main ring)
#pragma unroll
            for (int t = 0; t < M; ++t) {
                const Vec<LC> h = hq[t % HD];
                if (t + HD < M) hq[t % HD] = ld<LC>(frame(x, z, t + HD, true), hoff);
                else hq[t % HD] = ld<LC>(frame(x, z + 1, t + HD - M, z + 1 <= z_hi), hoff);
                const Rsrc rg = frame(G, z, t, store);
#pragma unroll
                for (int i = 0; i < LR; ++i) {
                    Vec<LC> v = cur[t][i];
                    if (i == 0) addh<LC>(v, h, g.zero);
                    if (RG > 0 && i == 1 && t == 0) { Vec<LC> gv; gv = v; if constexpr (LC == 1) gv.v = gsum; addh<LC>(v, gv, 1.f); }
                    st<LC, NT>(rg, soff[i], v);
                    cur[t][i] = nq[t % D][i];
                }
                if (t + D < M) {
                    const Rsrc r = frame(x, z + 1, t + D, z + 1 <= z_hi);
#pragma unroll
                    for (int i = 0; i < LR; ++i) nq[t % D][i] = ld<LC>(r, roff[i]);
                } else {
                    const Rsrc r = frame(x, z + 2, t + D - M, z + 2 <= z_hi);
#pragma unroll
                    for (int i = 0; i < LR; ++i) nq[t % D][i] = ld<LC>(r, roff[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}


// ---- the same tile with the x rows delivered by LDS-DMA -----------------------------------------------------------------------
// Every wave owns a private ring of DL frame slots in LDS, a slot = 6 rows x 64 columns (its 4 rows + the row above and below:
// the neighbour rows the stencil needs, L1 / L2 hits).  `buffer_load_dword ... lds` writes a row of a slot without touching a
// VGPR, so the look-ahead is DL frames whatever the register budget; the compiler does not see these loads (inline asm), the
// kernel counts them itself: vmcnt is in order, an iteration issues 6 DMA rows + 4 stores, the data of frame j was requested DL
// iterations ago => s_waitcnt vmcnt(4 + (DL - 1) * 10).
__device__ __forceinline__ v4i mk_desc(const float* p, int nbytes) {
    const unsigned long long a = (unsigned long long)p;
    v4i d;
    d.x = (int)(unsigned)a;
    d.y = (int)((unsigned)(a >> 32) & 0xffffu);
    d.z = nbytes;
    d.w = 0x00020000;
    return d;
}
__device__ __forceinline__ void dma6(v4i d, unsigned lds_addr, unsigned o0, unsigned o1, unsigned o2, unsigned o3, unsigned o4, unsigned o5) {
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"                  // the ds_reads of this slot have returned before it is overwritten
        "s_mov_b32 %[k], m0\n\t"
        "s_mov_b32 m0, %[l]\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %[o0], %[d], 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x100\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %[o1], %[d], 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x100\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %[o2], %[d], 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x100\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %[o3], %[d], 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x100\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %[o4], %[d], 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x100\n\t"
        "s_nop 0\n\t"
        "buffer_load_dword %[o5], %[d], 0 offen lds\n\t"
        "s_mov_b32 m0, %[k]"
        : [k] "=&s"(keep)
        : [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [o4] "v"(o4), [o5] "v"(o5), [d] "s"(d), [l] "s"(lds_addr)
        : "memory", "scc");
}
template <int DL, int NT>
__global__ __launch_bounds__(512, 2) void k_pattern_dma(Geo g, const float* __restrict__ x, float* __restrict__ G) {
    extern __shared__ float lds[];
    constexpr int LR = 4;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int wyi = wid % g.nwy, wxi = wid / g.nwy;
    const int TB = g.nwy * LR, TC = g.nwx * 64;
    const int UR = TB - 2 * g.ry, UC = TC - 2 * g.rx;
    const long long s_t = (long long)g.ny * g.nx, s_z = s_t * g.m;
    const int fbytes = (int)(s_t * 4);
    const long long ntiles = (long long)g.tx * g.ty;
    float* ring = lds + wid * (DL * 6 * 64);                                          // this wave's slots
    const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) float*)ring;
    long long id = blockIdx.x, total = gridDim.x;
    if (g.xcd) { const long long per = (total + 7) / 8; id = (id % 8) * per + id / 8; if (id >= total) return; }
    long long u0, u1;
    const long long units = ntiles * g.nz;
    if (g.grid > 0) { u0 = units * id / total; u1 = units * (id + 1) / total; }
    else { const long long tile = id % ntiles; const int ch = (int)(id / ntiles); u0 = tile * g.nz + (long long)ch * g.zc; u1 = tile * g.nz + std::min<long long>(g.nz, (long long)(ch + 1) * g.zc); }
    constexpr int K = 4 + (DL - 1) * 10;
    static_assert(K < 64, "vmcnt has 6 bits");
    while (u0 < u1) {
        const long long tile = u0 / g.nz;
        const int zs = (int)(u0 % g.nz);
        const int ze = (int)std::min<long long>(g.nz, zs + (u1 - u0));
        u0 += ze - zs;
        const int bx = (int)(tile % g.tx), by = (int)(tile / g.tx);
        const int lc0 = wxi * 64 + lane;
        const int cx = bx * UC - g.rx + lc0;
        const int yb = by * UR - g.ry + wyi * LR;
        const bool in_x = cx >= 0 && cx < g.nx && lc0 >= g.rx - g.rl && lc0 < TC - g.rx + g.rl;
        const bool own_x = in_x && lc0 >= g.rx && lc0 < TC - g.rx;
        unsigned roff[6], soff[LR];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int y = yb - 1 + i;
            // rows of the strip always; the row above / below only where the stencil of a ring row would look at it (halo) or a
            // neighbouring wave of the block holds it
            const bool need = (i >= 1 && i <= 4) || g.halo || (i == 0 && wyi > 0) || (i == 5 && wyi < g.nwy - 1);
            roff[i] = (need && in_x && y >= 0 && y < g.ny) ? (unsigned)(((long long)y * g.nx + cx) * 4) : OOB;
        }
#pragma unroll
        for (int i = 0; i < LR; ++i) {
            const int y = yb + i, yl = wyi * LR + i;
            const bool own = own_x && y >= 0 && y < g.ny && yl >= g.ry && yl < TB - g.ry;
            soff[i] = own ? (unsigned)(((long long)y * g.nx + cx) * 4) : OOB;
        }
        auto desc = [&](const float* base, int z, int t, bool ok) {
            const bool v = ok && z >= 0 && z < g.nz;
            return mk_desc(base + (v ? (long long)z * s_z + (long long)t * s_t : 0), v ? fbytes : 0);
        };
        auto frame = [&](const float* base, int z, int t, bool ok) {
            const bool v = ok && z >= 0 && z < g.nz;
            return __builtin_amdgcn_make_buffer_rsrc((void*)(base + (v ? (long long)z * s_z + (long long)t * s_t : 0)), 0, v ? fbytes : 0, 0x00020000);
        };
        const int z_lo = zs - g.ovl, z_hi = ze - 1 + g.ovh;
        float cur[M][LR];
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const Rsrc r = frame(x, z_lo, t, true);
#pragma unroll
            for (int i = 0; i < LR; ++i) cur[t][i] = __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)roff[i + 1], 0, 0));
        }
        // frames are numbered from (z_lo + 1, 0); frame j lives in slot j % DL
        int jz = z_lo + 1, jt = 0, js = 0;          // the next frame to request: plane, frame, slot
#pragma unroll
        for (int d = 0; d < DL; ++d) {
            dma6(desc(x, jz, jt, jz <= z_hi), ring_addr + js * (6 * 256), roff[0], roff[1], roff[2], roff[3], roff[4], roff[5]);
            if (++jt == M) { jt = 0; ++jz; }
            if (++js == DL) js = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int rs = 0;                                  // slot of the frame about to be consumed
        for (int z = z_lo; z <= z_hi; ++z) {
            const bool store = z >= zs && z < ze;
#pragma unroll
            for (int t = 0; t < M; ++t) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
                const float* sl = ring + rs * (6 * 64);
                float nx[LR];
                const float up = sl[lane], dn = sl[5 * 64 + lane];
#pragma unroll
                for (int i = 0; i < LR; ++i) nx[i] = sl[(i + 1) * 64 + lane];
                // refill the slot: frame j + DL
                dma6(desc(x, jz, jt, jz <= z_hi), ring_addr + rs * (6 * 256), roff[0], roff[1], roff[2], roff[3], roff[4], roff[5]);
                if (++jt == M) { jt = 0; ++jz; }
                if (++rs == DL) rs = 0;
                const Rsrc rg = frame(G, z, t, store);
#pragma unroll
                for (int i = 0; i < LR; ++i) {
                    float v = cur[t][i];
                    if (i == 0) v += up * g.zero;
                    if (i == LR - 1) v += dn * g.zero;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rg, (int)soff[i], 0, NT);
                    cur[t][i] = nx[i];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of this segment's ring in flight when the next one starts
    }
}

__global__ void k_fill(float* x, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = (float)((i * 2654435761ull) % 65521ull);
}
__global__ void k_check(const float* x, const float* G, long long n, unsigned long long* bad) {
    unsigned long long b = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b += (x[i] != G[i]);
    if (b) atomicAdd(bad, b);
}

struct Cfg { int LR, LC; Geo g; std::string txt; };

template <int LR, int LC> static void launch(const Geo& g, int blocks, int threads, const float* x, float* G) {
    if (g.nt) {
        if (g.dd != 2) { fprintf(stderr, "store cache bits: depth 2 only\n"); exit(1); }
#define AUXCASE(a) if (g.nt == a) { hipLaunchKernelGGL((k_pattern<LR, LC, 2, a>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G); return; }
        AUXCASE(1) AUXCASE(2) AUXCASE(3) AUXCASE(16) AUXCASE(17) AUXCASE(18) AUXCASE(19)
#undef AUXCASE
        fprintf(stderr, "store aux %d not instantiated\n", g.nt); exit(1);
    }
    else if (g.rg > 0) {
        if constexpr (LR == 4 && LC == 1) {
            if (g.rg == 1) hipLaunchKernelGGL((k_pattern<4, 1, 2, 0, 2, 1>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
            else if (g.rg == 2) hipLaunchKernelGGL((k_pattern<4, 1, 2, 0, 2, 2>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
            else if (g.rg == 3) hipLaunchKernelGGL((k_pattern<4, 1, 2, 0, 2, 3>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
            else if (g.rg == 5) hipLaunchKernelGGL((k_pattern<4, 1, 2, 0, 2, 5>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
            else { fprintf(stderr, "gathers: 1, 2, 3 or 5\n"); exit(1); }
        } else { fprintf(stderr, "gathers: 4x1 lanes only\n"); exit(1); }
    }
    else if (g.dd == 2 && g.hd != 8) hipLaunchKernelGGL((k_pattern<LR, LC, 2, 0>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
    else if (g.dd == 4 && g.hd == 2) hipLaunchKernelGGL((k_pattern<LR, LC, 4, 0, 2>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
    else if (g.dd == 2 && g.hd == 8) hipLaunchKernelGGL((k_pattern<LR, LC, 2, 0, 8>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
    else if (g.dd == 4) hipLaunchKernelGGL((k_pattern<LR, LC, 4, 0>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
    else if (g.dd == 8) hipLaunchKernelGGL((k_pattern<LR, LC, 8, 0>), dim3(blocks), dim3(threads), 96 * 1024, 0, g, x, G);
    else { fprintf(stderr, "register ring depth %d not instantiated\n", g.dd); exit(1); }
}
static void launch_dma(const Geo& g, int blocks, int threads, const float* x, float* G) {
    if (threads != 512 || g.wx != 64) { fprintf(stderr, "the LDS-DMA variant is 4x1 lanes, 8 waves\n"); exit(1); }
    const int lds = 96 * 1024;      // >= 8 waves x DL x 1.5 KiB, and one block per CU
    if (g.dl == 2) hipLaunchKernelGGL((k_pattern_dma<2, 0>), dim3(blocks), dim3(threads), lds, 0, g, x, G);
    else if (g.dl == 4) hipLaunchKernelGGL((k_pattern_dma<4, 0>), dim3(blocks), dim3(threads), lds, 0, g, x, G);
    else if (g.dl == 6 && g.nt) hipLaunchKernelGGL((k_pattern_dma<6, 2>), dim3(blocks), dim3(threads), lds, 0, g, x, G);
    else if (g.dl == 6) hipLaunchKernelGGL((k_pattern_dma<6, 0>), dim3(blocks), dim3(threads), lds, 0, g, x, G);
    else { fprintf(stderr, "DMA ring depth %d not instantiated\n", g.dl); exit(1); }
}
static void dispatch(const Cfg& c, int blocks, int threads, const float* x, float* G) {
    if (c.g.dl > 0) { launch_dma(c.g, blocks, threads, x, G); return; }
#define CASE(a, b) if (c.LR == a && c.LC == b) { launch<a, b>(c.g, blocks, threads, x, G); return; }
    CASE(4, 1) CASE(2, 2) CASE(1, 4) CASE(4, 2) CASE(2, 4)
#undef CASE
    fprintf(stderr, "lane shape %dx%d not instantiated\n", c.LR, c.LC); exit(1);
}

int main(int argc, char** argv) {
    int nz = 64, m = 8, ny = 1024, nx = 1024, a0 = 1;
    if (argc >= 5 && !strchr(argv[1], ',')) { nz = atoi(argv[1]); m = atoi(argv[2]); ny = atoi(argv[3]); nx = atoi(argv[4]); a0 = 5; }
    if (m != M) { fprintf(stderr, "m must be %d\n", M); return 1; }
    std::vector<std::string> cfgs;
    for (int i = a0; i < argc; ++i) cfgs.push_back(argv[i]);
    if (cfgs.empty()) {
        const char* def[] = {
            // LR,LC,WX,NWY,NWX, ry,rx,halo, zc,ovl,ovh, grid,xcd
            "4,1,64,4,2, 1,2,1, 16,2,2, 0,1",      // k_subgrad_col hybrid as it is: 16 x 128 tiles, 14 x 120 stored, 16-plane chunks
            "4,1,64,4,2, 1,2,1, 32,2,2, 0,1",
            "4,1,64,4,2, 1,2,1, 64,2,2, 0,1",
            "4,1,64,4,2, 1,2,1, 0,2,2, 256,1",     // balanced: one share of the (tile, plane) list per CU
            "4,1,64,4,2, 1,2,1, 0,2,2, 256,0",
            "4,1,64,4,2, 1,2,1, 0,2,2, 512,1",
            "4,1,64,4,2, 1,0,1, 16,2,2, 0,1",      // no column ring (aligned 512-byte rows)
            "4,1,64,4,2, 0,2,0, 16,2,2, 0,1",      // no row ring
            "4,1,64,4,2, 0,0,0, 16,2,2, 0,1",      // no ring at all
            "4,1,64,4,2, 0,0,0, 0,2,2, 256,1",
            "4,1,64,4,2, 0,0,0, 0,0,0, 256,1",     // plain tiled copy
            "4,1,64,2,4, 1,2,1, 0,2,2, 256,1",     // 8 x 256
            "4,1,64,8,1, 1,2,1, 0,2,2, 256,1",     // 32 x 64
            "2,2,64,8,1, 1,2,1, 0,2,2, 256,1",     // 8-byte lanes: wave = 2 rows x 128 columns, 16 x 128 tile
            "2,2,64,4,2, 1,2,1, 0,2,2, 256,1",     // 8 x 256
            "4,2,64,4,1, 1,2,1, 0,2,2, 256,1",     // 4 fat waves: 16 x 128
            "4,2,64,4,2, 1,2,1, 0,2,2, 256,1",     // 8 fat waves: 16 x 256 (needs twice the registers of the real kernel)
            "1,4,64,8,1, 1,4,1, 0,2,2, 256,1",     // 16-byte lanes: wave = 1 row x 256 columns, 8 x 256
            "1,4,16,4,2, 1,4,1, 0,2,2, 256,1",     // 16-byte lanes, wave = 4 rows x 64 columns: 16 x 128
            "1,4,32,8,1, 1,4,1, 0,2,2, 256,1",     // wave = 2 rows x 128 columns: 16 x 128
            "2,4,32,4,1, 1,4,1, 0,2,2, 256,1",     // 4 fat waves, wave = 4 rows x 128 columns
            "4,1,64,4,4, 1,2,1, 0,2,2, 256,1",     // 16 waves: 16 x 256
            "2,2,64,8,2, 1,2,1, 0,2,2, 256,1",     // 16 waves, 8-byte lanes: 16 x 256
            "1,4,64,16,1, 1,4,1, 0,2,2, 256,1",    // 16 waves, 16-byte lanes: 16 x 256
        };
        for (const char* s : def) cfgs.push_back(s);
    }
    const long long n = (long long)nz * m * ny * nx;
    float *x, *G;
    CK(hipMalloc(&x, n * 4));
    CK(hipMalloc(&G, n * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, x, n);
    unsigned long long* bad;
    CK(hipMalloc(&bad, 8));
    CK(hipMemset(G, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# sgpattern %dx%dx%dx%d fp32: copy-only z-marching tile kernel; algorithmic = 2 words per voxel = %.3f GB\n", nz, m, ny, nx, 2.0 * n * 4 / 1e9);
    printf("# %-38s tile(stored)      blocks  read GB  written GB   min ms   med ms   alg TB/s  real TB/s\n", "LR,LC,WX,NWY,NWX,ry,rx,halo,zc,ovl,ovh,grid,xcd[,nt]");
    for (int rep = 0; rep < 2; ++rep)
    for (const std::string& s : cfgs) {
        Cfg c;
        int v[21] = {0}; v[14] = -1; v[15] = 2;
        int k = 0;
        std::string tmp = s;
        for (char* tok = strtok(&tmp[0], ", "); tok && k < 21; tok = strtok(nullptr, ", ")) v[k++] = atoi(tok);
        if (k < 13) { fprintf(stderr, "bad config %s\n", s.c_str()); return 1; }
        c.LR = v[0]; c.LC = v[1];
        Geo& g = c.g;
        g.nz = nz; g.m = m; g.ny = ny; g.nx = nx;
        g.wx = v[2]; g.nwy = v[3]; g.nwx = v[4]; g.ry = v[5]; g.rx = v[6]; g.halo = v[7]; g.zc = v[8]; g.ovl = v[9]; g.ovh = v[10]; g.grid = v[11]; g.xcd = v[12]; g.nt = v[13]; g.rl = (v[14] < 0 || v[14] > v[6]) ? v[6] : v[14]; g.dd = v[15]; g.dl = v[16]; g.hd = v[17]; g.ustride = v[18]; g.xoff = v[19]; g.rg = v[20];
        g.zero = 0.f;
        const int wy = 64 / g.wx, TB = g.nwy * wy * c.LR, TC = g.nwx * g.wx * c.LC, UR = TB - 2 * g.ry, UC = TC - 2 * g.rx;
        g.tx = g.ustride ? (nx + g.ustride - 1) / g.ustride : (nx + UC - 1) / UC; g.ty = (ny + UR - 1) / UR;
        g.nchunks = g.grid > 0 ? 0 : (nz + g.zc - 1) / g.zc;
        const long long ntiles = (long long)g.tx * g.ty;
        long long blocks = g.grid > 0 ? g.grid : ntiles * g.nchunks;
        if (g.xcd) blocks = (blocks + 7) / 8 * 8;
        const int threads = 64 * g.nwy * g.nwx;
        // bytes: rows read per tile (ring + halo, clipped at the frame), planes read per stored plane
        double rows_read = 0, cols_read = 0;
        for (int by = 0; by < g.ty; ++by) { const int y0 = by * UR - g.ry - g.halo, y1 = by * UR - g.ry + TB + g.halo; rows_read += std::min(y1, ny) - std::max(y0, 0); }
        for (int bx = 0; bx < g.tx; ++bx) { const int x0 = bx * UC - g.rl, x1 = bx * UC + UC + g.rl; cols_read += std::min(x1, nx) - std::max(x0, 0); }
        double planes_read;
        if (g.grid > 0) planes_read = nz + (double)(g.grid) * (g.ovl + g.ovh) / ntiles;        // one cut per block boundary (upper bound)
        else { planes_read = 0; for (int ch = 0; ch < g.nchunks; ++ch) { const int zs = ch * g.zc, ze = std::min(nz, zs + g.zc); planes_read += std::min(ze + g.ovh, nz) - std::max(zs - g.ovl, 0); } }
        const double rd = rows_read * cols_read * planes_read * m * 4 / 1e9, wr = (double)n * 4 / 1e9;
        std::vector<float> ms;
        unsigned long long nbad = 0;
        if (rep == 0) {               // the copy must be complete and exact: every voxel stored once, ring / halo values never
            CK(hipMemset(G, 0xFF, n * 4));
            CK(hipMemset(bad, 0, 8));
            dispatch(c, (int)blocks, threads, x, G);
            hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, x, G, n, bad);
            CK(hipMemcpy(&nbad, bad, 8, hipMemcpyDeviceToHost));
        }
        for (int it = 0; it < 7; ++it) {
            CK(hipEventRecord(e0));
            dispatch(c, (int)blocks, threads, x, G);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float t;
            CK(hipEventElapsedTime(&t, e0, e1));
            if (it >= 2) ms.push_back(t);
        }
        CK(hipGetLastError());
        std::sort(ms.begin(), ms.end());
        const double mn = ms.front(), md = ms[ms.size() / 2];
        char geo[64];
        snprintf(geo, sizeof geo, "%dx%d(%dx%d)", TB, TC, UR, UC);
        printf("%-40s %-16s %7lld  %7.3f  %7.3f   %7.3f  %7.3f   %7.3f   %7.3f\n", s.c_str(), geo, blocks, rd, wr, mn, md, 2.0 * n * 4 / 1e9 / mn, (rd + wr) / mn);
        if (nbad) printf("    ^^^^ %llu WRONG VOXELS\n", nbad);
        fflush(stdout);
    }
    return 0;
}
