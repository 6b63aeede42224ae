// bwtest4 -- the one-sweep CP kernel's MEMORY SHAPE without its arithmetic, with run-time pitches (round 4).
//
// Round-3 verdict: tools/bwtest3's "8 R + 8 W" mix places every stream of every block congruent modulo 2 MiB, and so does the real
// sweep (the 64 frames of a q plane sit at exactly 4 MiB strides, x / x0 / p frames too).  If the HBM channel / bank hash does not
// mix the high address bits, all ~20 streams of a block camp on the same channels.  This tool walks the sweep's exact footprint --
// block = 8 rows x 1 KiB of a frame (8 waves side by side, wave = 8 rows x 128 B, as k_cp_fused) marching z inside a 32-plane chunk,
// per frame t: read x(z+1,t), x0(z-1,t), p(z-1,t), q(z,0..7,t); write q(z,0..7,t), x_out(z-1,t), p(z-1,t) -- and takes
//     frame pad (bytes added to the 4 MiB frame pitch), row pad (bytes added to the 4 KiB row pitch), array stagger (bytes between arrays)
// as arguments, so that "does de-aliasing the streams move the ceiling" is ONE run on ONE box with everything else equal.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bwtest4 tools/bwtest4.hip
//   tools/bwtest4 [nz=128] [reps=5]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int M = 8, ND = 8, NY = 1024, NXB = 4096;      // frames, channels, rows, bytes per row (1024 floats)

struct Geo {
    long long rp, fp;          // row pitch, frame pitch (bytes)
    long long xz, qz;          // plane stride of x-like arrays (M frames) and of q (ND * M frames), bytes
    int nz, zchunk;
};

template <bool NT> __device__ __forceinline__ f4 ld(const char* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4*>(p));
    return *reinterpret_cast<const f4*>(p);
}
template <bool NT> __device__ __forceinline__ void st(char* p, f4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p));
    else *reinterpret_cast<f4*>(p) = v;
}

// MAP 0: the sweep's mapping (wave = 8 rows x 8 lanes of 16 B, 8 waves side by side = 8 rows x 1 KiB)
// MAP 1: wave = 1 row x 1 KiB (64 lanes x 16 B), 8 waves stacked = the same 8 rows x 1 KiB block tile
// LDSKB: kilobytes of LDS reserved per block (128: one block per CU like the real sweep, whose R / U slots take 128 KiB; 0: as many
// blocks per CU as the registers allow)
template <int MAP, bool NT, int LDSKB>
__global__ __launch_bounds__(512, 2) void k_sweep(Geo g, const char* __restrict__ x, const char* __restrict__ x0, char* __restrict__ p,
                                                   char* q, char* __restrict__ xo, char* qo) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    __shared__ float lds_pad[LDSKB > 0 ? LDSKB * 256 : 1];
    if (g.nz < 0) lds_pad[tid] = 1.f;            // never true: keeps the allocation
    const int tiles_x = NXB / 1024;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    int row, colb;
    if (MAP == 0) { row = lane >> 3; colb = bx * 1024 + wave * 128 + (lane & 7) * 16; }
    else { row = wave; colb = bx * 1024 + lane * 16; }
    const long long inpl = (long long)(by * 8 + row) * g.rp + colb;
    const int zs = blockIdx.y * g.zchunk, ze = min(zs + g.zchunk, g.nz);
    f4 carry[M];
#pragma unroll
    for (int t = 0; t < M; ++t) carry[t] = ld<false>(x + (long long)zs * g.xz + t * g.fp + inpl);
    for (int z = zs; z < ze; ++z) {
        const int zn = (z + 1 < g.nz) ? z + 1 : z, zp = (z > 0) ? z - 1 : 0;
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const f4 xn = ld<false>(x + (long long)zn * g.xz + t * g.fp + inpl);
            f4 qv[ND];
            const char* qb = q + (long long)z * g.qz + t * g.fp + inpl;
#pragma unroll
            for (int c = 0; c < ND; ++c) qv[c] = ld<NT>(qb + (long long)c * M * g.fp);
            const long long po = (long long)zp * g.xz + t * g.fp + inpl;
            const f4 x0v = ld<NT>(x0 + po), pv = ld<NT>(p + po);
            f4 s = carry[t] + xn;
#pragma unroll
            for (int c = 0; c < ND; ++c) {
                s += qv[c];
                st<NT>(qo + (qb - q) + (long long)c * M * g.fp, qv[c] * 1.0001f + xn);      // qo == q: in place; else ping-pong
            }
            st<NT>(p + po, pv + x0v);
            st<NT>(xo + po, s + pv);
            carry[t] = xn;
        }
    }
}

static hipEvent_t e0, e1;
template <typename F> static double run(F&& launch, int reps, std::vector<float>* all = nullptr) {
    std::vector<float> ms;
    launch();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    CK(hipGetLastError());
    if (all) *all = ms;
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
    const int nz = (argc > 1) ? atoi(argv[1]) : 128;
    const int reps = (argc > 2) ? atoi(argv[2]) : 5;
    const long long maxpad_f = 80 * 1024, maxpad_r = 256, maxstag = 4ll << 20;
    const long long fpmax = (long long)NY * (NXB + maxpad_r) + maxpad_f;
    const long long xbytes = (long long)nz * M * fpmax + maxstag, qbytes = (long long)nz * ND * M * fpmax + maxstag;
    char *pool;
    const bool pingpong = (argc > 3);                 // third argument: also run the q ping-pong variant (needs a second q array)
    const long long total = 4 * xbytes + (pingpong ? 2 : 1) * qbytes + 6 * maxstag;
    CK(hipMalloc(&pool, total));
    CK(hipMemset(pool, 0, total));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# bwtest4: sweep-shaped traffic, nz = %d planes x %d frames x %d rows x %d B; algorithmic bytes = (5 + 2 Nd) x 4 B/voxel = %.2f GB per launch\n",
           nz, M, NY, NXB, 84.0 * nz * M * NY * (NXB / 4) * 1e-9);
    printf("# pool %.1f GiB at %p\n", total / 1073741824.0, (void*)pool);
    const double bytes = 84.0 * nz * M * NY * (NXB / 4);
    struct Case { const char* name; long long fpad, rpad, stag; };
    const Case cases[] = {
        {"dense (4 MiB frames, 4 KiB rows, arrays back to back)", 0, 0, 0},
        {"frame pad 4352 B", 4352, 0, 0},
        {"frame pad 4 KiB + 256 B x 17 = 8448 B", 8448, 0, 0},
        {"frame pad 36 KiB + 256", 36 * 1024 + 256, 0, 0},
        {"frame pad 68 KiB + 768", 68 * 1024 + 768, 0, 0},
        {"frame pad 16 KiB", 16 * 1024, 0, 0},
        {"frame pad 64 KiB", 64 * 1024, 0, 0},
        {"row pad 128 B", 0, 128, 0},
        {"row pad 256 B", 0, 256, 0},
        {"row pad 128 B + frame pad 4352 B", 4352, 128, 0},
        {"arrays staggered by 1 MiB + 4352 B (dense frames)", 0, 0, (1ll << 20) + 4352},
        {"dense again", 0, 0, 0},
    };
    // variants: 0 = sweep mapping, nt, one block per CU (the real kernel's shape); 1 = the same with plain loads / stores;
    // 2 = wave-per-row mapping, nt, one block per CU; 3 = sweep mapping, nt, no LDS reservation (2 blocks per CU)
    // 4 (with a third argument) = variant 0 with q written to a SECOND array (ping-pong) instead of in place
    for (int var = 0; var < (pingpong ? 5 : 4); ++var) {
            const int map = (var == 2) ? 1 : 0, nt = (var != 1);
            for (const Case& c : cases) {
                Geo g;
                g.rp = NXB + c.rpad;
                g.fp = (long long)NY * g.rp + c.fpad;
                g.xz = (long long)M * g.fp;
                g.qz = (long long)ND * M * g.fp;
                g.nz = nz;
                g.zchunk = 32;
                char* base = pool;
                char* x = base; base += xbytes + c.stag;
                char* x0 = base; base += xbytes + c.stag;
                char* p = base; base += xbytes + c.stag;
                char* xo = base; base += xbytes + c.stag;
                char* q = base; base += qbytes + c.stag;
                char* qo = (var == 4) ? base : q;
                const dim3 grid(NXB / 1024 * NY / 8, (nz + 31) / 32), blk(512);
                std::vector<float> all;
                double ms;
                if (var == 0 || var == 4) ms = run([&] { hipLaunchKernelGGL((k_sweep<0, true, 128>), grid, blk, 0, 0, g, x, x0, p, q, xo, qo); }, reps, &all);
                else if (var == 1) ms = run([&] { hipLaunchKernelGGL((k_sweep<0, false, 128>), grid, blk, 0, 0, g, x, x0, p, q, xo, qo); }, reps, &all);
                else if (var == 2) ms = run([&] { hipLaunchKernelGGL((k_sweep<1, true, 128>), grid, blk, 0, 0, g, x, x0, p, q, xo, qo); }, reps, &all);
                else ms = run([&] { hipLaunchKernelGGL((k_sweep<0, true, 0>), grid, blk, 0, 0, g, x, x0, p, q, xo, qo); }, reps, &all);
                printf("var %d map %d %s  %-58s median %7.3f ms %6.0f GB/s  (", var, map, nt ? "nt   " : "plain", c.name, ms, bytes / ms * 1e-6);
                (void)map;
                for (float v : all) printf(" %.2f", v);
                printf(" )\n");
                fflush(stdout);
            }
        }
    return 0;
}
