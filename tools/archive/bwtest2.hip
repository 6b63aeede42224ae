// bwtest2.hip -- does the ORDER of loads and stores inside a wave set the practical HBM ceiling?
//
// On gfx9 / CDNA a wave has ONE in-order counter (vmcnt) for vector loads and stores.  A load whose result is needed
// while older stores are still in flight can only be waited for with a count that also covers those stores, so the
// pattern "load(t) -> compute -> store(t) -> load(t+1) -> use" exposes one full store round trip per step.
// tools/bwtest.hip (round 1) measured its ceilings with exactly that pattern (copy 5.1 TB/s, the CP mix 5.15 TB/s).
// Here the same traffic is issued with the loads of step t+1 BEFORE the stores of step t (software pipelining).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bwtest2.hip -o tools/bwtest2 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float float4_ __attribute__((ext_vector_type(4)));

// ---- copy: U independent 16-byte loads per trip; PIPE: the next trip's loads are issued before this trip's stores
template <int U, bool PIPE>
__global__ __launch_bounds__(256) void k_copy(const float4_* __restrict__ a, float4_* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (!PIPE) {
        for (; i + (U - 1) * stride < n; i += U * stride) {
            float4_ v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
            for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
        }
    } else {
        float4_ v[U], w[U];
        if (i + (U - 1) * stride < n) {
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
        }
        for (; i + (U - 1) * stride < n; i += U * stride) {
            const long long j = i + U * stride;
            const bool more = j + (U - 1) * stride < n;
            if (more) {
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = a[j + u * stride];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = w[u];
        }
    }
    for (; i < n; i += stride) b[i] = a[i];
}

// ---- the CP mix: block (64,4) owns a 4-row x 256-col tile, marches z in a chunk, M frames; per (z, t): one x vector and
// NC channel vectors are read, NC channel vectors written (+ NX extra image streams read and written, like x0 / p / x_out).
// PIPE 0: loads(t) -> stores(t) -> loads(t+1) ...   (what the kernels of round 1 do)
// PIPE 1: loads(t+1) are issued before stores(t)     (double-buffered registers)
// PIPE 2: stores(t) are issued after loads(t+1) by DELAYING them one frame (same register cost, no prologue)
template <int M, int NC, int PIPE>
__global__ __launch_bounds__(256, 2) void k_march(const float* __restrict__ x, float* __restrict__ q, int nz, int ny, int nx, int zchunk) {
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int lane = tid % 64, ty = tid / 64;
    const int tiles_x = nx / 256;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const long long inpl = (long long)(by * 4 + ty) * nx + (bx * 64 + lane) * 4;
    const long long s_t = (long long)ny * nx, s_z = s_t * M, s_dz = s_z * NC;
    const int zs = blockIdx.y * zchunk, ze = min(zs + zchunk, nz);
    auto ldx = [&](int z, int t) { return *(const float4_*)(x + (long long)z * s_z + t * s_t + inpl); };
    auto qp = [&](int z, int t, int c) { return (float4_*)(q + (long long)z * s_dz + c * s_z + t * s_t + inpl); };
    if (PIPE == 0) {
        for (int z = zs; z < ze; ++z) {
#pragma unroll
            for (int t = 0; t < M; ++t) {
                const float4_ xv = ldx(z, t);
                float4_ v[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = *qp(z, t, c);
#pragma unroll
                for (int c = 0; c < NC; ++c) *qp(z, t, c) = v[c] * 1.0001f + xv;
            }
        }
    } else if (PIPE == 1) {
        float4_ v[NC], w[NC], xv, xw;
        xv = ldx(zs, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = *qp(zs, 0, c);
        for (int z = zs; z < ze; ++z) {
#pragma unroll
            for (int t = 0; t < M; ++t) {
                const int tn = (t + 1 < M) ? t + 1 : 0, zn = (t + 1 < M) ? z : z + 1;
                if (zn < ze) {
                    xw = ldx(zn, tn);
#pragma unroll
                    for (int c = 0; c < NC; ++c) w[c] = *qp(zn, tn, c);
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) *qp(z, t, c) = v[c] * 1.0001f + xv;
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = w[c];
                xv = xw;
            }
        }
    } else {
        float4_ pend[NC];
        bool have = false;
        int pz = 0, pt = 0;
        for (int z = zs; z < ze; ++z) {
#pragma unroll
            for (int t = 0; t < M; ++t) {
                const float4_ xv = ldx(z, t);
                float4_ v[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = *qp(z, t, c);
                if (have) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) *qp(pz, pt, c) = pend[c];
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) pend[c] = v[c] * 1.0001f + xv;
                have = true; pz = z; pt = t;
            }
        }
        if (have) {
#pragma unroll
            for (int c = 0; c < NC; ++c) *qp(pz, pt, c) = pend[c];
        }
    }
}


// ---- does the TILE SHAPE matter?  wave tile = ROWS rows x (64 / ROWS) lanes, WAVES waves side by side per block; same traffic
// as k_march<8, 8, 0> (1 + NC read, NC written per (z, t))
// EDGE: the first / last lane of a row segment also loads the 4-byte element on the other side of the segment border (what the
// TV kernels do for their column neighbours) -- same exact byte count, used to calibrate FETCH_SIZE for narrow requests
template <int M, int NC, int ROWS, int WAVES, int EDGE = 0>
__global__ __launch_bounds__(64 * WAVES, (WAVES >= 8) ? 2 : 2) void k_march_tile(const float* __restrict__ x, float* __restrict__ q, int nz, int ny, int nx, int zchunk) {
    constexpr int TL = 64 / ROWS;
    const int lane = threadIdx.x, wave = threadIdx.y;
    const int row = lane / TL, lx = lane % TL;
    const int tiles_x = nx / (WAVES * TL * 4);
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const long long inpl = (long long)(by * ROWS + row) * nx + (bx * WAVES * TL + wave * TL + lx) * 4;
    const long long s_t = (long long)ny * nx, s_z = s_t * M, s_dz = s_z * NC;
    const int zs = blockIdx.y * zchunk, ze = min(zs + zchunk, nz);
    for (int z = zs; z < ze; ++z) {
#pragma unroll
        for (int t = 0; t < M; ++t) {
            float4_ xv = *(const float4_*)(x + (long long)z * s_z + t * s_t + inpl);
            if (EDGE) {
                const int col = (bx * WAVES * TL + wave * TL + lx) * 4;
                const bool le = (lx == 0) && col > 0, re = (lx == TL - 1) && col + 4 < nx;
                if (le || re) xv.x += x[(long long)z * s_z + t * s_t + inpl + (le ? -1 : 4)];
            }
            float4_ v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = *(const float4_*)(q + (long long)z * s_dz + c * s_z + t * s_t + inpl);
#pragma unroll
            for (int c = 0; c < NC; ++c) *(float4_*)(q + (long long)z * s_dz + c * s_z + t * s_t + inpl) = v[c] * 1.0001f + xv;
        }
    }
}

// ---- read-dominated (tv_DT): NC channel vectors read, one image vector written; LATE: the store is issued after the next
// frame's loads
template <int M, int NC, bool LATE>
__global__ __launch_bounds__(256, 3) void k_dtload(const float* __restrict__ q, float* __restrict__ x, int nz, int ny, int nx, int zchunk) {
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int lane = tid % 64, ty = tid / 64;
    const int tiles_x = nx / 256;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const long long inpl = (long long)(by * 4 + ty) * nx + (bx * 64 + lane) * 4;
    const long long s_t = (long long)ny * nx, s_z = s_t * M, s_dz = s_z * NC;
    const int zs = blockIdx.y * zchunk, ze = min(zs + zchunk, nz);
    auto qp = [&](int z, int t, int c) { return (const float4_*)(q + (long long)z * s_dz + c * s_z + t * s_t + inpl); };
    auto xq = [&](int z, int t) { return (float4_*)(x + (long long)z * s_z + t * s_t + inpl); };
    float4_ pend = {0, 0, 0, 0};
    float4_* pp = nullptr;
    for (int z = zs; z < ze; ++z) {
#pragma unroll
        for (int t = 0; t < M; ++t) {
            float4_ v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = *qp(z, t, c);
            if (LATE && pp != nullptr) *pp = pend;
            float4_ r = v[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) r += v[c] * (float)c;
            if (LATE) { pend = r; pp = xq(z, t); } else *xq(z, t) = r;
        }
    }
    if (LATE && pp != nullptr) *pp = pend;
}

// ---- write-dominated (tv_D): one x vector read, NC channel vectors written; PIPE: x of the next plane requested a plane ahead
template <int M, int NC, bool PIPE>
__global__ __launch_bounds__(256, 3) void k_dstore(const float* __restrict__ x, float* __restrict__ q, int nz, int ny, int nx, int zchunk) {
    const int tid = threadIdx.y * 64 + threadIdx.x;
    const int lane = tid % 64, ty = tid / 64;
    const int tiles_x = nx / 256;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const long long inpl = (long long)(by * 4 + ty) * nx + (bx * 64 + lane) * 4;
    const long long s_t = (long long)ny * nx, s_z = s_t * M, s_dz = s_z * NC;
    const int zs = blockIdx.y * zchunk, ze = min(zs + zchunk, nz);
    auto ldx = [&](int z, int t) { return *(const float4_*)(x + (long long)z * s_z + t * s_t + inpl); };
    auto qp = [&](int z, int t, int c) { return (float4_*)(q + (long long)z * s_dz + c * s_z + t * s_t + inpl); };
    float4_ C[M];
#pragma unroll
    for (int t = 0; t < M; ++t) C[t] = ldx(zs, t);
    for (int z = zs; z < ze; ++z) {
#pragma unroll
        for (int t = 0; t < M; ++t) {
            const float4_ c = C[t];
            float4_ nxt = c;
            if (PIPE) { if (z + 1 < ze) nxt = ldx(z + 1, t); }          // consumed one plane later
            const float4_ other = C[(t + 1 < M) ? t + 1 : t];
#pragma unroll
            for (int k = 0; k < NC; ++k) *qp(z, t, k) = (other - c) * (float)(k + 1);
            if (!PIPE) { if (z + 1 < ze) nxt = ldx(z + 1, t); }         // issued behind the stores
            C[t] = nxt;
        }
    }
}

template <typename F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
    const int nz = argc > 1 ? atoi(argv[1]) : 128, M = 8, ny = 1024, nx = 1024, NC = 8;
    const long long V = (long long)nz * M * ny * nx;
    float *x, *q;
    CK(hipMalloc(&x, V * 4)); CK(hipMalloc(&q, V * 4 * NC));
    CK(hipMemset(x, 0, V * 4)); CK(hipMemset(q, 0, V * 4 * NC));
    printf("V=%lld voxels, x %.1f GiB, q %.1f GiB\n", V, V * 4 / 1073741824.0, V * 4.0 * NC / 1073741824.0);
    const long long n4 = V * NC / 4 / 2;
    float ms;
#define COPY(U, P, NB) ms = timeit([&] { hipLaunchKernelGGL((k_copy<U, P>), dim3(NB), dim3(256), 0, 0, (const float4_*)q, (float4_*)q + n4, n4); }); \
    printf("copy  U=%d pipe=%d blocks %6d          : %7.2f ms  %5.0f GB/s (r+w)\n", U, (int)P, NB, ms, 2.0 * n4 * 16 / ms / 1e6);
    COPY(1, false, 8192) COPY(4, false, 8192) COPY(8, false, 8192) COPY(4, true, 8192) COPY(8, true, 8192)
    COPY(4, true, 2048) COPY(8, true, 2048) COPY(4, false, 65536) COPY(4, true, 65536)
    const double b_rw = (1.0 + 2 * NC) * 4 * V, b_w = (1.0 + NC) * 4 * V;
    for (int zc : {16, 32}) {
        dim3 grid((nx / 256) * (ny / 4), (nz + zc - 1) / zc), blk(64, 4);
        ms = timeit([&] { hipLaunchKernelGGL((k_march<8, 8, 0>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); });
        printf("march 1+8 read 8 write, zchunk %2d, loads behind stores : %7.2f ms  %5.0f GB/s\n", zc, ms, b_rw / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_march<8, 8, 1>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); });
        printf("march 1+8 read 8 write, zchunk %2d, next loads first    : %7.2f ms  %5.0f GB/s\n", zc, ms, b_rw / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_march<8, 8, 2>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); });
        printf("march 1+8 read 8 write, zchunk %2d, stores delayed      : %7.2f ms  %5.0f GB/s\n", zc, ms, b_rw / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_dstore<8, 8, false>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); });
        printf("dstore 1 read 8 write,   zchunk %2d, loads behind stores : %7.2f ms  %5.0f GB/s\n", zc, ms, b_w / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_dstore<8, 8, true>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); });
        printf("dstore 1 read 8 write,   zchunk %2d, next plane first    : %7.2f ms  %5.0f GB/s\n", zc, ms, b_w / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_dtload<8, 8, false>), grid, blk, 0, 0, q, x, nz, ny, nx, zc); });
        printf("dtload 8 read 1 write,   zchunk %2d, store at once       : %7.2f ms  %5.0f GB/s\n", zc, ms, b_w / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_dtload<8, 8, true>), grid, blk, 0, 0, q, x, nz, ny, nx, zc); });
        printf("dtload 8 read 1 write,   zchunk %2d, store after loads   : %7.2f ms  %5.0f GB/s\n", zc, ms, b_w / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_dstore<8, 4, false>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); });
        printf("dstore 1 read 4 write,   zchunk %2d                      : %7.2f ms  %5.0f GB/s\n", zc, ms, 5.0 * 4 * V / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((k_dstore<8, 1, false>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); });
        printf("dstore 1 read 1 write,   zchunk %2d                      : %7.2f ms  %5.0f GB/s\n", zc, ms, 2.0 * 4 * V / ms / 1e6);
    }
        {
        const int zc = 32;
#define TILE(R, W) { dim3 grid((nx / (W * (64 / R) * 4)) * (ny / R), (nz + zc - 1) / zc), blk(64, W); \
        ms = timeit([&] { hipLaunchKernelGGL((k_march_tile<8, 8, R, W>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); }); \
        printf("march 1+8 read 8 write, wave tile %2d rows x %2d lanes, %d waves/block (block tile %2d x %4d cols): %7.2f ms  %5.0f GB/s\n", R, 64 / R, W, R, W * (64 / R) * 4, ms, b_rw / ms / 1e6); }
#define TILE_E(R, W) { dim3 grid((nx / (W * (64 / R) * 4)) * (ny / R), (nz + zc - 1) / zc), blk(64, W); \
        ms = timeit([&] { hipLaunchKernelGGL((k_march_tile<8, 8, R, W, 1>), grid, blk, 0, 0, x, q, nz, ny, nx, zc); }); \
        printf("the same + 4-byte edge loads, wave tile %2d rows x %2d lanes, %d waves/block: %7.2f ms  %5.0f GB/s\n", R, 64 / R, W, ms, b_rw / ms / 1e6); }
        TILE_E(8, 8) TILE_E(4, 4)
        TILE(4, 4) TILE(8, 4) TILE(8, 8) TILE(4, 8) TILE(16, 8) TILE(16, 16) TILE(2, 4) TILE(1, 4) TILE(8, 16)
    }
    return 0;
}
