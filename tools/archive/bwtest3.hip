// bwtest3.hip -- round 3: settle the mixed read/write ceiling of HBM3E on this pool.
//
// /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; tools/bwtest2 never saw more than 5.15 TB/s.
// This program runs the copy in every form the round-2 verdict lists, on buffers far larger than the 256 MiB Infinity
// Cache (4 GiB in, 4 GiB out by default):
//   * grid-stride copy (the round-2 baseline)                                         "gs"
//   * persistent blocks (1..8 per CU), each owning a CONTIGUOUS chunk, U x 16 B loads in flight per lane "chunk"
//   * the same with non-temporal loads and stores                                      "nt"
//   * the same with SGPR-base buffer_load / buffer_store                               "buf"
//   * per-block read burst -> LDS -> write burst (64 KiB phases)                       "lds"
//   * read-only and write-only twins of the persistent form (the component ceilings)
//   * hipMemcpyAsync device-to-device (the runtime's own blit kernel)
// Output: one line per variant, best and median of 7 timed launches, GB/s counted as bytes read + bytes written.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/bwtest3.hip -o tools/bwtest3
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));

enum { PLAIN = 0, NT = 1, BUF = 2 };

template <int MODE> __device__ __forceinline__ f4 ld(const f4* base, long long i, __amdgpu_buffer_rsrc_t r) {
    if (MODE == NT) return __builtin_nontemporal_load(base + i);
    if (MODE == BUF) {
        i4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(i * 16), 0, 0);
        return __builtin_bit_cast(f4, v);
    }
    return base[i];
}
template <int MODE> __device__ __forceinline__ void st(f4* base, long long i, __amdgpu_buffer_rsrc_t r, f4 v) {
    if (MODE == NT) { __builtin_nontemporal_store(v, base + i); return; }
    if (MODE == BUF) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i4, v), r, (int)(i * 16), 0, 0); return; }
    base[i] = v;
}

// grid-stride copy
template <int U>
__global__ __launch_bounds__(256) void k_gs(const f4* __restrict__ a, f4* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) b[i] = a[i];
}

// persistent chunk copy: block owns [c0, c0 + per) vectors; per trip every lane has U loads in flight, the block moves
// U * T * 16 bytes contiguous (T = threads).  DIR: 0 copy, 1 read only, 2 write only.  PIPE: next trip's loads first.
template <int MODE, int U, int T, int DIR, bool PIPE>
__global__ __launch_bounds__(T) void k_chunk(const f4* __restrict__ a, f4* __restrict__ b, long long per, float* sink) {
    const long long c0 = (long long)blockIdx.x * per;
    const f4* ab = a + c0;
    f4* bb = b + c0;
    // 32-bit offsets inside the chunk (per * 16 < 2^31 is guaranteed by the host)
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (int)(per * 16), 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bb, 0, (int)(per * 16), 0x00020000);
    const int tid = threadIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long trips = per / ((long long)U * T);
    if (!PIPE || DIR != 0) {
        for (long long k = 0; k < trips; ++k) {
            const long long o = k * U * T + tid;
            f4 v[U];
            if (DIR != 2) {
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ld<MODE>(ab, o + u * T, ra);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = f4{(float)k, 1.f, 2.f, (float)u};
            }
            if (DIR != 1) {
#pragma unroll
                for (int u = 0; u < U; ++u) st<MODE>(bb, o + u * T, rb, v[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) acc += v[u];
            }
        }
    } else {
        f4 v[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<MODE>(ab, tid + u * T, ra);
        for (long long k = 0; k < trips; ++k) {
            const long long o = k * U * T + tid;
            if (k + 1 < trips) {
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = ld<MODE>(ab, o + (long long)U * T + u * T, ra);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) st<MODE>(bb, o + u * T, rb, v[u]);
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = w[u];
        }
    }
    if (DIR == 1 && acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// read burst -> LDS -> write burst: the block fills PH KiB of LDS with loads (all issued before any store), then drains it
template <int MODE, int T, int PHKB>
__global__ __launch_bounds__(T) void k_lds(const f4* __restrict__ a, f4* __restrict__ b, long long per) {
    constexpr int NV = PHKB * 1024 / 16;       // vectors per phase
    constexpr int U = NV / T;
    __shared__ f4 buf[NV];
    const long long c0 = (long long)blockIdx.x * per;
    const f4* ab = a + c0;
    f4* bb = b + c0;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)ab, 0, (int)(per * 16), 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)bb, 0, (int)(per * 16), 0x00020000);
    const int tid = threadIdx.x;
    const long long trips = per / NV;
    for (long long k = 0; k < trips; ++k) {
        const long long o = k * NV + tid;
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<MODE>(ab, o + u * T, ra);
#pragma unroll
        for (int u = 0; u < U; ++u) buf[tid + u * T] = v[u];
        __syncthreads();
        // drain: a different thread writes each vector (so the phase really goes through LDS)
#pragma unroll
        for (int u = 0; u < U; ++u) st<MODE>(bb, k * NV + ((tid + 64) % T) + u * T, rb, buf[((tid + 64) % T) + u * T]);
        __syncthreads();
    }
}

// NS read streams + NS write streams (the one-sweep CP kernel's shape: one vector from each of ~10 arrays per frame, then ~10
// stores): stream k of block b is the region [k * sper + b * per, ...) of a / b.  LDM / STM: PLAIN or NT.
template <int LDM, int STM, int NS, int T, int UU>
__global__ __launch_bounds__(T) void k_streams(const f4* __restrict__ a, f4* __restrict__ b, long long per, long long sper) {
    const long long c0 = (long long)blockIdx.x * per;
    const int tid = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, 0, 0, 0x00020000);   // unused by PLAIN / NT
    for (long long k = 0; k + UU * T <= per; k += UU * T) {
        f4 v[NS][UU];
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) v[sidx][u] = ld<LDM>(a + sidx * sper + c0, k + u * T + tid, r);
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) st<STM>(b + sidx * sper + c0, k + u * T + tid, r, v[sidx][u] * 1.0001f);
    }
}

static hipEvent_t e0, e1;
template <typename F> static void run(const char* name, double bytes, F&& launch) {
    std::vector<float> ms;
    launch();
    CK(hipDeviceSynchronize());
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("%-64s best %7.3f ms %6.0f GB/s | median %7.3f ms %6.0f GB/s\n", name, ms[0], bytes / ms[0] * 1e-6, ms[3], bytes / ms[3] * 1e-6);
    fflush(stdout);
}

template <int MODE, int U, int T, int DIR, bool PIPE>
static void chunk(const char* tag, const f4* a, f4* b, long long n, int bpc, float* sink) {
    const int blocks = 256 * bpc;
    long long per = n / blocks;
    per -= per % ((long long)U * T);
    char name[128];
    snprintf(name, sizeof name, "%s %s U=%-2d T=%-4d blocks/CU=%d%s", DIR == 0 ? "copy " : (DIR == 1 ? "read " : "write"), tag, U, T, bpc, PIPE ? " pipe" : "");
    const double bytes = (double)per * blocks * 16 * (DIR == 0 ? 2 : 1);
    run(name, bytes, [&] { hipLaunchKernelGGL((k_chunk<MODE, U, T, DIR, PIPE>), dim3(blocks), dim3(T), 0, 0, a, b, per, sink); });
}
template <int MODE, int T, int PHKB>
static void ldsrun(const char* tag, const f4* a, f4* b, long long n, int bpc) {
    const int blocks = 256 * bpc;
    long long per = n / blocks;
    per -= per % (PHKB * 1024 / 16);
    char name[128];
    snprintf(name, sizeof name, "copy  lds-phase %s %d KiB T=%-4d blocks/CU=%d", tag, PHKB, T, bpc);
    run(name, (double)per * blocks * 32, [&] { hipLaunchKernelGGL((k_lds<MODE, T, PHKB>), dim3(blocks), dim3(T), 0, 0, a, b, per); });
}

template <int LDM, int STM, int NS, int T, int UU>
static void streams(const f4* a, f4* b, long long n, int bpc) {
    const int blocks = 256 * bpc;
    const long long sper = n / NS;
    long long per = sper / blocks;
    per -= per % ((long long)UU * T);
    char name[128];
    snprintf(name, sizeof name, "mix %d R + %d W streams, ld %s st %s, T=%-4d U=%d blocks/CU=%d", NS, NS, LDM == NT ? "nt" : "plain", STM == NT ? "nt" : "plain", T, UU, bpc);
    run(name, (double)per * blocks * NS * 32, [&] { hipLaunchKernelGGL((k_streams<LDM, STM, NS, T, UU>), dim3(blocks), dim3(T), 0, 0, a, b, per, sper); });
}

int main(int argc, char** argv) {
    const long long gib = (argc > 1) ? atoll(argv[1]) : 4;
    const long long n = gib * (1ll << 30) / 16;       // vectors
    f4 *a, *b;
    float* sink;
    CK(hipMalloc(&a, n * 16));
    CK(hipMalloc(&b, n * 16));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, n * 16));
    CK(hipMemset(b, 0, n * 16));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# bwtest3: %lld GiB in + %lld GiB out, 256 CUs; GB/s = (bytes read + bytes written) / time\n", gib, gib);
    if (argc > 2) {       // second study: load / store policy mix and the many-stream shape of the CP sweep
        for (int bpc : {1, 2}) {
            streams<PLAIN, PLAIN, 8, 512, 1>(a, b, n, bpc);
            streams<NT, PLAIN, 8, 512, 1>(a, b, n, bpc);
            streams<NT, NT, 8, 512, 1>(a, b, n, bpc);
            streams<PLAIN, PLAIN, 8, 1024, 1>(a, b, n, bpc);
            streams<NT, PLAIN, 8, 1024, 1>(a, b, n, bpc);
            streams<NT, NT, 8, 1024, 1>(a, b, n, bpc);
            streams<NT, PLAIN, 8, 512, 2>(a, b, n, bpc);
            streams<NT, NT, 8, 512, 2>(a, b, n, bpc);
            streams<NT, PLAIN, 8, 256, 2>(a, b, n, bpc);
            streams<NT, PLAIN, 2, 1024, 4>(a, b, n, bpc);
            streams<NT, NT, 2, 1024, 4>(a, b, n, bpc);
            streams<NT, PLAIN, 1, 1024, 8>(a, b, n, bpc);
            streams<PLAIN, NT, 1, 1024, 8>(a, b, n, bpc);
        }
        streams<NT, PLAIN, 8, 256, 1>(a, b, n, 4);
        streams<PLAIN, PLAIN, 8, 256, 1>(a, b, n, 4);
        streams<NT, NT, 8, 256, 1>(a, b, n, 4);
        return 0;
    }
    run("hipMemcpyAsync D2D", (double)n * 32, [&] { CK(hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0)); });
    run("copy  grid-stride U=4 blocks=8192", (double)n * 32, [&] { hipLaunchKernelGGL((k_gs<4>), dim3(8192), dim3(256), 0, 0, a, b, n); });
    run("copy  grid-stride U=8 blocks=65536", (double)n * 32, [&] { hipLaunchKernelGGL((k_gs<8>), dim3(65536), dim3(256), 0, 0, a, b, n); });
    for (int bpc : {1, 2, 4, 8}) {
        chunk<PLAIN, 8, 256, 0, false>("plain", a, b, n, bpc, sink);
        chunk<NT, 8, 256, 0, false>("nt   ", a, b, n, bpc, sink);
        chunk<BUF, 8, 256, 0, false>("buf  ", a, b, n, bpc, sink);
    }
    for (int bpc : {1, 2}) {
        chunk<PLAIN, 8, 1024, 0, false>("plain", a, b, n, bpc, sink);
        chunk<NT, 8, 1024, 0, false>("nt   ", a, b, n, bpc, sink);
        chunk<NT, 16, 512, 0, false>("nt   ", a, b, n, bpc, sink);
        chunk<BUF, 16, 512, 0, false>("buf  ", a, b, n, bpc, sink);
        chunk<NT, 8, 512, 0, true>("nt   ", a, b, n, bpc, sink);
        chunk<BUF, 8, 512, 0, true>("buf  ", a, b, n, bpc, sink);
    }
    for (int bpc : {2, 4}) {
        chunk<PLAIN, 4, 256, 0, false>("plain", a, b, n, bpc, sink);
        chunk<NT, 4, 256, 0, true>("nt   ", a, b, n, bpc, sink);
        chunk<PLAIN, 16, 256, 0, false>("plain", a, b, n, bpc, sink);
        chunk<NT, 16, 256, 0, false>("nt   ", a, b, n, bpc, sink);
    }
    for (int bpc : {1, 2}) {
        ldsrun<PLAIN, 512, 64>("plain", a, b, n, bpc);
        ldsrun<NT, 512, 64>("nt   ", a, b, n, bpc);
        ldsrun<BUF, 512, 64>("buf  ", a, b, n, bpc);
        ldsrun<NT, 1024, 64>("nt   ", a, b, n, bpc);
        ldsrun<NT, 256, 32>("nt   ", a, b, n, bpc);
    }
    ldsrun<NT, 256, 32>("nt   ", a, b, n, 4);
    for (int bpc : {2, 4, 8}) {
        chunk<PLAIN, 8, 256, 1, false>("plain", a, b, n, bpc, sink);
        chunk<NT, 8, 256, 1, false>("nt   ", a, b, n, bpc, sink);
        chunk<PLAIN, 8, 256, 2, false>("plain", a, b, n, bpc, sink);
        chunk<NT, 8, 256, 2, false>("nt   ", a, b, n, bpc, sink);
    }
    // smaller footprints: does the guide's figure live in the Infinity Cache?
    for (long long mib : {64ll, 128ll, 256ll, 512ll, 1024ll}) {
        const long long m = mib * (1ll << 20) / 16;
        char name[96];
        snprintf(name, sizeof name, "copy  grid-stride U=4 blocks=8192, %lld MiB in + out", mib);
        run(name, (double)m * 32, [&] { hipLaunchKernelGGL((k_gs<4>), dim3(8192), dim3(256), 0, 0, a, b, m); });
        snprintf(name, sizeof name, "copy  chunk nt U=8 T=256 blocks/CU=4, %lld MiB in + out", mib);
        long long per = m / 1024;
        per -= per % (8 * 256);
        run(name, (double)per * 1024 * 32, [&] { hipLaunchKernelGGL((k_chunk<NT, 8, 256, 0, false>), dim3(1024), dim3(256), 0, 0, a, b, per, sink); });
    }
    return 0;
}
