// bwtest5.hip -- round 4: the ceiling of the tv_D memory pattern: ONE read stream (the image) and NW write streams (the gradient channels, one
// plane-strided array each), 16-byte lanes, whole 128-byte lines.  tv_D hybrid (1 + 8 words per voxel) runs at 0.63 of 8 TB/s = 5.0 TB/s;
// a pure fill reaches 6.7 TB/s (profiles/r1_bwtest_ceilings.txt).  What does 1 R + 8 W reach without any stencil?
// build: hipcc -O3 --offload-arch=gfx950 -o tools/bwtest5 tools/bwtest5.hip      run: tools/bwtest5 [Mvoxels=537]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ void st(f4* p, f4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <int NT> __device__ __forceinline__ f4 ld(const f4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// every block owns a contiguous piece of `per` vectors of the image; U vectors per thread and step
template <int NW, int T, int U, int LNT, int SNT>
__global__ __launch_bounds__(T) void k_fan(const f4* __restrict__ x, f4* __restrict__ out, long long per, long long sstride) {
    const long long c0 = (long long)blockIdx.x * per;
    for (long long k = threadIdx.x; k < per; k += (long long)U * T) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<LNT>(x + c0 + k + (long long)u * T);
#pragma unroll
        for (int s = 0; s < NW; ++s)
#pragma unroll
            for (int u = 0; u < U; ++u) st<SNT>(out + (long long)s * sstride + c0 + k + (long long)u * T, v[u] * (1.0f + s));
    }
}

static hipEvent_t e0, e1;
template <int NW, int T, int U, int LNT, int SNT> static void run(const f4* x, f4* out, long long n, int bpc) {
    const int blocks = 256 * bpc;
    long long per = n / blocks;
    per -= per % ((long long)U * T);
    std::vector<float> ms;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_fan<NW, T, U, LNT, SNT>), dim3(blocks), dim3(T), 0, 0, x, out, per, n);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)per * blocks * 16 * (1 + NW);
    printf("1 R + %d W  T=%-4d U=%d blocks/CU=%-2d ld %-5s st %-5s  best %7.3f ms %6.0f GB/s (%.3f of 8 TB/s) | median %7.3f ms\n", NW, T, U, bpc, LNT ? "nt" : "plain",
           SNT ? "nt" : "plain", ms[0], bytes / ms[0] * 1e-6, bytes / ms[0] * 1e-6 / 8000.0, ms[ms.size() / 2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const long long mv = (argc > 1) ? atoll(argv[1]) : 537;
    const long long n = mv * 1000000ll / 4 / 65536 * 65536;      // vectors
    f4 *x, *out;
    CK(hipMalloc(&x, n * 16));
    CK(hipMalloc(&out, n * 16 * 8));
    CK(hipMemset(x, 0, n * 16));
    CK(hipMemset(out, 0, n * 16 * 8));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# bwtest5: %lld Mvoxel image (%.2f GB), the tv_D pattern without a stencil\n", mv, n * 16 / 1e9);
    for (int rep = 0; rep < 2; ++rep) {
        run<8, 256, 1, 0, 0>(x, out, n, 8);
        run<8, 256, 1, 0, 1>(x, out, n, 8);
        run<8, 256, 2, 0, 1>(x, out, n, 8);
        run<8, 256, 4, 0, 1>(x, out, n, 4);
        run<8, 512, 1, 0, 1>(x, out, n, 4);
        run<8, 512, 2, 0, 1>(x, out, n, 4);
        run<8, 1024, 1, 0, 1>(x, out, n, 2);
        run<8, 1024, 2, 0, 1>(x, out, n, 2);
        run<8, 1024, 2, 1, 1>(x, out, n, 2);
        run<8, 1024, 4, 0, 1>(x, out, n, 1);
        run<8, 1024, 1, 0, 0>(x, out, n, 2);
        run<8, 256, 1, 0, 1>(x, out, n, 32);
        run<8, 256, 1, 0, 1>(x, out, n, 128);
        run<4, 256, 1, 0, 1>(x, out, n, 8);
        run<4, 1024, 2, 0, 1>(x, out, n, 2);
        run<1, 1024, 2, 0, 1>(x, out, n, 2);
    }
    return 0;
}
