// deltatest -- does the DISTANCE between the source and the destination of a streaming copy (inside ONE allocation, so that the
// virtual distance is the physical one wherever the allocation is physically contiguous) decide its speed?  (round 4, verdict item 1)
// If the HBM channel / bank mapping makes reads and writes of the same stream position collide for some distances, the rate is a
// periodic function of delta; if the rate does not depend on delta, relative placement is not what the "placement lottery" is about.
//   tools/deltatest [GiB copied = 4]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// persistent blocks, contiguous chunk per block, 1024 threads, 8 x 16 B in flight per lane, non-temporal (tools/bwtest3: the fastest copy)
template <int U, int T>
__global__ __launch_bounds__(T) void k_copy(const f4* __restrict__ a, f4* __restrict__ b, long long per) {
    const long long c0 = (long long)blockIdx.x * per;
    const int tid = threadIdx.x;
    for (long long k = 0; k + U * T <= per; k += U * T) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(a + c0 + k + u * T + tid);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u] * 1.0001f, b + c0 + k + u * T + tid);
    }
}

int main(int argc, char** argv) {
    const long long gib = (argc > 1) ? atoll(argv[1]) : 4;
    const size_t size = (size_t)gib << 30;
    const size_t pool_bytes = 3 * size + (2ull << 30);
    char* pool;
    CK(hipMalloc(&pool, pool_bytes));
    CK(hipMemset(pool, 1, pool_bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("# deltatest: copy %lld GiB -> %lld GiB inside one %.1f GiB allocation at %p; destination = source + %lld GiB + delta\n", gib, gib,
           pool_bytes / 1073741824.0, (void*)pool, gib);
    const int blocks = 256, T = 1024, U = 8;
    long long per = (long long)(size / 16) / blocks;
    per -= per % (U * T);
    const double bytes = (double)per * blocks * 32;
    std::vector<long long> deltas = {0};
    for (long long d = 128; d <= (1ll << 30); d *= 2) deltas.push_back(d);
    for (long long d : {384ll, 4352ll, 36864ll + 256, 1048576ll + 4352, 3ll << 20, 5ll << 20, 96ll << 20, 1ll << 30 | 4352}) deltas.push_back(d);
    deltas.push_back(0);
    for (int pass = 0; pass < 2; ++pass)
        for (long long d : deltas) {
            const f4* a = (const f4*)pool;
            f4* b = (f4*)(pool + size + d);
            std::vector<float> ms;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL((k_copy<U, T>), dim3(blocks), dim3(T), 0, 0, a, b, per);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float t;
                CK(hipEventElapsedTime(&t, e0, e1));
                if (r > 0) ms.push_back(t);
            }
            CK(hipGetLastError());
            std::sort(ms.begin(), ms.end());
            printf("pass %d delta %12lld B  median %7.3f ms %6.0f GB/s  (min %.3f max %.3f)\n", pass, d, ms[ms.size() / 2], bytes / ms[ms.size() / 2] * 1e-6, ms.front(), ms.back());
            fflush(stdout);
        }
    return 0;
}
