// coherence_probe.hip -- round 6: how can blocks on DIFFERENT XCDs (each XCD has its own L2) exchange data inside one persistent kernel?
// Every iteration each block writes its span, a grid barrier follows, then it reads the span of the next block and checks it.
// Cache policies of the raw buffer accesses (aux): 0 = plain, 16 = sc1 (agent-coherent), 17 = sc0|sc1 (system-coherent), 2 = nt.
// A bulk agent-scope fence per wave costs ~60 - 100 us per barrier (barrier_bench.hip, variant C); this probe looks for the cheap way.
//   hipcc --offload-arch=gfx950 -O3 -o coherence_probe tools/archive/coherence_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using Rsrc = __amdgpu_buffer_rsrc_t;
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)); }

// two-level barrier (barrier_bench.hip variant B); w: [xcd*16] XCD counters, [128] top, [144] flag, [160+xcd] blocks per XCD
__device__ __forceinline__ void barrier_b(unsigned* w, unsigned xcd, unsigned mine, unsigned nx, unsigned k) {
    __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0) expcnt(0) lgkmcnt(0): this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned a = __hip_atomic_fetch_add(&w[xcd * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a + 1 == mine * k) {
            const unsigned b = __hip_atomic_fetch_add(&w[128], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b + 1 == nx * k) __hip_atomic_store(&w[144], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        while (ld_relaxed(&w[144]) < k) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ void k_count(unsigned* w) { if (threadIdx.x == 0) atomicAdd(&w[160 + xcc_id()], 1u); }

template <int AS, int AL, int L1INV>
__global__ __launch_bounds__(1024) void k_probe(unsigned* w, int iters, float* buf, int vec_per_thread, unsigned long long* bad, int stride_blocks) {
    const unsigned xcd = xcc_id();
    const unsigned mine = w[160 + xcd];
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += (w[160 + i] > 0);
    const unsigned nb = gridDim.x;
    const long long span = (long long)blockDim.x * vec_per_thread * 4;            // floats per block
    const Rsrc rm = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (long long)blockIdx.x * span), 0, (int)(span * 4), 0x00020000);
    const unsigned ob = (blockIdx.x + stride_blocks) % nb;
    const Rsrc ro = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (long long)ob * span), 0, (int)(span * 4), 0x00020000);
    unsigned long long nbad = 0;
    for (int k = 1; k <= iters; ++k) {
        for (int j = 0; j < vec_per_thread; ++j) {
            const int e = (j * (int)blockDim.x + (int)threadIdx.x) * 16;
            const int v = k * 4096 + (int)blockIdx.x;
            v4i val = {v, v + 1, v + 2, j};
            __builtin_amdgcn_raw_buffer_store_b128(val, rm, e, 0, AS);
        }
        barrier_b(w, xcd, mine, nx, (unsigned)(2 * k - 1));
        if (L1INV == 1) asm volatile("buffer_inv sc1" ::: "memory");      // what an agent-scope acquire emits on gfx942 / gfx950
        if (L1INV == 2) asm volatile("buffer_inv sc0" ::: "memory");      // workgroup-scope flavour
        for (int j = 0; j < vec_per_thread; ++j) {
            const int e = (j * (int)blockDim.x + (int)threadIdx.x) * 16;
            const v4i got = __builtin_amdgcn_raw_buffer_load_b128(ro, e, 0, AL);
            const int v = k * 4096 + (int)ob;
            if (got[0] != v || got[1] != v + 1 || got[2] != v + 2 || got[3] != j) ++nbad;
        }
        barrier_b(w, xcd, mine, nx, (unsigned)(2 * k));
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int AS, int AL, int L1INV>
void run(const char* name, unsigned* w, float* buf, unsigned long long* bad, int nb, int threads, int stride) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int vpt : {1, 4, 16}) {
        const int iters = 300;
        float ms = 0;
        unsigned long long hb = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(w, 0, 4096)); CK(hipMemset(bad, 0, 8));
            hipLaunchKernelGGL(k_count, dim3(nb), dim3(threads), 0, 0, w);
            void* args[] = {&w, (void*)&iters, &buf, (void*)&vpt, &bad, (void*)&stride};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void*)k_probe<AS, AL, L1INV>, dim3(nb), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        }
        const double mb = (double)nb * threads * vpt * 16 / 1e6;
        printf("%-34s blocks %4d x %4d threads, neighbour +%d, %6.2f MB: %8.3f us/iteration (%7.1f GB/s written+read), stale vectors %llu of %.0f\n", name, nb, threads, stride, mb,
               1e3 * ms / iters, 2 * mb / (1e3 * ms / iters) * 1e3, hb, (double)nb * threads * vpt * iters);
    }
}

int main() {
    unsigned* w; float* buf; unsigned long long* bad;
    CK(hipMalloc(&w, 4096)); CK(hipMalloc(&buf, 1ll << 30)); CK(hipMalloc(&bad, 8));
    for (int stride : {1, 3}) {           // +1: the next block sits on the next XCD (round-robin dispatch); +8 would be the same XCD
        run<0, 0, 0>("plain store / plain load", w, buf, bad, 256, 1024, stride);
        run<0, 0, 1>("plain / plain + L1 invalidate", w, buf, bad, 256, 1024, stride);
        run<16, 16, 0>("sc1 store / sc1 load", w, buf, bad, 256, 1024, stride);
        run<16, 0, 0>("sc1 store / plain load", w, buf, bad, 256, 1024, stride);
        run<16, 0, 1>("sc1 store / plain load + inv sc1", w, buf, bad, 256, 1024, stride);
        run<16, 0, 2>("sc1 store / plain load + inv sc0", w, buf, bad, 256, 1024, stride);
        run<0, 16, 0>("plain store / sc1 load", w, buf, bad, 256, 1024, stride);
        run<17, 17, 0>("sc0|sc1 store / sc0|sc1 load", w, buf, bad, 256, 1024, stride);
        run<2, 2, 0>("nt store / nt load", w, buf, bad, 256, 1024, stride);
        run<18, 18, 0>("sc1|nt store / sc1|nt load", w, buf, bad, 256, 1024, stride);
    }
    run<16, 16, 0>("sc1 store / sc1 load, same XCD", w, buf, bad, 256, 1024, 8);
    run<0, 0, 0>("plain / plain, same XCD", w, buf, bad, 256, 1024, 8);
    run<0, 0, 1>("plain / plain + L1 inv, same XCD", w, buf, bad, 256, 1024, 8);
    run<16, 16, 0>("sc1 / sc1, 1024 x 256", w, buf, bad, 1024, 256, 1);
    return 0;
}
