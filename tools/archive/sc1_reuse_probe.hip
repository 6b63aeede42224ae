// sc1_reuse_probe.hip -- round 6: cost model of agent-coherent (sc1) loads on MI355X.  Each iteration a block writes its span (sc1 stores), a grid
// barrier follows, then it reads a span R times (R "stencil neighbours" touching the same lines) with plain or sc1 loads, own or foreign.
//   hipcc --offload-arch=gfx950 -O3 -o sc1_reuse_probe tools/archive/sc1_reuse_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using Rsrc = __amdgpu_buffer_rsrc_t;
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)); }
__device__ __forceinline__ void barrier_b(unsigned* w, unsigned xcd, unsigned mine, unsigned nx, unsigned k) {
    __builtin_amdgcn_s_waitcnt(0);
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

// AS: store policy; AL: load policy; FOREIGN: read the next block's span; R: passes over the span (shifted by one vector each: neighbour loads)
template <int AS, int AL, int FOREIGN, int R>
__global__ __launch_bounds__(1024) void k_probe(unsigned* w, int iters, float* buf, int vpt, unsigned long long* bad) {
    const unsigned xcd = xcc_id();
    const unsigned mine = w[160 + xcd];
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += (w[160 + i] > 0);
    const unsigned nb = gridDim.x;
    const long long span = (long long)blockDim.x * vpt * 4;
    const Rsrc rm = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (long long)blockIdx.x * span), 0, (int)(span * 4), 0x00020000);
    const unsigned ob = FOREIGN ? (blockIdx.x + 1) % nb : blockIdx.x;
    const Rsrc ro = __builtin_amdgcn_make_buffer_rsrc((void*)(buf + (long long)ob * span), 0, (int)(span * 4), 0x00020000);
    unsigned long long nbad = 0;
    const int nvec = (int)blockDim.x * vpt;
    for (int k = 1; k <= iters; ++k) {
        for (int j = 0; j < vpt; ++j) {
            const int e = j * (int)blockDim.x + (int)threadIdx.x;
            v4i val = {k, (int)blockIdx.x, e, 7};
            __builtin_amdgcn_raw_buffer_store_b128(val, rm, e * 16, 0, AS);
        }
        barrier_b(w, xcd, mine, nx, (unsigned)(2 * k - 1));
#pragma unroll
        for (int r = 0; r < R; ++r)
            for (int j = 0; j < vpt; ++j) {
                int e = j * (int)blockDim.x + (int)threadIdx.x + r;
                if (e >= nvec) e -= nvec;
                const v4i got = __builtin_amdgcn_raw_buffer_load_b128(ro, e * 16, 0, AL);
                if (got[0] != k || got[1] != (int)ob || got[2] != e) ++nbad;
            }
        barrier_b(w, xcd, mine, nx, (unsigned)(2 * k));
    }
    if (nbad) atomicAdd(bad, nbad);
}

template <int AS, int AL, int FOREIGN, int R>
void run(const char* name, unsigned* w, float* buf, unsigned long long* bad) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nb = 256, threads = 1024;
    for (int vpt : {1, 4}) {
        const int iters = 300;
        float ms = 0;
        unsigned long long hb = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(w, 0, 4096)); CK(hipMemset(bad, 0, 8));
            hipLaunchKernelGGL(k_count, dim3(nb), dim3(threads), 0, 0, w);
            void* args[] = {&w, (void*)&iters, &buf, (void*)&vpt, &bad};
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void*)k_probe<AS, AL, FOREIGN, R>, dim3(nb), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
        }
        const double mb = (double)nb * threads * vpt * 16 / 1e6;
        printf("%-46s %6.2f MB written, read x%d: %8.3f us/iteration, stale %llu\n", name, mb, R, 1e3 * ms / iters, hb);
    }
}

int main() {
    unsigned* w; float* buf; unsigned long long* bad;
    CK(hipMalloc(&w, 4096)); CK(hipMalloc(&buf, 1ll << 30)); CK(hipMalloc(&bad, 8));
    run<16, 0, 0, 1>("sc1 store, OWN span, plain loads", w, buf, bad);
    run<16, 0, 0, 7>("sc1 store, OWN span, plain loads", w, buf, bad);
    run<16, 16, 0, 1>("sc1 store, OWN span, sc1 loads", w, buf, bad);
    run<16, 16, 0, 7>("sc1 store, OWN span, sc1 loads", w, buf, bad);
    run<0, 0, 0, 1>("plain store, OWN span, plain loads", w, buf, bad);
    run<0, 0, 0, 7>("plain store, OWN span, plain loads", w, buf, bad);
    run<16, 16, 1, 1>("sc1 store, FOREIGN span (next XCD), sc1 loads", w, buf, bad);
    run<16, 16, 1, 7>("sc1 store, FOREIGN span (next XCD), sc1 loads", w, buf, bad);
    run<16, 1, 1, 1>("sc1 store, FOREIGN span, sc0 loads", w, buf, bad);
    run<16, 1, 1, 7>("sc1 store, FOREIGN span, sc0 loads", w, buf, bad);
    return 0;
}
