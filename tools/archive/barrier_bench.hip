// barrier_bench.hip -- what does a grid-wide barrier cost on MI355X (256 CUs, 8 XCDs)?  round 6, persistent small-volume solver.
//   hipcc --offload-arch=gfx950 -O3 -o barrier_bench tools/archive/barrier_bench.hip && ./barrier_bench
// Variants: (A) one monotonic counter, every block arrives with one agent-scope atomic and polls it;
//           (B) two levels: one counter per XCD (blocks of an XCD arrive there), the last arriver of an XCD arrives at a top counter,
//               everybody polls a single release flag;
//           (C) like A but the fences are the full agent-scope release / acquire with DIRTY data (each block writes `bytes` before the barrier
//               and reads its neighbour's bytes after it): the cost of the L2 write-back / invalidate that coherence across XCDs needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool FENCE>
__device__ __forceinline__ void barrier_a(unsigned* counter, unsigned target) {
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_relaxed(counter) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ void k_a(unsigned* counter, int iters, long long* cycles) {
    const unsigned nb = gridDim.x;
    long long t0 = wall_clock64();
    for (int k = 1; k <= iters; ++k) barrier_a<false>(counter, nb * (unsigned)k);
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = wall_clock64() - t0;
}

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 /* HW_REG_XCC_ID */) | (0 << 6) | ((4 - 1) << 11)); }

// words: [0..7*16] per-XCD counters (64-byte apart), [128] top counter, [144] flag, [160..] per-XCD block counts (filled by a first pass)
__global__ void k_count(unsigned* w) {
    if (threadIdx.x == 0) atomicAdd(&w[160 + xcc_id()], 1u);
}
__global__ void k_b(unsigned* w, int iters, long long* cycles) {
    const unsigned xcd = xcc_id();
    const unsigned mine = w[160 + xcd];
    unsigned nx = 0;
    for (int i = 0; i < 8; ++i) nx += (w[160 + i] > 0);
    long long t0 = wall_clock64();
    for (int k = 1; k <= iters; ++k) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned a = __hip_atomic_fetch_add(&w[xcd * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a + 1 == mine * (unsigned)k) {
                const unsigned b = __hip_atomic_fetch_add(&w[128], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (b + 1 == nx * (unsigned)k) __hip_atomic_store(&w[144], (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            while (ld_relaxed(&w[144]) < (unsigned)k) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = wall_clock64() - t0;
}

// E: flag array -- every block stores its own epoch (no read-modify-write, no contention), wave 0 polls ALL flags with one 16-byte
// agent-coherent load per lane (256 blocks = 1 KiB) and reduces with a ballot.  flags: one unsigned per block, padded to a multiple of 256.
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__global__ void k_e(unsigned* flags, int iters, long long* cycles) {
    const unsigned nb = gridDim.x;
    const unsigned nvec = (nb + 3) / 4;         // 16-byte vectors to poll
    long long t0 = wall_clock64();
    for (int k = 1; k <= iters; ++k) {
        __syncthreads();
        if (threadIdx.x < 64) {
            if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], (unsigned)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok;
            do {
                ok = true;
                for (unsigned v = threadIdx.x; v < nvec; v += 64) {
                    const unsigned* p = flags + 4 * v;
                    const unsigned a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned c = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), d = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = ok && a >= (unsigned)k && b >= (unsigned)k && c >= (unsigned)k && d >= (unsigned)k;
                }
                ok = __all(ok);
            } while (!ok);
        }
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = wall_clock64() - t0;
}

// C: barrier with real release / acquire and dirty data: every thread writes `per_thread` floats (coalesced) and after the barrier reads what
// the NEXT block wrote
__global__ void k_c(unsigned* counter, int iters, long long* cycles, float* buf, int per_thread, float* sink) {
    const unsigned nb = gridDim.x;
    const long long span = (long long)blockDim.x * per_thread;
    float acc = 0.f;
    long long t0 = wall_clock64();
    for (int k = 1; k <= iters; ++k) {
        float* mine = buf + (long long)blockIdx.x * span;
        for (int j = 0; j < per_thread; ++j) mine[(long long)j * blockDim.x + threadIdx.x] = acc + (float)k;
        barrier_a<true>(counter, nb * (unsigned)(2 * k - 1));
        const float* other = buf + (long long)((blockIdx.x + 1) % nb) * span;
        for (int j = 0; j < per_thread; ++j) acc += other[(long long)j * blockDim.x + threadIdx.x];
        barrier_a<true>(counter, nb * (unsigned)(2 * k));
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
    if (blockIdx.x == 0 && threadIdx.x == 0) cycles[0] = wall_clock64() - t0;
}

int main() {
    unsigned* w; long long* cyc; float* buf; float* sink; unsigned* fl;
    CK(hipMalloc(&fl, 8192));
    CK(hipMalloc(&w, 4096)); CK(hipMalloc(&cyc, 64)); CK(hipMalloc(&buf, 512ll << 20)); CK(hipMalloc(&sink, 1 << 16));
    int wc_khz = 0; CK(hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    printf("# %s, %d CUs, wall clock %d kHz\n", pr.name, pr.multiProcessorCount, wc_khz);
    const int iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int threads : {256, 512, 1024}) {
        for (int nb : {64, 128, 256, 512, 1024}) {
            if ((long long)nb * threads > 256ll * 2048) continue;
            int per_cu = 0;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_a, threads, 0));
            if (nb > per_cu * pr.multiProcessorCount) continue;
            float ms_a = 0, ms_b = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(w, 0, 4096));
                void* args[] = {&w, (void*)&iters, &cyc};
                CK(hipEventRecord(e0));
                CK(hipLaunchCooperativeKernel((void*)k_a, dim3(nb), dim3(threads), args, 0, 0));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_a, e0, e1));
            }
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(w, 0, 4096));
                hipLaunchKernelGGL(k_count, dim3(nb), dim3(threads), 0, 0, w);      // which XCD each block lands on (same grid -> same round-robin)
                void* args[] = {&w, (void*)&iters, &cyc};
                CK(hipEventRecord(e0));
                CK(hipLaunchCooperativeKernel((void*)k_b, dim3(nb), dim3(threads), args, 0, 0));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_b, e0, e1));
            }
            unsigned hw[176]; CK(hipMemcpy(hw, w, sizeof(hw), hipMemcpyDeviceToHost));
            float ms_e = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(fl, 0xff, 8192)); CK(hipMemset(fl, 0, 4 * nb));       // pad entries beyond the grid never block
                void* args[] = {&fl, (void*)&iters, &cyc};
                CK(hipEventRecord(e0));
                CK(hipLaunchCooperativeKernel((void*)k_e, dim3(nb), dim3(threads), args, 0, 0));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_e, e0, e1));
            }
            printf("threads %4d blocks %4d | E flag array %.3f us/barrier", threads, nb, 1e3 * ms_e / iters);
            printf(" | A one counter %.3f us/barrier | B per-XCD + top %.3f us/barrier | blocks per XCD %u %u %u %u %u %u %u %u\n",
                   1e3 * ms_a / iters, 1e3 * ms_b / iters, hw[160], hw[161], hw[162], hw[163], hw[164], hw[165], hw[166], hw[167]);
        }
    }
    if (getenv("BB_FENCED")) for (int threads : {256, 1024}) {
        const int nb = (threads == 256) ? 1024 : 256;
        for (int per_thread : {0, 1, 4, 16, 64}) {          // bytes written per iteration = nb * threads * per_thread * 4
            float ms = 0;
            const int it2 = 500;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemset(w, 0, 4096));
                void* args[] = {&w, (void*)&it2, &cyc, &buf, (void*)&per_thread, &sink};
                CK(hipEventRecord(e0));
                CK(hipLaunchCooperativeKernel((void*)k_c, dim3(nb), dim3(threads), args, 0, 0));
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const double mb = (double)nb * threads * per_thread * 4 / 1e6;
            printf("C threads %4d blocks %4d, %7.2f MB written + read per iteration, 2 fenced barriers: %.3f us/iteration (%.1f GB/s write+read)\n", threads, nb, mb,
                   1e3 * ms / it2, mb > 0 ? 2 * mb / (1e3 * ms / it2) * 1e3 : 0.0);
        }
    }
    return 0;
}
