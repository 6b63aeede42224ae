// vmtest -- does the PHYSICAL make-up of an allocation (fragment size seen by the GPU's TLBs, alignment of virtual against
// physical addresses) decide how fast a streaming kernel runs over it?  (round 4, verdict item 1: the "placement lottery")
//
// The same many-stream copy (8 read + 8 write streams, 512-thread blocks, one per CU -- the one-sweep CP kernel's shape, tools/bwtest3
// "mix") runs over buffers obtained in different ways:
//   A  hipMalloc                                   (what PyTorch's caching allocator hands the solvers)
//   B  virtual-memory API: ONE physical handle for the whole buffer, virtual address aligned to 2 GiB
//   C  the same physical handle mapped at a virtual address that is only 2 MiB aligned (2 GiB boundary + 2 MiB)
//   D  one physical handle per `chunk` (2 MiB ... 1 GiB), mapped back to back at a 2 GiB aligned virtual address
//   E  like D with the chunks of the SOURCE and the DESTINATION created alternately (physically interleaved allocations)
// usage: tools/vmtest [GiB per buffer = 8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s (%d) at %s:%d: %s\n", hipGetErrorString(e_), (int)e_, __FILE__, __LINE__, #x); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NS, int T>
__global__ __launch_bounds__(T) void k_streams(const f4* __restrict__ a, f4* __restrict__ b, long long per, long long sper) {
    const long long c0 = (long long)blockIdx.x * per;
    const int tid = threadIdx.x;
    for (long long k = 0; k + T <= per; k += T) {
        f4 v[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) v[s] = __builtin_nontemporal_load(a + s * sper + c0 + k + tid);
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_nontemporal_store(v[s] * 1.0001f, b + s * sper + c0 + k + tid);
    }
}

static hipEvent_t e0, e1;
static double bench(const char* name, const f4* a, f4* b, long long n) {
    const int NS = 8, T = 512, blocks = 256;
    const long long sper = n / NS;
    long long per = sper / blocks;
    per -= per % T;
    std::vector<float> ms;
    for (int r = 0; r < 6; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_streams<NS, T>), dim3(blocks), dim3(T), 0, 0, a, b, per, sper);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (r > 0) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    const double bytes = (double)per * blocks * NS * 32;
    printf("%-88s a=%p b=%p  median %7.3f ms %6.0f GB/s (min %.3f max %.3f)\n", name, (const void*)a, (void*)b, ms[ms.size() / 2], bytes / ms[ms.size() / 2] * 1e-6,
           ms.front(), ms.back());
    fflush(stdout);
    return ms[ms.size() / 2];
}

struct VM {
    void* va = nullptr;
    size_t reserved = 0;
    void* mapped_at = nullptr;
    size_t size = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    size_t chunk = 0;
};

static hipMemAllocationProp prop_dev(int dev) {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = dev;
    return p;
}
static void set_access(void* p, size_t size, int dev) {
    hipMemAccessDesc d = {};
    d.location.type = hipMemLocationTypeDevice;
    d.location.id = dev;
    d.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(p, size, &d, 1));
}
// reserve `size + slack` bytes of virtual address space aligned to `align`; map the handles at va + offset
static void vm_reserve(VM& v, size_t size, size_t align, size_t offset) {
    v.size = size;
    v.reserved = size + offset + (2ull << 20);
    CK(hipMemAddressReserve(&v.va, v.reserved, align, nullptr, 0));
    v.mapped_at = (char*)v.va + offset;
}
static void vm_release(VM& v) {
    if (v.mapped_at && !v.handles.empty()) CK(hipMemUnmap(v.mapped_at, v.size));
    for (auto h : v.handles) CK(hipMemRelease(h));
    v.handles.clear();
    if (v.va) CK(hipMemAddressFree(v.va, v.reserved));
    v.va = nullptr;
}

int main(int argc, char** argv) {
    const long long gib = (argc > 1) ? atoll(argv[1]) : 8;
    const size_t size = (size_t)gib << 30;
    const long long n = (long long)(size / 16);
    int dev = 0;
    CK(hipSetDevice(dev));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipMemAllocationProp prop = prop_dev(dev);
    size_t gran_min = 0, gran_rec = 0;
    CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
    printf("# vmtest: %lld GiB per buffer; allocation granularity min %zu, recommended %zu bytes\n", gib, gran_min, gran_rec);

    for (int rep = 0; rep < 2; ++rep) {
        {   // A: hipMalloc
            f4 *a, *b;
            CK(hipMalloc(&a, size));
            CK(hipMalloc(&b, size));
            CK(hipMemset(a, 1, size));
            CK(hipMemset(b, 0, size));
            bench("A hipMalloc + hipMalloc", a, b, n);
            CK(hipFree(a));
            CK(hipFree(b));
        }
        for (int variant = 0; variant < 2; ++variant) {   // B / C: one handle per buffer, VA aligned to 2 GiB / off by 2 MiB
            VM va, vb;
            const size_t off = variant ? (2ull << 20) : 0;
            vm_reserve(va, size, 2ull << 30, off);
            vm_reserve(vb, size, 2ull << 30, off);
            hipMemGenericAllocationHandle_t ha, hb;
            CK(hipMemCreate(&ha, size, &prop, 0));
            CK(hipMemCreate(&hb, size, &prop, 0));
            va.handles.push_back(ha);
            vb.handles.push_back(hb);
            CK(hipMemMap(va.mapped_at, size, 0, ha, 0));
            CK(hipMemMap(vb.mapped_at, size, 0, hb, 0));
            set_access(va.mapped_at, size, dev);
            set_access(vb.mapped_at, size, dev);
            CK(hipMemset(va.mapped_at, 1, size));
            CK(hipMemset(vb.mapped_at, 0, size));
            bench(variant ? "C one physical handle per buffer, VA = 2 GiB boundary + 2 MiB" : "B one physical handle per buffer, VA aligned to 2 GiB", (const f4*)va.mapped_at,
                  (f4*)vb.mapped_at, n);
            vm_release(va);
            vm_release(vb);
        }
        for (size_t chunk : {(size_t)2 << 20, (size_t)32 << 20, (size_t)1 << 30}) {   // D / E: many handles
            for (int inter = 0; inter < 2; ++inter) {
                if (chunk < gran_min) continue;
                VM va, vb;
                vm_reserve(va, size, 2ull << 30, 0);
                vm_reserve(vb, size, 2ull << 30, 0);
                const size_t nch = size / chunk;
                if (inter) {
                    for (size_t i = 0; i < nch; ++i) {
                        hipMemGenericAllocationHandle_t h1, h2;
                        CK(hipMemCreate(&h1, chunk, &prop, 0));
                        CK(hipMemCreate(&h2, chunk, &prop, 0));
                        va.handles.push_back(h1);
                        vb.handles.push_back(h2);
                    }
                } else {
                    for (size_t i = 0; i < nch; ++i) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); va.handles.push_back(h); }
                    for (size_t i = 0; i < nch; ++i) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, chunk, &prop, 0)); vb.handles.push_back(h); }
                }
                for (size_t i = 0; i < nch; ++i) {
                    CK(hipMemMap((char*)va.mapped_at + i * chunk, chunk, 0, va.handles[i], 0));
                    CK(hipMemMap((char*)vb.mapped_at + i * chunk, chunk, 0, vb.handles[i], 0));
                }
                set_access(va.mapped_at, size, dev);
                set_access(vb.mapped_at, size, dev);
                CK(hipMemset(va.mapped_at, 1, size));
                CK(hipMemset(vb.mapped_at, 0, size));
                char name[160];
                snprintf(name, sizeof name, "%s one physical handle per %zu MiB%s", inter ? "E" : "D", chunk >> 20, inter ? ", source / destination chunks created alternately" : "");
                bench(name, (const f4*)va.mapped_at, (f4*)vb.mapped_at, n);
                // unmap chunk by chunk
                for (size_t i = 0; i < nch; ++i) {
                    CK(hipMemUnmap((char*)va.mapped_at + i * chunk, chunk));
                    CK(hipMemUnmap((char*)vb.mapped_at + i * chunk, chunk));
                }
                for (auto h : va.handles) CK(hipMemRelease(h));
                for (auto h : vb.handles) CK(hipMemRelease(h));
                va.handles.clear();
                vb.handles.clear();
                CK(hipMemAddressFree(va.va, va.reserved));
                CK(hipMemAddressFree(vb.va, vb.reserved));
            }
        }
    }
    return 0;
}
