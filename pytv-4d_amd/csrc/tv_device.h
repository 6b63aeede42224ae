// tv_device.h -- device-side building blocks of the MI355X (gfx950) TV stencil engine.
//
// Everything here is written for CDNA4 directly: 64-lane wavefronts, 16-byte per-lane global
// accesses along the fastest (column) axis, fp64 wave-level reductions with DPP shuffles.
// The stencils are HBM-bandwidth bound (O(1) flop/byte): no MFMA anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tv {

enum : int { UPWIND = 0, DOWNWIND = 1, CENTRAL = 2, HYBRID = 3 };

// ------------------------------------------------------------------------------------------
// geometry as the kernels see it (built on the host from tv_geom)
// ------------------------------------------------------------------------------------------
struct DG {
    int nz, m, ny, nx;     // local slab
    int nzg, z0;           // global plane count, global index of local plane 0
    int nd;                // gradient channels
    int za, ta;            // z / time axis active
    int z_two, t_two;      // central scheme: axis has exactly two points -> forward stencil
    int vl;                // columns per 16-byte lane of the image dtype: 4 (fp32) / 2 (fp64)
    int ch_z, ch_t;        // first channel of the z / time axis
    int rp;                // row pitch in elements (tv_geom::row_pitch; == nx for dense arrays): rows of every image-like array
    int pitched;           // 1: row_pitch / frame_pitch given (pads exist and hold zeros); 0: the reference's dense layout
    long long s_t;         // frame stride            tv_geom::frame_pitch, dense: ny*nx
    long long s_z;         // image plane stride      m*s_t   (== gradient channel stride)
    long long s_dz;        // gradient plane stride   nd*m*s_t
    const uint8_t* mask;   // ny*nx or nullptr
    const void* tf;        // ny*nx elements of the image dtype (per-pixel time-channel factor) or nullptr
    const void* wv;        // nz*m*ny*nx elements of the image dtype (per-VOXEL time-channel factor, local planes) or nullptr
    const void* wvp;       // plane z0-1 / z0+nz of the same (ghost planes of the two-pass sub-gradient) or nullptr
    const void* wvn;
};

template <typename T> struct WT { T wz, wt, sf; };   // sqrt(reg_z), sqrt(reg_time), sqrt(factor_static)

template <typename T> struct Consts;
template <> struct Consts<float>  { static __device__ __forceinline__ float  inv_sqrt2() { return 0.70710678118654752440f; } };
template <> struct Consts<double> { static __device__ __forceinline__ double inv_sqrt2() { return 0.70710678118654752440; } };

// IEEE-correct sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt keeps it so) and max
__device__ __forceinline__ float  tsqrt(float v)  { return __fsqrt_rn(v); }
__device__ __forceinline__ double tsqrt(double v) { return __dsqrt_rn(v); }
__device__ __forceinline__ float  tmax(float a, float b)   { return fmaxf(a, b); }
__device__ __forceinline__ double tmax(double a, double b) { return fmax(a, b); }

// ------------------------------------------------------------------------------------------
// V-wide vectors along the column axis (V*sizeof(T) == 16 on the fast path, V == 1 fallback)
// ------------------------------------------------------------------------------------------
template <typename T, int V> struct alignas(sizeof(T) * V) Vec { T v[V]; };

template <typename T, int V> __device__ __forceinline__ Vec<T, V> vload(const T* p) {
    return *reinterpret_cast<const Vec<T, V>*>(p);
}
template <typename T, int V> __device__ __forceinline__ void vstore(T* p, const Vec<T, V>& a) {
    *reinterpret_cast<Vec<T, V>*>(p) = a;
}
// Streamed-once arrays of the epilogues: non-temporal loads AND stores (mixing nt loads with plain stores
// is slower than either, tools/archive/bwtest3.hip).  Which epilogue uses them was settled by A/B on the device
// (profiles/r3_epi_nt_ab.txt): CpPrimal and AxpyDT gain 4-8 %, CpDual gains for central and loses for the
// 8-channel schemes, StoreDT / AdmmZU lose up to 30 % for central, StoreD is neutral.
#ifndef TV_EPI_NT
#define TV_EPI_NT 1
#endif
template <typename T, int V, bool NT = true> __device__ __forceinline__ Vec<T, V> vload_s(const T* p) {
#if TV_EPI_NT
    if constexpr (!NT) return vload<T, V>(p);
    else if constexpr (V == 1) {
        Vec<T, V> r; r.v[0] = __builtin_nontemporal_load(p); return r;
    } else {
        typedef T nt_v __attribute__((ext_vector_type(V)));
        const nt_v v = __builtin_nontemporal_load(reinterpret_cast<const nt_v*>(p));
        Vec<T, V> r;
#pragma unroll
        for (int i = 0; i < V; ++i) r.v[i] = v[i];
        return r;
    }
#else
    return vload<T, V>(p);
#endif
}
template <typename T, int V, bool NT = true> __device__ __forceinline__ void vstore_s(T* p, const Vec<T, V>& a) {
#if TV_EPI_NT
    if constexpr (!NT) vstore<T, V>(p, a);
    else if constexpr (V == 1) {
        __builtin_nontemporal_store(a.v[0], p);
    } else {
        typedef T nt_v __attribute__((ext_vector_type(V)));
        nt_v w;
#pragma unroll
        for (int i = 0; i < V; ++i) w[i] = a.v[i];
        __builtin_nontemporal_store(w, reinterpret_cast<nt_v*>(p));
    }
#else
    vstore<T, V>(p, a);
#endif
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> vsplat(T s) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = s;
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> operator-(const Vec<T, V>& a, const Vec<T, V>& b) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = a.v[i] - b.v[i];
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> operator+(const Vec<T, V>& a, const Vec<T, V>& b) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = a.v[i] + b.v[i];
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> operator*(const Vec<T, V>& a, const Vec<T, V>& b) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = a.v[i] * b.v[i];
    return r;
}
template <typename T, int V> __device__ __forceinline__ Vec<T, V> operator*(T s, const Vec<T, V>& a) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i < V; ++i) r.v[i] = s * a.v[i];
    return r;
}
// {a[1], .., a[V-1], tail}: the vector one column to the right
template <typename T, int V> __device__ __forceinline__ Vec<T, V> shift_left(const Vec<T, V>& a, T tail) {
    Vec<T, V> r;
#pragma unroll
    for (int i = 0; i + 1 < V; ++i) r.v[i] = a.v[i + 1];
    r.v[V - 1] = tail;
    return r;
}
// {head, a[0], .., a[V-2]}: the vector one column to the left
template <typename T, int V> __device__ __forceinline__ Vec<T, V> shift_right(const Vec<T, V>& a, T head) {
    Vec<T, V> r;
    r.v[0] = head;
#pragma unroll
    for (int i = 1; i < V; ++i) r.v[i] = a.v[i - 1];
    return r;
}

// the last lane of a ragged PITCHED row (tv_geom::row_pitch; nx not a multiple of V) holds pad columns: results there are
// forced to zero before they are stored, so that "pads hold zeros" survives every kernel (V == 1 / whole lanes: nothing to do)
template <typename T, int V> __device__ __forceinline__ void zero_pad_cols(const DG& g, int col0, Vec<T, V>& r) {
    if (V > 1 && col0 + V > g.nx) {
#pragma unroll
        for (int i = 0; i < V; ++i)
            if (col0 + i >= g.nx) r.v[i] = T(0);
    }
}

// ------------------------------------------------------------------------------------------
// thread -> voxel-vector mapping.  grid = (tiles_x * tiles_y, m, planes); block = (BX, BY)
// ------------------------------------------------------------------------------------------
struct Coord { int zl, t, y, col0; bool ok; };

template <int V> __device__ __forceinline__ Coord thread_coord(const DG& g, int z_first) {
    const int nxv = (g.nx + V - 1) / V;      // (a last lane with pad columns exists only on pitched rows: tv_geom::row_pitch)
    const int tiles_x = (nxv + (int)blockDim.x - 1) / (int)blockDim.x;
    const int bx = (int)blockIdx.x % tiles_x, by = (int)blockIdx.x / tiles_x;
    Coord c;
    const int jv = bx * (int)blockDim.x + (int)threadIdx.x;
    c.col0 = jv * V;
    c.y = by * (int)blockDim.y + (int)threadIdx.y;
    c.t = (int)blockIdx.y;
    c.zl = z_first + (int)blockIdx.z;
    c.ok = (jv < nxv) && (c.y < g.ny);
    return c;
}

// Pointer to the image plane with LOCAL index zl (zl may lie in the halo), nullptr if that plane
// does not exist globally or was not supplied.  hp = planes per halo buffer.
template <typename T>
__device__ __forceinline__ const T* zplane(const DG& g, const T* x, const T* xp, const T* xn, int hp, int zl) {
    const int gz = g.z0 + zl;
    if (gz < 0 || gz >= g.nzg) return nullptr;
    if (zl >= 0 && zl < g.nz) return x + (long long)zl * g.s_z;
    if (zl < 0) return (xp != nullptr && zl >= -hp) ? xp + (long long)(hp + zl) * g.s_z : nullptr;
    const int k = zl - g.nz;
    return (xn != nullptr && k < hp) ? xn + (long long)k * g.s_z : nullptr;
}

// per-pixel factor of the time channels: sqrt(factor_reg_static) where mask_static is set
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> mask_factor(const DG& g, T sf, int y, int col0) {
    Vec<T, V> r = vsplat<T, V>(T(1));
    // mask / factor maps are DENSE (ny * nx); the last lane of a ragged pitched row must not read past the row (V == 1: no test)
    if (g.mask != nullptr) {
        const uint8_t* mp = g.mask + (long long)y * g.nx + col0;
#pragma unroll
        for (int i = 0; i < V; ++i) r.v[i] = (V == 1 || col0 + i < g.nx) ? (mp[i] ? sf : T(1)) : T(1);
    }
    if (g.tf != nullptr) {
        const T* fp = static_cast<const T*>(g.tf) + (long long)y * g.nx + col0;
#pragma unroll
        for (int i = 0; i < V; ++i) r.v[i] *= (V == 1 || col0 + i < g.nx) ? fp[i] : T(1);
    }
    return r;
}
// per-VOXEL factor of the time channels (tv_geom::time_weight_vol) at local plane zl (ghost planes: -1, nz), frame t;
// 1 where there is no weight volume, the frame does not exist or the ghost plane was not supplied
template <typename T> __device__ __forceinline__ const T* vol_plane(const DG& g, int zl, int t) {
    if (g.wv == nullptr || t < 0 || t >= g.m) return nullptr;
    const T* pl = (zl >= 0 && zl < g.nz) ? static_cast<const T*>(g.wv) + (long long)zl * g.s_z
                                          : static_cast<const T*>(zl < 0 ? g.wvp : g.wvn);
    return pl != nullptr ? pl + (long long)t * g.s_t : nullptr;
}
template <typename T, int V>
__device__ __forceinline__ Vec<T, V> vol_factor(const DG& g, int zl, int t, int y, int col0) {
    const T* pl = vol_plane<T>(g, zl, t);
    if (pl == nullptr) return vsplat<T, V>(T(1));
    return vload<T, V>(pl + (long long)y * g.rp + col0);
}
template <typename T> __device__ __forceinline__ T vol_factor1(const DG& g, int zl, int t, int y, int col) {
    const T* pl = vol_plane<T>(g, zl, t);
    return pl != nullptr ? pl[(long long)y * g.rp + col] : T(1);
}
// the same for one pixel
template <typename T> __device__ __forceinline__ T mask_factor1(const DG& g, T sf, int y, int col) {
    T r = T(1);
    if (g.mask != nullptr && g.mask[(long long)y * g.nx + col]) r = sf;
    if (g.tf != nullptr) r *= static_cast<const T*>(g.tf)[(long long)y * g.nx + col];
    return r;
}

// ------------------------------------------------------------------------------------------
// fp64 reductions: DPP/shuffle inside the 64-lane wave, LDS across the block's waves
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// every thread of the block must call this; the result is valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* smem /* >= 16 doubles */) {
    const int tid = (int)(threadIdx.y * blockDim.x + threadIdx.x);
    const int lane = tid & 63, wid = tid >> 6;
    const int nw = ((int)(blockDim.x * blockDim.y) + 63) >> 6;
    v = wave_sum(v);
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    if (wid == 0) {
        v = (lane < nw) ? smem[lane] : 0.0;
        v = wave_sum(v);
    }
    __syncthreads();
    return v;
}

__device__ __forceinline__ long long linear_block_id() {
    return (long long)blockIdx.x + (long long)gridDim.x * ((long long)blockIdx.y + (long long)gridDim.y * (long long)blockIdx.z);
}

}  // namespace tv
