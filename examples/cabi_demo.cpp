// cabi_demo.cpp -- the drop-in boundary used WITHOUT Python or PyTorch: plain hipMalloc'd buffers, include/pytv4d.h,
// libpytv4d_hip.so.  Checks <D x, y> == <x, D^T y> (the reference's own adjointness test, pytv/tests.py:363-404) on the
// GPU results and runs a few Chambolle-Pock iterations (README.md:145-157) through tv_cp_dual / tv_cp_primal and through
// the one-sweep pair tv_cp_fused / tv_cp_fixup, which must produce the same loss -- and through the persistent loop tv_small_cp.
//
//   hipcc -O2 -Iinclude examples/cabi_demo.cpp -Lpytv-4d_amd/pytv -lpytv4d_hip -Wl,-rpath,$PWD/pytv-4d_amd/pytv -o /tmp/cabi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pytv4d.h"

#define HIP_OK(call)                                                                     \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e__)); return 2; } \
    } while (0)
#define TV_OK(call)                                                                      \
    do {                                                                                 \
        int rc__ = (call);                                                               \
        if (rc__ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, tv_last_error()); return 3; } \
    } while (0)

static float* to_device(const std::vector<float>& h) {
    float* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(float)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main() {
    tv_geom g;
    tv_geom_init(&g);            // zeroes the struct and stamps it with sizeof(tv_geom) / TV_ABI_VERSION
    g.nz = 6; g.m = 4; g.ny = 40; g.nx = 128;
    g.nz_global = g.nz; g.z0 = 0;
    g.scheme = TV_HYBRID; g.dtype = TV_F32;
    g.reg_z_over_reg = 1.5; g.reg_time = 0.5; g.factor_reg_static = 0.0; g.mask_static = nullptr;
    const int nd = tv_num_channels(&g);
    if (nd != 8) { fprintf(stderr, "expected 8 channels, got %d\n", nd); return 1; }
    const size_t V = (size_t)g.nz * g.m * g.ny * g.nx, VD = V * nd;
    srand(7);
    std::vector<float> hx(V), hy(VD), hx0(V);
    for (auto& v : hx) v = rand() / (float)RAND_MAX;
    for (auto& v : hy) v = rand() / (float)RAND_MAX - 0.5f;
    for (auto& v : hx0) v = 100.f * rand() / (float)RAND_MAX;
    float *x = to_device(hx), *y = to_device(hy), *d = nullptr, *dt = nullptr;
    HIP_OK(hipMalloc(&d, VD * sizeof(float)));
    HIP_OK(hipMalloc(&dt, V * sizeof(float)));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    // ---- adjointness -----------------------------------------------------------------------------------------
    TV_OK(tv_D(&g, x, nullptr, nullptr, d, st));
    TV_OK(tv_DT(&g, y, nullptr, nullptr, dt, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> hd(VD), hdt(V);
    HIP_OK(hipMemcpy(hd.data(), d, VD * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hdt.data(), dt, V * sizeof(float), hipMemcpyDeviceToHost));
    double lhs = 0.0, rhs = 0.0;
    for (size_t i = 0; i < VD; ++i) lhs += (double)hd[i] * hy[i];
    for (size_t i = 0; i < V; ++i) rhs += (double)hx[i] * hdt[i];
    printf("<Dx, y> = %.9e   <x, D^T y> = %.9e   rel diff %.2e\n", lhs, rhs, fabs(lhs - rhs) / fabs(lhs));
    if (fabs(lhs - rhs) > 1e-5 * fabs(lhs)) { fprintf(stderr, "adjointness violated\n"); return 1; }

    // ---- Chambolle-Pock, two ways ---------------------------------------------------------------------------------
    const double lambda = 25.0, sigma_D = 0.5, sigma_A = 1.0, tau = 1.0 / (1.0 + 4.0 * (2.0 + g.reg_z_over_reg + g.reg_time));
    const size_t wsb = tv_workspace_bytes(&g);
    void* ws = nullptr;
    double* sc = nullptr;       // device scalars: tv, fid (sweep), fid (fix-up)
    HIP_OK(hipMalloc(&ws, wsb));
    HIP_OK(hipMalloc(&sc, 3 * sizeof(double)));
    double loss[2][5];
    for (int variant = 0; variant < 2; ++variant) {
        if (variant == 1 && !tv_cp_fused_supported(&g)) { printf("one-sweep path not supported here\n"); break; }
        float *xa = to_device(hx0), *xb = to_device(hx0), *x0 = to_device(hx0), *p = nullptr, *q = nullptr;
        HIP_OK(hipMalloc(&p, V * sizeof(float)));
        HIP_OK(hipMalloc(&q, VD * sizeof(float)));
        HIP_OK(hipMemset(p, 0, V * sizeof(float)));
        HIP_OK(hipMemset(q, 0, VD * sizeof(float)));
        for (int it = 0; it < 5; ++it) {
            double h[3] = {0, 0, 0};
            if (variant == 0) {
                TV_OK(tv_cp_dual(&g, xa, nullptr, nullptr, q, sigma_D, lambda, sc, ws, st));
                TV_OK(tv_cp_primal(&g, q, nullptr, nullptr, xa, x0, p, tau, sigma_A, sc + 1, ws, st));
                HIP_OK(hipMemcpyAsync(h, sc, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
            } else {
                TV_OK(tv_cp_fused(&g, xa, nullptr, nullptr, q, x0, p, xb, sigma_D, lambda, tau, sigma_A, 0, -1, sc, sc + 1, ws, st));
                TV_OK(tv_cp_fixup(&g, q, nullptr, nullptr, xb, x0, tau, 0, -1, sc + 2, ws, st));
                HIP_OK(hipMemcpyAsync(h, sc, 3 * sizeof(double), hipMemcpyDeviceToHost, st));
                float* t = xa; xa = xb; xb = t;
            }
            HIP_OK(hipStreamSynchronize(st));
            loss[variant][it] = h[1] + h[2] + lambda * h[0];
        }
        printf("%-22s loss: %.6e -> %.6e\n", variant == 0 ? "tv_cp_dual + primal" : "tv_cp_fused + fixup", loss[variant][0], loss[variant][4]);
        for (float* b : {xa, xb, x0, p, q}) (void)hipFree(b);
        if (!(loss[variant][4] < loss[variant][0])) { fprintf(stderr, "loss did not decrease\n"); return 1; }
    }
    for (int it = 0; it < 5; ++it)
        if (fabs(loss[0][it] - loss[1][it]) > 1e-5 * fabs(loss[0][it])) { fprintf(stderr, "the two CP paths disagree at iteration %d\n", it); return 1; }
    // ---- the same five iterations as ONE persistent launch (round 6: tv_small_cp, volumes that fit the caches) ---------------------------
    if (tv_small_supported(&g)) {
        float *xs = to_device(hx0), *x0 = to_device(hx0), *p = nullptr, *q = nullptr;
        void* wss = nullptr;
        double* hist = nullptr;              // device: (5, 2) = TV of the iterate each iteration saw, 1/2 |x_new - x0|^2
        const size_t wsb_small = tv_small_workspace_bytes(&g, 5);
        HIP_OK(hipMalloc(&p, V * sizeof(float)));
        HIP_OK(hipMalloc(&q, VD * sizeof(float)));
        HIP_OK(hipMalloc(&wss, wsb_small));
        HIP_OK(hipMalloc(&hist, 10 * sizeof(double)));
        HIP_OK(hipMemset(p, 0, V * sizeof(float)));
        HIP_OK(hipMemset(q, 0, VD * sizeof(float)));
        HIP_OK(hipMemset(wss, 0, wsb_small));                 // ONCE: the block flags continue from call to call
        TV_OK(tv_small_cp(&g, xs, x0, p, q, sigma_D, lambda, tau, sigma_A, 3, hist, 2, 1, wss, st));          // three iterations ...
        TV_OK(tv_small_cp(&g, xs, x0, p, q, sigma_D, lambda, tau, sigma_A, 2, hist + 6, 2, 1, wss, st));      // ... and two more on the same state
        double hh[10];
        HIP_OK(hipMemcpyAsync(hh, hist, sizeof(hh), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        printf("%-22s loss: %.6e -> %.6e\n", "tv_small_cp (2 calls)", hh[1] + lambda * hh[0], hh[9] + lambda * hh[8]);
        for (int it = 0; it < 5; ++it) {
            const double l = hh[2 * it + 1] + lambda * hh[2 * it];
            if (fabs(l - loss[0][it]) > 1e-5 * fabs(loss[0][it])) { fprintf(stderr, "the persistent loop disagrees with the kernel pair at iteration %d\n", it); return 1; }
        }
        for (void* b : {(void*)xs, (void*)x0, (void*)p, (void*)q, wss, (void*)hist}) (void)hipFree(b);
    }
    // ---- sharding from C++: the multi-GPU surface (tv_ctx_*, RCCL underneath) ----------------------------------------------
    // A host with one process per GPU gives every rank a z-slab (tv_geom::z0 / nz_global) and trades boundary planes with
    // tv_halo_exchange.  This box has one GPU, so the one rank here plays BOTH neighbours of a two-slab split: slab A =
    // planes [0, 3), slab B = planes [3, 6); with itself as prev and next peer the messages meet in posting order
    // (first send -> first receive), i.e. what goes out as "send_prev" comes back as "recv_prev".
    {
        unsigned char id[TV_UNIQUE_ID_BYTES];
        tv_ctx* ctx = nullptr;
        TV_OK(tv_ctx_unique_id(id));
        TV_OK(tv_ctx_create(&ctx, 0, 1, id, 0));
        const int64_t plane = g.m * g.ny * g.nx, h = 3;
        float *halo_for_B = nullptr, *halo_for_A = nullptr, *dA = nullptr, *dB = nullptr;
        HIP_OK(hipMalloc(&halo_for_B, plane * sizeof(float)));
        HIP_OK(hipMalloc(&halo_for_A, plane * sizeof(float)));
        HIP_OK(hipMalloc(&dA, h * nd * plane * sizeof(float)));
        HIP_OK(hipMalloc(&dB, (g.nz - h) * nd * plane * sizeof(float)));
        // A's last plane -> B's "previous" halo, B's first plane -> A's "next" halo, one grouped exchange on the stream
        TV_OK(tv_halo_exchange(ctx, TV_F32, plane, 0, 0, x + (h - 1) * plane, x + h * plane, halo_for_B, halo_for_A, st));
        tv_geom ga = g, gb = g;
        ga.nz = h; ga.z0 = 0;
        gb.nz = g.nz - h; gb.z0 = h;
        TV_OK(tv_D(&ga, x, nullptr, halo_for_A, dA, st));                 // enqueued behind the exchange: no host sync
        TV_OK(tv_D(&gb, x + h * plane, halo_for_B, nullptr, dB, st));
        double* red = nullptr;
        HIP_OK(hipMalloc(&red, sizeof(double)));
        const double one = 1.0;
        HIP_OK(hipMemcpyAsync(red, &one, sizeof(double), hipMemcpyHostToDevice, st));
        TV_OK(tv_allreduce_f64(ctx, red, 1, TV_SUM, st));
        HIP_OK(hipStreamSynchronize(st));
        std::vector<float> ha(h * nd * plane), hb((g.nz - h) * nd * plane);
        HIP_OK(hipMemcpy(ha.data(), dA, ha.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hb.data(), dB, hb.size() * sizeof(float), hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < ha.size(); ++i) bad += (ha[i] != hd[i]);
        for (size_t i = 0; i < hb.size(); ++i) bad += (hb[i] != hd[ha.size() + i]);
        double redh = 0.0;
        HIP_OK(hipMemcpy(&redh, red, sizeof(double), hipMemcpyDeviceToHost));
        printf("two z-slabs with exchanged halos vs the whole volume: %zu differing elements; all-reduce over %d rank(s) = %.1f\n",
               bad, tv_ctx_size(ctx), redh);
        TV_OK(tv_ctx_destroy(ctx));
        if (bad != 0 || redh != 1.0) { fprintf(stderr, "sharded D differs from the unsharded one\n"); return 1; }
    }
    // ---- error reporting across the ABI: no exception, a status and a message -----------------------------------------
    g.scheme = 9;
    const int rc = tv_D(&g, x, nullptr, nullptr, d, st);
    printf("bad scheme -> status %d (%s)\n", rc, tv_last_error());
    if (rc >= 0) return 1;
    printf("cabi_demo: OK\n");
    return 0;
}
