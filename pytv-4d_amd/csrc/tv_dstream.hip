// tv_dstream.hip -- instantiations + launcher of the streaming forward kernel (tv_dstream.h).
#include "tv_host.h"
#include "tv_stencil.h"
#include "tv_dstream.h"

namespace tvm {

bool D_stream_ok(const tv_geom* g, const DG& d, bool vec) {
    // fp32 and (round 3) fp64; a weight volume is one more read stream (round 3); pitched arrays incl. ragged rows (round 4: the
    // last lane of a row then holds pad columns, zeros in and -- through d_slots -- zeros out)
    if (!vec || d.nx < 64) return false;
    if (d.s_t * (g->dtype == TV_F32 ? 4 : 8) > (1ll << 32)) return false;      // 32-bit per-lane byte offsets inside a frame
    return true;
}

int D_stream(const tv_geom* g, const DG& d, const void* x, const void* xp, const void* xn, hipStream_t st, void* dout) {
    const int V = (g->dtype == TV_F32) ? 4 : 2;
    const long long tx = ((d.nx + V - 1) / V + ST_BCV - 1) / ST_BCV, ty = (d.ny + ST_BR - 1) / ST_BR;
    // planes per z-chunk: a chunk re-reads one plane (its trailing step), so chunks are long; >= ~4096 blocks in flight
    int zc = env_int("TV_ZCHUNK", 0);
    if (zc <= 0) {
        const long long want = (4096 + tx * ty - 1) / (tx * ty);
        zc = (int)(d.nz / (want > 0 ? want : 1));
        if (zc > 64) zc = 64;
        if (zc < 8) zc = 8;
    }
    if (zc > d.nz) zc = d.nz;
    const long long nch = (d.nz + zc - 1) / zc;
    const long long nwin = (d.m > DS_TWN) ? (d.m + DS_TWN - 1) / DS_TWN : 1;
    const long long nb = tx * ty * nch * nwin, per_xcd = (nb + 7) / 8;
    const dim3 grid((unsigned)(8 * per_xcd), 1, 1), block(64, ST_NWX * ST_NWY, 1);
#define TV_DS_LAUNCH(SC, MM, TW)                                                                                              \
    do {                                                                                                                      \
        if (g->dtype == TV_F32)                                                                                               \
            hipLaunchKernelGGL((k_D_stream<SC, MM, TW, float>), grid, block, 0, st, d, make_w<float>(g), (const float*)x, (const float*)xp, \
                               (const float*)xn, (float*)dout, zc, (int)nch);                                                 \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_D_stream<SC, MM, TW, double>), grid, block, 0, st, d, make_w<double>(g), (const double*)x, (const double*)xp, \
                               (const double*)xn, (double*)dout, zc, (int)nch);                                               \
    } while (0)
#define TV_DS_CASE(SC)                                              \
    case SC:                                                        \
        switch (d.m > DS_TWN ? 0 : d.m) {                           \
            case 0: TV_DS_LAUNCH(SC, DS_TWN, true); break;          \
            case 1: TV_DS_LAUNCH(SC, 1, false); break;              \
            case 2: TV_DS_LAUNCH(SC, 2, false); break;              \
            case 3: TV_DS_LAUNCH(SC, 3, false); break;              \
            case 4: TV_DS_LAUNCH(SC, 4, false); break;              \
            case 5: TV_DS_LAUNCH(SC, 5, false); break;              \
            case 6: TV_DS_LAUNCH(SC, 6, false); break;              \
            case 7: TV_DS_LAUNCH(SC, 7, false); break;              \
            default: TV_DS_LAUNCH(SC, 8, false); break;             \
        }                                                           \
        break;
    switch (g->scheme) {
        TV_DS_CASE(0) TV_DS_CASE(1) TV_DS_CASE(2) TV_DS_CASE(3)
        default: return fail(TV_E_ARG, "unknown scheme");
    }
#undef TV_DS_CASE
#undef TV_DS_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace tvm
