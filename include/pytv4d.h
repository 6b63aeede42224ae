/* pytv4d.h -- C-ABI of the MI355X-native PyTV-4D hot path (libpytv4d_hip.so).
 *
 * Drop-in boundary: these entry points are what a binding of the reference's L2/L3 Python API
 * (pytv/tv_operators_GPU.py, pytv/tv_GPU.py) calls instead of torch.nn.functional.conv3d +
 * slice-assign.  Plain pointers and sizes only; every pointer is a DEVICE pointer unless the
 * comment says otherwise; every function enqueues on the caller's hipStream_t (passed as
 * void*, NULL = default stream) and returns without synchronising.
 *
 * Return value: 0 = ok, <0 = argument error (TV_E_*), >0 = hipError_t of a failed HIP call.
 * tv_last_error() gives a human readable message for the calling thread.
 *
 * Layouts (C-contiguous, cols fastest; pytv/tv_operators_CPU.py:82-83,96-97):
 *   image     x : (nz, m, ny, nx)
 *   gradient  d : (nz, nd, m, ny, nx)      channel axis between z and time
 * Channel order: rows, cols, [z], [t]; hybrid: row-up, col-up, row-down, col-down,
 * [z-up, z-down], [t-up, t-down]  (pytv/tv_operators_CPU.py:117-152).
 *
 * z-slab sharding: a rank holds planes [z0, z0+nz) of a volume of nz_global planes.  Halo
 * arguments point to the neighbouring ranks' boundary planes (m*ny*nx elements each); they may be
 * NULL when the corresponding global plane does not exist or is not needed by the scheme.
 */
#ifndef PYTV4D_H
#define PYTV4D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TV_UPWIND   0
#define TV_DOWNWIND 1
#define TV_CENTRAL  2
#define TV_HYBRID   3

#define TV_F32 0
#define TV_F64 1

#define TV_E_ARG      (-1)   /* NULL / negative size / unknown enum            */
#define TV_E_HALO     (-2)   /* a halo plane the scheme needs was not supplied */
#define TV_E_CHANNELS (-3)   /* nd does not match scheme and active axes       */

/* Geometry and weights of one z-slab.  Mirrors the keyword arguments every reference operator
 * takes: reg_z_over_reg, reg_time, mask_static, factor_reg_static
 * (pytv/tv_operators_GPU.py:134,253,362,471,583,719,828,938; pytv/tv_GPU.py:47,142,217,290). */
/* Version of the binary interface declared in this header: bumped whenever tv_geom, the workspace layout or the
 * meaning of an entry point's arguments changes incompatibly (round 1: 1; round 2 added the three weight-volume
 * pointers and a second partial-sum array in the workspace: 2; round 3 added the two leading fields below: 3;
 * round 4 added row_pitch / frame_pitch at the end: 4; round 5 grew the workspace to THREE partial-sum arrays
 * (TV_CP_FID_BOTH, tv_cheb_step(dots = NULL)), gave tv_cp_sweep its flags and changed tv_cheb_step's NULL-dots contract -- stamped as 5
 * in round 6, together with the persistent small-volume entry points tv_small_*).
 * Every entry point that takes a tv_geom rejects a struct whose struct_size / abi_version are not the library's own
 * with TV_E_ARG -- a host built against an older header fails loudly instead of having its trailing fields read as
 * garbage.  tv_geom_init() fills the two fields in. */
#define TV_ABI_VERSION 5

typedef struct tv_geom {
    uint32_t struct_size;       /* sizeof(tv_geom) of the header the HOST was compiled against   */
    uint32_t abi_version;       /* TV_ABI_VERSION of that header                                  */
    int64_t nz;                 /* planes held by this rank                                  */
    int64_t m, ny, nx;          /* time frames, rows, cols                                   */
    int64_t nz_global;          /* planes of the whole volume (== nz when not sharded)       */
    int64_t z0;                 /* global index of local plane 0                             */
    int32_t scheme;             /* TV_UPWIND .. TV_HYBRID                                    */
    int32_t dtype;              /* TV_F32 / TV_F64                                           */
    double  reg_z_over_reg;     /* z weight; z axis active iff nz_global > 1 and this > 0    */
    double  reg_time;           /* time weight; time axis active iff m > 1 and this > 0      */
    double  factor_reg_static;  /* time channels scaled by sqrt(this) where mask_static != 0 */
    const uint8_t* mask_static; /* device, ny*nx bytes, or NULL (the reference's `False`)    */
    const void* time_factor;    /* device, ny*nx elements of `dtype`, or NULL: per-pixel multiplier f(y,x) of the
                                 * time channels, i.e. f^2 is a per-pixel weight on reg_time.  Generalises
                                 * (mask_static, factor_reg_static), which is f = mask ? sqrt(factor) : 1 -- the
                                 * "weight matrix" of the reference's to-do list (README.md:258); both may be set,
                                 * the factors multiply */
    const void* time_weight_vol;/* device, nz*m*ny*nx elements of `dtype`, or NULL: per-VOXEL multiplier f(z,t,y,x) of the
                                 * time channels (f^2 = weight on reg_time at that voxel): the reference's to-do "weight matrix
                                 * of size Nz x M x N x N" (README.md:258), local planes of the slab.  Multiplies with the two
                                 * per-pixel forms above.  D scales the time channel of a voxel by f at THAT voxel; D^T is the
                                 * exact adjoint (every sample is scaled by its own voxel's f before the difference), which for
                                 * a weight that does not vary along t is the reference's "scale the time part at the output
                                 * voxel" (pytv/tv_operators_CPU.py:442-446); the sub-gradient keeps its unit-weight adjoint
                                 * (the weight enters through D only, pytv/tv_GPU.py:104-122).  The plane-marching, one-sweep and
                                 * one-pass fast paths do not take a weight volume (tv_cp_fused_supported /
                                 * tv_subgrad_fused_supported answer 0): the one-site-per-thread kernels do the work */
    const void* time_weight_prev;/* plane z0-1 / z0+nz of the same volume (m*ny*nx elements each) or NULL: only the          */
    const void* time_weight_next;/* ghost-plane norms of tv_subgrad on a slab read them                                       */
    /* PITCHED arrays (round 4; 0 = dense, the reference's layout).  Every image-like array of a call (x, x0, p, G, norms, halo
     * planes, a weight volume) and every gradient-like array (d, q, z, u: one image per channel) shares them:
     *   row_pitch   : elements from one row of a frame to the next  (>= nx; a multiple of 16 bytes)
     *   frame_pitch : elements from one frame to the next           (>= ny * row_pitch; a multiple of 16 bytes)
     * element (z, [c,] t, y, x) lives at ((z [* nd + c]) * m + t) * frame_pitch + y * row_pitch + x.  mask_static / time_factor stay
     * dense (ny * nx).  Why: (i) frames whose rows are not multiples of 128 bytes -- the reference's own shapes, pytv/tests.py:48
     * N = 100, README.md:76-79 rand(20, 4, 100, 100) -- straddle cache lines and write sectors (1000-column frames ran at 0.55 x the
     * rate of 1024-column ones in round 3); (ii) the one-sweep kernels read ~20 streams per block that are congruent modulo the
     * frame size: a frame pitch that is not a power of two spreads them over the HBM channels.
     * The PAD elements (columns >= nx of a row, the tail of a frame) belong to the library's caller but must hold ZEROS on entry
     * wherever an array is read; every pitch-aware entry point leaves them zero (or untouched).  An entry point that does not
     * take pitched arrays refuses them with TV_E_ARG (tv_last_error names it) -- nothing reads a pitched array as a dense one.
     * nx need not be a multiple of the 16-byte lane (4 floats / 2 doubles) when row_pitch is: the vector kernels then own the pad
     * columns of the last lane (this is how 1001-column frames leave the scalar kernels). */
    int64_t row_pitch;
    int64_t frame_pitch;
} tv_geom;

/* zero a tv_geom and stamp it with the header's struct size and interface version */
static inline void tv_geom_init(tv_geom* g) {
    unsigned char* b = (unsigned char*)g;
    for (size_t i = 0; i < sizeof(tv_geom); ++i) b[i] = 0;
    g->struct_size = (uint32_t)sizeof(tv_geom);
    g->abi_version = TV_ABI_VERSION;
}

/* ---- housekeeping ------------------------------------------------------------------------ */
const char* tv_last_error(void);
int         tv_version(void);                         /* 10000*major + 100*minor + patch     */
int         tv_abi_version(void);                     /* TV_ABI_VERSION the library was built with */
/* Tuning / debugging options (TV_ZCHUNK, TV_NO_FUSED, ...: DESIGN.md section 7).  The table is process-wide and explicit:
 * every entry is initialised ONCE from the environment variable of the same name when the library first consults it,
 * and afterwards changes only through these calls -- no entry point reads the environment per call.
 *   tv_set_option   : 0, or TV_E_ARG for an unknown name
 *   tv_unset_option : back to "not set" (every call site then uses its built-in default)
 *   tv_get_option   : the value, or dflt when the option is not set                                            */
int         tv_set_option(const char* name, int value);
int         tv_unset_option(const char* name);
int         tv_get_option(const char* name, int dflt);
/* Number of gradient channels nd for this geometry (pytv/tv_operators_GPU.py:170-174,
 * 290-294,399-403,507-511), or TV_E_ARG. */
int         tv_num_channels(const tv_geom* g);
/* Bytes of device scratch the reducing entry points need in `ws` for this geometry. */
size_t      tv_workspace_bytes(const tv_geom* g);

/* ---- operators: replace pytv.tv_operators_GPU.D_* / D_T_* / compute_L21_norm ---------------- */
/* d = D x.  x_prev / x_next: the plane z0-1 / z0+nz of the image (or NULL).
 * Replaces D_hybrid/D_downwind/D_upwind/D_central (pytv/tv_operators_GPU.py:134-581). */
int tv_D(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* d, void* stream);

/* out = D^T y.  y_prev: plane z0-1 of the z channel whose adjoint looks backwards (upwind /
 * central z channel; hybrid: z-up channel); y_next: plane z0+nz of the channel whose adjoint
 * looks forwards (downwind / central z channel; hybrid: z-down channel).
 * Replaces D_T_hybrid/D_T_downwind/D_T_upwind/D_T_central (pytv/tv_operators_GPU.py:583-1052). */
int tv_DT(const tv_geom* g, const void* y, const void* y_prev, const void* y_next, void* out, void* stream);

/* *result (device, fp64) = sum_p sqrt(sum_c d[p,c]^2); norms (nz,m,ny,nx) optional (NULL = skip).
 * Replaces compute_L21_norm (pytv/tv_operators_GPU.py:46-90). */
int tv_l21(const tv_geom* g, const void* d, int32_t nd, void* norms, double* result, void* ws, void* stream);

/* ---- direct TV API: replaces pytv.tv_GPU.tv_* ------------------------------------------------ */
/* *tv (device fp64) = TV of the local planes; G = the reference's sub-gradient
 * (pytv/tv_GPU.py:47-375).  norms_ext: REQUIRED scratch/output of (nz + 2) planes; on return
 * plane k+1 holds 1 / |D x| of local plane k, 0 where |D x| == 0 -- the reciprocal of the
 * reference's grad_norms (which has those zeros replaced by +inf, pytv/tv_GPU.py:88).  "== 0" means here, and in every
 * other sub-gradient entry point: |D x|^2 below the smallest NORMAL number of the dtype (fp32: |D x| < 1.1e-19), where a
 * square carries no information any more; the reference zeroes only at exactly 0.
 * x_prev / x_next: TWO planes each (z0-2, z0-1) / (z0+nz, z0+nz+1) when sharded, else NULL. */
int tv_subgrad(const tv_geom* g, const void* x, const void* x_prev, const void* x_next,
               void* G, void* norms_ext, double* tv, void* ws, void* stream);

/* The same TV value and sub-gradient in ONE pass over x (1/|Dx| never leaves the chip): all four schemes (central:
 * not with a two-point z or time axis), fp32, Nx % 4 == 0, any M (more than 8 frames: overlapping time windows), 16-byte aligned arrays
 * (tv_subgrad_fused_supported).  Use it when the per-voxel norms are not wanted (return_grad_norms=False,
 * pytv/tv_GPU.py:47).  Halos as for tv_subgrad.  A |Dx|^2 below the smallest normal fp32 number counts as 0. */
int tv_subgrad_fused_supported(const tv_geom* g);
int tv_subgrad_fused(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* G, double* tv,
                     void* ws, void* stream);
/* The same with the per-voxel norms as a by-product: norms (nz,m,ny,nx) = |D x| with zeros replaced by +inf -- the
 * reference's grad_norms (return_grad_norms=True, pytv/tv_GPU.py:47,88,135-139).  Three words per voxel. */
int tv_subgrad_fused_norms(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* G, void* norms,
                           double* tv, void* ws, void* stream);
/* One iteration of the README's sub-gradient loop (README.md:118-124) in the same single pass, G never stored:
 *   x_out = x - step * ((x - x0) + lambda * G(x));  *tv = TV(x);  *fid = 1/2 |x_out - x0|^2   (local planes).
 * x and x_out must be different buffers (ping-pong).  Same geometries as tv_subgrad_fused. */
int tv_subgrad_step_fused(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, const void* x0,
                          void* x_out, double step, double lambda, double* tv, double* fid, void* ws, void* stream);

/* ---- fused Chambolle-Pock inner loop (README.md:141-157) ----------------------------------- */
/* q <- proj_{|.|_2 <= lambda}(q + sigma_D * D x); *tv (device fp64) = |D x|_{2,1} of local planes. */
int tv_cp_dual(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* q,
               double sigma_D, double lambda, double* tv, void* ws, void* stream);
/* p <- (p + sigma_A (x - x0)) / (1 + sigma_A);  x <- x - tau p - tau D^T q;
 * *fid (device fp64) = 1/2 |x_new - x0|^2 of local planes.  q_prev/q_next as y_prev/y_next in tv_DT. */
int tv_cp_primal(const tv_geom* g, const void* q, const void* q_prev, const void* q_next, void* x,
                 const void* x0, void* p, double tau, double sigma_A, double* fid, void* ws, void* stream);

/* One-sweep form of the same iteration (all four schemes; fp32, nx % 4 == 0, nx >= 64,
 * any m: more than 8 frames are processed as time windows of 8): q is read and written ONCE per
 * iteration.  x is ping-ponged.
 *   tv_cp_fused_supported : 1 if this geometry can take the one-sweep path, else 0
 *   tv_cp_fused           : q <- proj(q + sigma_D D x_in); p <- (p + sigma_A (x_in - x0)) / (1 + sigma_A);
 *                           x_out <- x_in - tau p - tau D^T q   EXCEPT the adjoint terms that cross a
 *                           wave tile (8 rows x 32 cols; columns: only the 256-col block tile), a z-chunk, a time window or the slab; *tv = |D x_in|_{2,1},
 *                           *fid = 1/2 |x_out - x0|^2 over the sites that are already complete
 *   tv_cp_fixup           : adds the missing terms to x_out (q_prev / q_next as in tv_DT) and returns
 *                           the fidelity of those sites in *fid; total fidelity = sum of the two.
 * Both can be restricted to a range so that halo exchanges hide behind the interior work: the sweep to
 * z-chunks [chunk_begin, chunk_begin + chunk_count) (tv_cp_zchunk() planes each, chunk_count < 0 = all),
 * the fix-up to local planes [z_begin, z_begin + z_count) (z_count < 0 = all).  Only the first chunk reads
 * x_prev, only the last x_next; only plane 0 reads q_prev, only plane nz-1 q_next.                         */
int tv_cp_fused_supported(const tv_geom* g);
int tv_cp_zchunk(const tv_geom* g);
int tv_cp_fused(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, void* q, const void* x0,
                void* p, void* x_out, double sigma_D, double lambda, double tau, double sigma_A, int64_t chunk_begin,
                int64_t chunk_count, double* tv, double* fid, void* ws, void* stream);
int tv_cp_fixup(const tv_geom* g, const void* q, const void* q_prev, const void* q_next, void* x_out, const void* x0,
                double tau, int64_t z_begin, int64_t z_count, double* fid, void* ws, void* stream);
/* tv_cp_fused with flags (round 4).  TV_CP_FID_OF_INPUT: *fid = 1/2 |x_in - x0|^2 over ALL local sites (the sweep has x_in
 * and x0 in registers) instead of 1/2 |x_out - x0|^2 over the sites that are already complete; tv_cp_fixup then needs no x0
 * (pass x0 = NULL: it adds the missing terms only, *fid = 0) and moves one word less per site it visits.  A loop takes the
 * fidelity of iterate k+1 from sweep k+1 (the README's loss line, README.md:157, pairs 1/2 |x_{k+1} - x0|^2 with the TV of x_k)
 * and needs one plain reduction for the last iterate.
 * q_in / q_out: the dual variable is read from q_in and written to q_out.  q_in == q_out is the in-place update of tv_cp_fused;
 * two arrays (ping-pong, the caller swaps them after every iteration) cost a second q but run ~9 % faster: HBM serves "read one
 * array, write another" better than a read-modify-write of the same lines (tools/archive/bwtest4: 5.98 against 5.50 TB/s for this
 * kernel's memory shape).  tv_cp_fixup takes q_out. */
#define TV_CP_FID_OF_INPUT 1
/* TV_CP_FID_BOTH (round 5; only together with TV_CP_FID_OF_INPUT): `fid` points to TWO doubles -- fid[0] as above, fid[1] = 1/2 |x_out - x0|^2 over
 * the sites that are already complete (what the flag-less tv_cp_fused returns); the tv_cp_fixup that follows is then called WITH x0 and
 * returns the rest.  The LAST sweep of a lagged loop uses it: the fidelity of the final iterate needs no separate reduction pass. */
#define TV_CP_FID_BOTH 2
int tv_cp_sweep(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, const void* q_in, void* q_out, const void* x0,
                void* p, void* x_out, double sigma_D, double lambda, double tau, double sigma_A, int32_t flags, int64_t chunk_begin,
                int64_t chunk_count, double* tv, double* fid, void* ws, void* stream);

/* ---- fused ADMM updates (not in the reference; README.md:26,135 mention only) --------------- */
/* v = D x + u; z = v * max(0, 1 - thresh/|v|_2); u = v - z; *tv (device fp64) = |D x|_{2,1}. */
int tv_admm_zu(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* z, void* u,
               double thresh, double* tv, void* ws, void* stream);
/* One-sweep form of the outer iteration's dual side (round 3; same geometries as tv_cp_fused: tv_cp_fused_supported):
 * the z / u update AND the residual the next x-solve starts from, from one pass over u --
 *   v = D x + u;  z = v * max(0, 1 - thresh/|v|_2);  u <- v - z (in place);  t' = (z - u) - D x;
 *   r <- (x0 - x) + rho D^T t'  =  [x0 + rho D^T (z - u)] - (I + rho D^T D) x
 *   *tv = |D x|_{2,1};  *rr = <r, r> over the sites that are already complete (device fp64, local planes)
 * EXCEPT, exactly as in tv_cp_fused, the adjoint terms that cross a wave tile, a z-chunk, a time window or the slab:
 * tv_admm_fixup adds those to r (t_prev / t_next: halo planes of t' as y_prev / y_next in tv_DT) and returns <r, r> of
 * the sites it completed; total = sum of the two.  t (same shape as u) receives t': every sample if bit 0 of full_store is set
 * (then z = t' + u + D x), otherwise only the samples the fix-up and the neighbouring ranks read (the rest of the array is
 * left untouched).  Bit 1 of full_store: *rr = |x - x0|^2 over ALL local sites instead (a solver that needs no <r, r> -- the Chebyshev
 * x-solve -- gets the fidelity of the iterate for free; tv_admm_fixup's *rr is then meaningless).
 * 2 Nd + 3 words per voxel where tv_admm_tu + tv_DT_axpy + tv_normal_op2(b) move 4 Nd + 6.
 * Chunk / plane ranges as in tv_cp_fused / tv_cp_fixup.  Replaces: nothing in the reference (README.md:26,135 name ADMM only). */
int tv_admm_fused(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* u, void* t, const void* x0, void* r,
                  double thresh, double rho, int32_t full_store, int64_t chunk_begin, int64_t chunk_count, double* tv, double* rr,
                  void* ws, void* stream);
/* tv_admm_fused with separate arrays for the dual variable (round 5): u_in is read, u_out written (equal: tv_admm_fused).  A caller that
 * keeps both and swaps them after every outer iteration can rebuild the split variable z = shrink(D x + u_in, thresh) whenever it is
 * asked for (tv_admm_zu on a copy of u_in), so the sweep need not store every sample of t' (full_store bit 0 clear): Nd words per voxel
 * and outer iteration less than the z-preserving form of tv_admm_fused.  Replaces: nothing in the reference (README.md:26,135). */
int tv_admm_sweep(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, const void* u_in, void* u_out, void* t,
                  const void* x0, void* r, double thresh, double rho, int32_t full_store, int64_t chunk_begin, int64_t chunk_count,
                  double* tv, double* rr, void* ws, void* stream);
int tv_admm_fixup(const tv_geom* g, const void* t, const void* t_prev, const void* t_next, void* r, double rho, int64_t z_begin,
                  int64_t z_count, double* rr, void* ws, void* stream);
/* One-sweep form of Chambolle-Pock with a data-fidelity operator A (README.md:2,148 with A != I; same geometries as tv_cp_fused):
 *   tv_cpop_fused : q <- proj(q + sigma_D D x_in);  x_out <- x_in - tau atp - tau D^T q   (atp = A^T p, an image, read only)
 *                   EXCEPT the adjoint terms tv_cp_fused leaves out;  *tv = |D x_in|_{2,1}
 *   tv_cpop_fixup : adds those terms to x_out (q_prev / q_next as in tv_DT)
 * 2 Nd + 4 words per voxel where tv_cp_dual + tv_DT_axpy2 move 3 Nd + 4.  Chunk / plane ranges as in tv_cp_fused / tv_cp_fixup. */
int tv_cpop_fused(const tv_geom* g, const void* x_in, const void* x_prev, const void* x_next, void* q, const void* atp, void* x_out,
                  double sigma_D, double lambda, double tau, int64_t chunk_begin, int64_t chunk_count, double* tv, void* ws, void* stream);
int tv_cpop_fixup(const tv_geom* g, const void* q, const void* q_prev, const void* q_next, void* x_out, double tau, int64_t z_begin,
                  int64_t z_count, void* ws, void* stream);
/* out = base + alpha * D^T (a - b)   (b and/or base may be NULL).  ab_prev / ab_next: halo planes
 * of (a - b) for the channels named in tv_DT. */
int tv_DT_axpy2(const tv_geom* g, const void* a, const void* b, const void* ab_prev, const void* ab_next,
                const void* base, const void* base2, double beta, double alpha, void* out, void* stream);
/* ^ out = base + beta * base2 + alpha * D^T (a - b): the primal step of Chambolle-Pock with a data-fidelity operator A
 *   (the CT use case, README.md:2,148), x - tau A^T p - tau D^T q in ONE pass: base = x, base2 = A^T p, beta = alpha = -tau.
 * The two data-space updates of that iteration (vectors of ANY length n, the user's A decides):
 *   tv_cpop_p        : p <- (p + sigma_A r) / (1 + sigma_A),  r = A x - b carried over from the previous iteration
 *   tv_cpop_residual : r <- ax - b;  *fid (device fp64) = 1/2 |r|^2;  ws: >= 2048 doubles of device scratch */
int tv_cpop_p(int32_t dtype, int64_t n, void* p, const void* r, double sigma_A, void* stream);
int tv_cpop_residual(int32_t dtype, int64_t n, const void* ax, const void* b, void* r, double* fid, void* ws, void* stream);
int tv_DT_axpy(const tv_geom* g, const void* a, const void* b, const void* ab_prev, const void* ab_next,
               const void* base, double alpha, void* out, void* stream);
/* out = x + rho * D^T D x, computed from x alone (radius-2 stencil); *dot (device fp64) = <x, out>
 * over local planes.  x_prev / x_next: TWO planes each as in tv_subgrad. */
int tv_normal_op(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, double rho,
                 void* out, double* dot, void* ws, void* stream);
/* The same operator with two dot products and an optional right-hand side, for conjugate gradients:
 *   b == NULL : out = A x,      dots[0] = <x, A x>,   dots[1] = <x, x>
 *   b != NULL : out = b - A x,  dots[0] = <out, out>, dots[1] = <x, x>;  out2 (or NULL) receives a second copy of out
 * A = I + rho D^T D; dots: two device fp64 words (local planes). */
int tv_normal_op2(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, double rho, const void* b, void* out,
                  void* out2, double* dots, void* ws, void* stream);
/* One step of the Chebyshev iteration on (I + rho D^T D) e = b (round 3: an x-solve whose scalars do not depend on the vectors --
 * no dot product, hence no all-reduce on a sharded volume, and 4 words per voxel and step where a CG step moves 11):
 *   out = [add +] x + alpha (b - A x) + beta (x - y)          y NULL: y = yscale * b (0: no y); add, ref may be NULL
 *   (x may be b itself: with e_1 = a_0 b never stored, the first two steps are  e_2 = b + a' (b - A b) + b' b  and a step with
 *   y = a_0 b -- 2 + 3 words instead of 2 + 3 + 4)
 *   dots[0] = |b - A x|^2,  dots[1] = |out - ref|^2 if ref is given, else |x|^2      (device fp64, local planes)
 *   dots may be NULL (round 5): no dot products -- the streaming kernel drops their fp64 arithmetic and the two reduction launches
 *   (the ADMM x-solve needs none of them unless it is asked for the fidelity of its last step); ref needs dots.
 * out must not alias an input.  x_prev / x_next: TWO halo planes each, as in tv_normal_op.  The coefficients of step k follow from
 * the spectral interval [1, 1 + rho L] alone (pytv/solvers.py::chebyshev_coefficients restates the recurrence).
 * tv_axpby: out = a x + b y (y NULL: a x); *dist2 (or NULL, then ref and ws may be NULL too) = |out - ref|^2; out NULL (with ref):
 * the distance alone, nothing is stored (round 4: the fidelity of the last iterate of a lagged Chambolle-Pock block).
 * Replaces: nothing in the reference (its README names ADMM only, README.md:26,135). */
int tv_cheb_step(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, double rho, const void* b, const void* y,
                 double yscale, const void* add, const void* ref, double alpha, double beta, void* out, double* dots, void* ws, void* stream);
int tv_axpby(const tv_geom* g, double a, const void* x, double b, const void* y, const void* ref, void* out, double* dist2, void* ws,
             void* stream);
/* One step of the SINGLE-REDUCTION conjugate gradient (Chronopoulos-Gear form; one all-reduce of two scalars per step on
 * a sharded volume instead of two all-reduces).  sc: four device fp64 words
 *     sc[0] = gamma = <r, r>,  sc[1] = delta = <r, w> with w = A r   (already summed over the ranks)
 *     sc[2] = gamma of the previous step,  sc[3] = alpha of the previous step; sc[3] == 0 marks the FIRST step of a solve
 *   beta = first ? 0 : gamma / gamma_old;   alpha = gamma / (first ? delta : delta - beta gamma / alpha_old)
 *   d = r + beta d;  s = w + beta s (= A d);  x += alpha d;  r -= alpha s;   then sc[2] = gamma, sc[3] = alpha
 * x0 != NULL: *fid = 1/2 |x_new - x0|^2 (the ADMM loss after the last step, for one extra read). */
int tv_cg_update(const tv_geom* g, void* x, void* r, void* d, void* s, const void* w, double* sc, const void* x0, double* fid,
                 void* ws, void* stream);
/* The z / u update of tv_admm_zu storing t = z - u_new in place of z: the next right-hand side x0 + rho D^T (z - u) is
 * then tv_DT_axpy(t, NULL, ..) over ONE gradient array; z = t + u whenever it is wanted. */
int tv_admm_tu(const tv_geom* g, const void* x, const void* x_prev, const void* x_next, void* t, void* u,
               double thresh, double* tv, void* ws, void* stream);
/* Conjugate-gradient vector updates with device-resident scalars (no host round trip):
 *   tv_cg_step1: alpha = rs/dAd;  x += alpha d;  r -= alpha Ad;  *rs_new = <r, r>
 *   tv_cg_step2: beta = rs_new/rs; d = r + beta d                                            */
int tv_cg_step1(const tv_geom* g, void* x, void* r, const void* d, const void* Ad, const double* rs,
                const double* dAd, double* rs_new, void* ws, void* stream);
int tv_cg_step2(const tv_geom* g, void* d, const void* r, const double* rs_new, const double* rs, void* stream);

/* ---- small vector helpers on (nz,m,ny,nx) arrays -------------------------------------------- */
/* out = a - b (used for ADMM halos and residuals) */
int tv_sub(int32_t dtype, int64_t n, const void* a, const void* b, void* out, void* stream);
/* *result (device fp64) = <a, b> */
int tv_dot(const tv_geom* g, const void* a, const void* b, double* result, void* ws, void* stream);
/* x <- x - step * ((x - x0) + lambda * G); *fid = 1/2 |x_new - x0|^2  (README.md:122-123) */
int tv_subgrad_step(const tv_geom* g, void* x, const void* x0, const void* G, double step, double lambda,
                    double* fid, void* ws, void* stream);

/* ---- persistent small-volume loops (round 6, interface version 5) ------------------------------------------------------------------
 * The reference's user loops (README.md:107-124 sub-gradient descent, :141-157 Chambolle-Pock) on volumes that fit the caches
 * (its own shapes: README.md:76-79 rand(20,4,100,100), 256 x 256 / 512 x 512 images, pytv/tests.py:48 N = 100): n_iter iterations in
 * ONE cooperative launch -- every block keeps its sites for the whole loop and waits only for the blocks that own neighbouring sites
 * between the two phases of an iteration; arrays that neighbours read are accessed agent-coherently (csrc/tv_small.hip).
 * Unsharded volumes only (g->nz == g->nz_global, no halos); any scheme, fp32 / fp64, any nx (16-byte lanes when nx and the pointers
 * allow), pitched arrays, weight maps / volumes.  The stream must not run other kernels of the caller concurrently with these
 * launches in a way that could starve them of compute units (the blocks of a launch wait for each other).
 *   tv_small_supported       : 1 if the geometry fits (<= TV_SMALL_MAX_KVOXELS thousand voxels, default 4096; arrays below 2^31 bytes)
 *   tv_small_workspace_bytes : bytes of the scratch buffer `ws` for n_iter iterations per launch (block flags + per-block partials).
 *                              `ws` must be ZERO-FILLED once before its first use (hipMemset); the calls keep it consistent afterwards
 *                              (the blocks' phase counters continue from call to call: no reset per launch).  One `ws` per stream.
 *                              Safety net: should the blocks of a launch not all become resident (they wait for each other), the launch
 *                              abandons itself after ~2 s of polling instead of hanging: every hist entry of the call is NaN, the state
 *                              arrays are undefined and `ws` must be zero-filled again.
 *   tv_small_cp              : n_iter iterations of  p <- (p + sigma_A (x - x0)) / (1 + sigma_A);  q <- proj(q + sigma_D D x);
 *                              x <- x - tau p - tau D^T q  exactly as tv_cp_dual + tv_cp_primal compute them, x / p / q updated in
 *                              place.  hist (device fp64): hist[k * hist_stride] = |D x_k|_{2,1} (the iterate the dual update of
 *                              iteration k saw), hist[k * hist_stride + hist_fid_offset] = 1/2 |x_{k+1} - x0|^2
 *                              (a dense (n_iter, 2) array: stride 2, offset 1; 0 < hist_fid_offset < hist_stride)
 *   tv_small_subgrad_descent : n_iter iterations of  x <- x - step ((x - x0) + lambda G(x))  (tv_subgrad + tv_subgrad_step's
 *                              arithmetic); the iterate is ping-ponged between x and x_alt: after an ODD n_iter the result is in
 *                              x_alt, after an even one in x; norms_ext: (nz + 2) planes of scratch (1 / |D x|);
 *                              hist as above: TV(x_k) and 1/2 |x_{k+1} - x0|^2 */
int    tv_small_supported(const tv_geom* g);
size_t tv_small_workspace_bytes(const tv_geom* g, int64_t n_iter);
int    tv_small_cp(const tv_geom* g, void* x, const void* x0, void* p, void* q, double sigma_D, double lambda, double tau,
                   double sigma_A, int64_t n_iter, double* hist, int64_t hist_stride, int64_t hist_fid_offset, void* ws, void* stream);
int    tv_small_subgrad_descent(const tv_geom* g, void* x, void* x_alt, const void* x0, void* norms_ext, double step, double lambda,
                                int64_t n_iter, double* hist, int64_t hist_stride, int64_t hist_fid_offset, void* ws, void* stream);

/* ---- multi-GPU: z-slab neighbours over RCCL, one process per GPU ------------------------------ */
/* The reference is single-GPU (its README only remarks that the (Nz, M, N, N) layout "can be decomposed easily along z",
 * README.md:235).  Rank r holds the planes [z0, z0 + nz) (tv_geom::z0 / nz_global) and, per operator apply, trades the
 * boundary plane(s) named by the halo arguments above with ranks r-1 / r+1: a chain, not a ring.
 *   tv_ctx_unique_id : rank 0 fills 128 bytes; the host program hands them to every rank by its own means
 *   tv_ctx_create    : collective over the nranks processes (ncclCommInitRank); `device` = HIP device of this process
 *   tv_halo_exchange : ONE grouped exchange on `stream`, in stream order, no host synchronisation: `count` elements of
 *                      `dtype` to / from prev_rank and next_rank (-1 = no such neighbour; NULL buffer = nothing that way).
 *                      Matching is per peer in posting order: a rank's send_next meets its next rank's recv_prev.
 *   tv_allreduce_f64 : in-place sum / max of n fp64 device words over all ranks, on `stream`
 * RCCL is bound at run time (dlopen): hosts that never create a context never load it.
 * Errors: 0 ok, < 0 argument, 1..999 hipError_t, 1000 + ncclResult_t. */
#define TV_UNIQUE_ID_BYTES 128
#define TV_SUM 0
#define TV_MAX 1
typedef struct tv_ctx tv_ctx;
int tv_ctx_unique_id(void* id_out);
int tv_ctx_create(tv_ctx** ctx, int rank, int nranks, const void* unique_id, int device);
int tv_ctx_destroy(tv_ctx* ctx);
int tv_ctx_rank(const tv_ctx* ctx);
int tv_ctx_size(const tv_ctx* ctx);
int tv_halo_exchange(tv_ctx* ctx, int32_t dtype, int64_t count, int prev_rank, int next_rank, const void* send_prev,
                     const void* send_next, void* recv_prev, void* recv_next, void* stream);
int tv_allreduce_f64(tv_ctx* ctx, double* buf, int64_t n, int32_t op, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PYTV4D_H */
