// issue_bench.hip -- what does one instruction cost on a gfx950 SIMD?  (round 3: the one-pass sub-gradient kernel is
// issue-bound, so its budget is instructions, and the budget depends on which instructions.)
//
// Every test is a loop of 64 independent instructions of ONE kind over 8 register sets (no dependency closer than 8
// instructions), run by W waves per SIMD on every CU; cycles per instruction and SIMD = s_memtime ticks of a wave /
// (instructions it issued) / ... reported both per wave (latency-ish when W = 1) and per SIMD (throughput: ticks * 1 /
// (instr * W)).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/issue_bench.hip -o tools/issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
// 8 instructions, one per register set k: v[8+2k:9+2k] are 64-bit pairs, v[k] singles
#define I_FMA(k) "v_fma_f32 v" #k ", v" #k ", v30, v31\n"
#define I_ADD(k) "v_add_f32 v" #k ", v" #k ", v30\n"
#define I_MUL(k) "v_mul_f32 v" #k ", v" #k ", v30\n"
#define I_MOV(k) "v_mov_b32 v" #k ", v30\n"
#define I_CND(k) "v_cndmask_b32 v" #k ", v30, v31, vcc\n"
#define I_CMP(k) "v_cmp_le_f32 vcc, v30, v" #k "\n"
#define I_RSQ(k) "v_rsq_f32 v" #k ", v" #k "\n"
#define I_SQRT(k) "v_sqrt_f32 v" #k ", v" #k "\n"
#define I_RCP(k) "v_rcp_f32 v" #k ", v" #k "\n"
#define P(k) "v[" #k "*2+8:" #k "*2+9]"
#define I_PKFMA(k) "v_pk_fma_f32 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "], v[28:29], v[30:31]\n"
#define I_PKADD(k) "v_pk_add_f32 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "], v[30:31]\n"
#define I_PKMUL(k) "v_pk_mul_f32 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "], v[30:31]\n"
#define I_FMA64(k) "v_fma_f64 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "], v[28:29], v[30:31]\n"
#define I_ADD64(k) "v_add_f64 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "], v[30:31]\n"
#define I_MUL64(k) "v_mul_f64 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "], v[30:31]\n"
#define I_RSQ64(k) "v_rsq_f64 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "]\n"
#define I_SQRT64(k) "v_sqrt_f64 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "]\n"
#define I_CVT64(k) "v_cvt_f64_f32 v[8+2*" #k ":9+2*" #k "], v" #k "\n"
#define I_DPPMOV(k) "v_mov_b32_dpp v" #k ", v30 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPPADD(k) "v_add_f32_dpp v" #k ", v30, v" #k " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPPWSHR(k) "v_mov_b32_dpp v" #k ", v30 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_BPERM(k) "ds_bpermute_b32 v" #k ", v26, v30\n"
#define I_SWZ(k) "ds_swizzle_b32 v" #k ", v30 offset:swizzle(SWAP,16)\n"
#define I_PL32(k) "v_permlane32_swap_b32 v" #k ", v3" #k "\n"
#define I_PL16(k) "v_permlane16_swap_b32 v" #k ", v3" #k "\n"
#define I_DSR128(k) "ds_read_b128 v[32+4*" #k ":35+4*" #k "], v27 offset:" #k "*1024\n"
#define I_DSR64(k) "ds_read_b64 v[32+4*" #k ":33+4*" #k "], v27 offset:" #k "*1024\n"
#define I_DSR32(k) "ds_read_b32 v" #k ", v27 offset:" #k "*1024\n"
#define I_DSW128(k) "ds_write_b128 v27, v[32+4*" #k ":35+4*" #k "] offset:" #k "*1024\n"
#define I_DSW32(k) "ds_write_b32 v27, v" #k " offset:" #k "*1024\n"
#define I_SALU(k) "s_add_u32 s2" #k ", s2" #k ", 1\n"
#define I_CNDS(k) "v_cndmask_b32_e64 v" #k ", v30, v31, s[22:23]\n"
#define I_CMPS(k) "v_cmp_le_f32_e64 s[24:25], v30, v" #k "\n"
#define I_CMPCND(k) "v_cmp_le_f32 vcc, v30, v" #k "\nv_cndmask_b32 v" #k ", v30, v31, vcc\n"
#define I_MAX(k) "v_max_f32 v" #k ", v" #k ", v30\n"
#define I_MIN3(k) "v_min3_f32 v" #k ", v" #k ", v30, v31\n"
#define I_MULLEG(k) "v_mul_legacy_f32 v" #k ", v" #k ", v30\n"
#define I_DSR32C(k) "ds_read_b32 v" #k ", v25 offset:" #k "*256\n"
#define I_DSW32C(k) "ds_write_b32 v25, v" #k " offset:" #k "*256\n"
#define I_DSR64C(k) "ds_read_b64 v[32+4*" #k ":33+4*" #k "], v24 offset:" #k "*512\n"
#define I_DSW64C(k) "ds_write_b64 v24, v[32+4*" #k ":33+4*" #k "] offset:" #k "*512\n"
#define I_PKMOV(k) "v_pk_mov_b32 v[8+2*" #k ":9+2*" #k "], v[28:29], v[30:31]\n"
#define I_FMAS(k) "v_fma_f32 v" #k ", v" #k ", s26, v31\n"
#define I_MIX(k) "v_pk_fma_f32 v[8+2*" #k ":9+2*" #k "], v[8+2*" #k ":9+2*" #k "], v[28:29], v[30:31]\nds_read_b128 v[32+4*" #k ":35+4*" #k "], v27 offset:" #k "*1024\n"

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23", \
             "v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50", \
             "v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","vcc","s20","s21","s22","s23","s24","s25","s26","s27","memory"

#define TEST(ID, INSTR, WAIT)                                                                                       \
    if (id == ID) {                                                                                                 \
        for (int it = 0; it < iters; ++it)                                                                          \
            asm volatile(R8(INSTR) R8(INSTR) R8(INSTR) R8(INSTR) R8(INSTR) R8(INSTR) R8(INSTR) R8(INSTR) WAIT ::: CLOB); \
    }

__global__ __launch_bounds__(1024) void k_issue(int id, int iters, long long* ticks) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.f;
    __syncthreads();
    // initialise the operand registers once
    asm volatile("v_mov_b32 v30, 1.0\nv_mov_b32 v31, 0.5\nv_mov_b32 v28, 1.0\nv_mov_b32 v29, 1.0\n"
                 "v_mbcnt_lo_u32_b32 v26, -1, 0\nv_mbcnt_hi_u32_b32 v26, -1, v26\nv_lshlrev_b32 v27, 4, v26\nv_lshlrev_b32 v25, 2, v26\nv_lshlrev_b32 v24, 3, v26\ns_mov_b64 s[22:23], 0x55555555\ns_mov_b32 s26, 1.0\nv_xor_b32 v26, 16, v26\nv_lshlrev_b32 v26, 2, v26\n"
                 "v_mov_b32 v0, 1.0\nv_mov_b32 v1, 1.0\nv_mov_b32 v2, 1.0\nv_mov_b32 v3, 1.0\nv_mov_b32 v4, 1.0\nv_mov_b32 v5, 1.0\nv_mov_b32 v6, 1.0\nv_mov_b32 v7, 1.0\n"
                 "v_mov_b32 v8, 0\nv_mov_b32 v9, 0\nv_mov_b32 v10, 0\nv_mov_b32 v11, 0\nv_mov_b32 v12, 0\nv_mov_b32 v13, 0\nv_mov_b32 v14, 0\nv_mov_b32 v15, 0\n"
                 "v_mov_b32 v16, 0\nv_mov_b32 v17, 0\nv_mov_b32 v18, 0\nv_mov_b32 v19, 0\nv_mov_b32 v20, 0\nv_mov_b32 v21, 0\nv_mov_b32 v22, 0\nv_mov_b32 v23, 0\n"
                 ::: CLOB);
    const long long t0 = __builtin_amdgcn_s_memtime();
    TEST(0, I_FMA, "")
    TEST(1, I_ADD, "")
    TEST(2, I_MUL, "")
    TEST(3, I_MOV, "")
    TEST(4, I_CND, "")
    TEST(5, I_CMP, "")
    TEST(6, I_RSQ, "")
    TEST(7, I_SQRT, "")
    TEST(8, I_RCP, "")
    TEST(9, I_PKFMA, "")
    TEST(10, I_PKADD, "")
    TEST(11, I_PKMUL, "")
    TEST(12, I_FMA64, "")
    TEST(13, I_ADD64, "")
    TEST(14, I_MUL64, "")
    TEST(15, I_RSQ64, "")
    TEST(16, I_SQRT64, "")
    TEST(17, I_CVT64, "")
    TEST(18, I_DPPMOV, "")
    TEST(19, I_DPPADD, "")
    TEST(20, I_DPPWSHR, "")
    TEST(21, I_BPERM, "s_waitcnt lgkmcnt(0)\n")
    TEST(22, I_SWZ, "s_waitcnt lgkmcnt(0)\n")
    TEST(23, I_PL32, "")
    TEST(24, I_PL16, "")
    TEST(25, I_DSR128, "s_waitcnt lgkmcnt(0)\n")
    TEST(26, I_DSR64, "s_waitcnt lgkmcnt(0)\n")
    TEST(27, I_DSR32, "s_waitcnt lgkmcnt(0)\n")
    TEST(28, I_DSW128, "s_waitcnt lgkmcnt(0)\n")
    TEST(29, I_DSW32, "s_waitcnt lgkmcnt(0)\n")
    TEST(30, I_SALU, "")
    TEST(31, I_MIX, "s_waitcnt lgkmcnt(0)\n")
    TEST(32, I_CNDS, "")
    TEST(33, I_CMPS, "")
    TEST(34, I_CMPCND, "")
    TEST(35, I_MAX, "")
    TEST(36, I_MIN3, "")
    TEST(37, I_MULLEG, "")
    TEST(38, I_DSR32C, "s_waitcnt lgkmcnt(0)\n")
    TEST(39, I_DSW32C, "s_waitcnt lgkmcnt(0)\n")
    TEST(40, I_DSR64C, "s_waitcnt lgkmcnt(0)\n")
    TEST(41, I_DSW64C, "s_waitcnt lgkmcnt(0)\n")
    TEST(42, I_PKMOV, "")
    TEST(43, I_FMAS, "")
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) ticks[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static const char* names[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_mov_b32", "v_cndmask_b32", "v_cmp_le_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_f32",
                              "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_rsq_f64", "v_sqrt_f64", "v_cvt_f64_f32",
                              "v_mov_b32 dpp row_shr:1", "v_add_f32 dpp row_shr:1", "v_mov_b32 dpp wave_shr:1", "ds_bpermute_b32", "ds_swizzle_b32",
                              "v_permlane32_swap", "v_permlane16_swap", "ds_read_b128", "ds_read_b64", "ds_read_b32", "ds_write_b128", "ds_write_b32", "s_add_u32",
                              "v_pk_fma_f32 + ds_read_b128 (pairs)", "v_cndmask_b32 e64 (sgpr pair mask)", "v_cmp_le_f32 e64 (sgpr pair dst)",
                              "v_cmp vcc + v_cndmask vcc (pairs)", "v_max_f32", "v_min3_f32", "v_mul_legacy_f32", "ds_read_b32 conflict-free",
                              "ds_write_b32 conflict-free", "ds_read_b64 conflict-free", "ds_write_b64 conflict-free", "v_pk_mov_b32", "v_fma_f32 with sgpr operand"};

int main(int argc, char** argv) {
    long long* d;
    CK(hipMalloc(&d, 8 * 256 * 16 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    printf("# cycles (s_memtime ticks) per instruction: per WAVE at W waves per SIMD, and per SIMD (= per wave / W)\n");
    printf("%-38s %9s %9s %9s %9s | %9s %9s %9s | wall-clock Ginstr/s/SIMD at W=2, W=8\n", "instruction", "W=1", "W=2", "W=4", "W=8", "SIMD W=2", "SIMD W=4", "SIMD W=8");
    for (int id = (argc > 1 ? atoi(argv[1]) : 0); id < 44; ++id) {
        double per_wave[4] = {0, 0, 0, 0};
        double wall2 = 0, wall8 = 0;
        int wi = 0;
        for (int W : {1, 2, 4, 8}) {
            // W waves per SIMD = 4 W waves per CU: blocks of 256 threads (4 waves, one per SIMD), W blocks per CU
            const int blocks = 256 * W;
            hipLaunchKernelGGL(k_issue, dim3(blocks), dim3(256), 16384, 0, id, 10, d);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_issue, dim3(blocks), dim3(256), 16384, 0, id, iters, d);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<long long> h(blocks * 4);
            CK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost));
            double s = 0;
            for (long long v : h) s += (double)v;
            const double ninstr = (double)iters * 64 * ((id == 31 || id == 34) ? 2 : 1);
            per_wave[wi++] = s / h.size() / ninstr;
            if (W == 2) wall2 = ninstr * W / (ms * 1e-3) * 1e-9;
            if (W == 8) wall8 = ninstr * W / (ms * 1e-3) * 1e-9;
        }
        printf("%-38s %9.2f %9.2f %9.2f %9.2f | %9.2f %9.2f %9.2f | %6.3f %6.3f\n", names[id], per_wave[0], per_wave[1], per_wave[2], per_wave[3],
               per_wave[1] / 2, per_wave[2] / 4, per_wave[3] / 8, wall2, wall8);
        fflush(stdout);
    }
    return 0;
}
