// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the FP64 / helper instructions the
// traversal uses. Each kernel runs a long run of independent instructions of one kind; one wave per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o tools/microbench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

#define KERNEL(NAME, ASM)                                                                                  \
    __global__ void __launch_bounds__(64) NAME(double *out, int iters, uint64_t *cyc) {                     \
        double a = out[threadIdx.x], b = out[threadIdx.x + 64], c = out[threadIdx.x + 128], d = 0;           \
        float fa = (float)a, fb = (float)b, fd = 0;                                                          \
        int ia = (int)a, id = 0;                                                                             \
        uint64_t t0 = __builtin_amdgcn_s_memtime();                                                          \
        uint64_t r0 = clock64();                                                                             \
        for (int i = 0; i < iters; i++) { REP64(asm volatile(ASM : "+v"(d), "+v"(fd), "+v"(id) : "v"(a), "v"(b), "v"(c), "v"(fa), "v"(fb), "v"(ia) : "vcc", "s20", "scc", "v100", "v101", "v102", "v103", "v104", "v105");) } \
        uint64_t r1 = clock64();                                                                             \
        uint64_t t1 = __builtin_amdgcn_s_memtime();                                                          \
        out[threadIdx.x] = d + fd + id;                                                                      \
        if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }                     \
    }

// operands: %0 d (f64), %1 fd (f32), %2 id (i32), %3 a, %4 b, %5 c (f64), %6 fa, %7 fb (f32), %8 ia (i32)
KERNEL(k_fma64, "v_fma_f64 %0, %3, %4, %5")
KERNEL(k_mul64, "v_mul_f64 %0, %3, %4")
KERNEL(k_fma64_x2, "v_fma_f64 %0, %3, %4, %5\n v_fma_f64 v[100:101], %3, %4, %5")
KERNEL(k_fma64_x4, "v_fma_f64 %0, %3, %4, %5\n v_fma_f64 v[100:101], %3, %4, %5\n v_fma_f64 v[102:103], %3, %4, %5\n v_fma_f64 v[104:105], %3, %4, %5")
KERNEL(k_mix, "v_fma_f64 %0, %3, %4, %5\n s_add_u32 s20, s21, s22\n v_fma_f64 v[100:101], %3, %4, %5\n s_add_u32 s20, s21, s22")
KERNEL(k_add64, "v_add_f64 %0, %3, %4")
KERNEL(k_min64, "v_min_f64 %0, %3, %4")
KERNEL(k_max64, "v_max_f64 %0, %3, %4")
KERNEL(k_cmp64, "v_cmp_gt_f64 vcc, %3, %4")
KERNEL(k_cvt6432, "v_cvt_f64_f32 %0, %6")
KERNEL(k_cvt3264, "v_cvt_f32_f64 %1, %3")
KERNEL(k_rcp64, "v_rcp_f64 %0, %3")
KERNEL(k_fma32, "v_fma_f32 %1, %6, %7, %6")
KERNEL(k_pkfma32, "v_pk_fma_f32 %0, %3, %4, %5")
KERNEL(k_min32, "v_min_f32 %1, %6, %7")
KERNEL(k_addu32, "v_add_u32 %2, %8, %8")
KERNEL(k_mbcnt, "v_mbcnt_lo_u32_b32 %2, %8, %8")
KERNEL(k_readlane, "v_readlane_b32 s20, %8, 3")
KERNEL(k_dppmov, "v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_salu, "s_add_u32 s20, s21, s22")
KERNEL(k_cndmask, "v_cndmask_b32 %2, %8, %8, vcc")
KERNEL(k_lshladd64, "v_lshl_add_u64 %0, %3, 3, %4")

typedef void (*kern_t)(double *, int, uint64_t *);

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    double *out; uint64_t *cyc;
    CHECK(hipMalloc(&out, 192 * 8)); CHECK(hipMemset(out, 0, 192 * 8));
    CHECK(hipMalloc(&cyc, 16));
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f64", k_fma64}, {"2 indep fma64", k_fma64_x2}, {"4 indep fma64", k_fma64_x4}, {"fma,salu,fma,salu", k_mix}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_min_f64", k_min64}, {"v_max_f64", k_max64},
        {"v_cmp_gt_f64", k_cmp64}, {"v_cvt_f64_f32", k_cvt6432}, {"v_cvt_f32_f64", k_cvt3264}, {"v_rcp_f64", k_rcp64},
        {"v_fma_f32", k_fma32}, {"v_pk_fma_f32", k_pkfma32}, {"v_min_f32", k_min32}, {"v_add_u32", k_addu32}, {"v_mbcnt_lo", k_mbcnt},
        {"v_readlane_b32", k_readlane}, {"v_mov_b32_dpp", k_dppmov}, {"s_add_u32", k_salu}, {"v_cndmask_b32", k_cndmask}, {"v_lshl_add_u64", k_lshladd64}};
    const int iters = 2000;
    for (auto &e : ks) {
        for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd++) {
            // one CU is enough: blocks of 64 threads, 4 * waves_per_simd of them per CU; launch for all CUs so clocks ramp
            const int blocks = 256 * 4 * waves_per_simd;
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            e.k<<<blocks, 64>>>(out, 10, cyc);
            CHECK(hipEventRecord(e0));
            e.k<<<blocks, 64>>>(out, iters, cyc);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t h[2]; CHECK(hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost));
            const double n = (double)iters * 64;
            printf("%-16s waves/SIMD %d: %.2f clock64 ticks/instr/wave, %.2f memtime ticks, %.2f ns/instr/wave (wall %.3f ms)\n", e.name, waves_per_simd,
                   h[1] / n, h[0] / n, ms * 1e6 / n, ms);
        }
    }
    return 0;
}
