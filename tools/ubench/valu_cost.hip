// Micro-benchmark: issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU
// instructions the composite kernels are made of, on gfx950.  s_memtime ticks are shader cycles.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_cost valu_cost.hip ; run: ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(X) X X X X X X X X X X X X X X X X

#define DEFINE_KERNEL(NAME, ASM)                                                          \
    __global__ void NAME(unsigned long long *out, float seed) {                           \
        float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, \
              v7 = seed + 7;                                                              \
        int s0 = 0;                                                                       \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                             \
        for (int i = 0; i < 256; i++) {                                                   \
            asm volatile(REP16(ASM) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+s"(s0) : : "s10", "s11", "vcc"); \
        }                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                  \
        if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + s0 == 123.456f) out[0] = 0;           \
    }

// 8 instructions per block, each on its own register (no dependencies inside a group of 8)
DEFINE_KERNEL(k_fma, "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
DEFINE_KERNEL(k_mul, "v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n")
DEFINE_KERNEL(k_add, "v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7\n")
DEFINE_KERNEL(k_min, "v_min_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n v_min_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_min_f32 %6, %6, %7\n v_min_f32 %7, %7, %0\n")
DEFINE_KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
DEFINE_KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
DEFINE_KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
DEFINE_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
DEFINE_KERNEL(k_cmp, "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n")
DEFINE_KERNEL(k_readlane, "v_readlane_b32 %8, %0, 5\n v_readlane_b32 %8, %1, 6\n v_readlane_b32 %8, %2, 7\n v_readlane_b32 %8, %3, 8\n v_readlane_b32 %8, %4, 9\n v_readlane_b32 %8, %5, 10\n v_readlane_b32 %8, %6, 11\n v_readlane_b32 %8, %7, 12\n")
DEFINE_KERNEL(k_fma_sgpr, "v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n")
DEFINE_KERNEL(k_sub_sgpr, "v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n")
DEFINE_KERNEL(k_dpp_add, "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
DEFINE_KERNEL(k_permlane32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n")
DEFINE_KERNEL(k_iadd, "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %4\n v_add_u32 %4, %4, %5\n v_add_u32 %5, %5, %6\n v_add_u32 %6, %6, %7\n v_add_u32 %7, %7, %0\n")

DEFINE_KERNEL(k_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]\n v_cndmask_b32_e64 %1, %1, %2, s[10:11]\n v_cndmask_b32_e64 %2, %2, %3, s[10:11]\n v_cndmask_b32_e64 %3, %3, %4, s[10:11]\n v_cndmask_b32_e64 %4, %4, %5, s[10:11]\n v_cndmask_b32_e64 %5, %5, %6, s[10:11]\n v_cndmask_b32_e64 %6, %6, %7, s[10:11]\n v_cndmask_b32_e64 %7, %7, %0, s[10:11]\n")
DEFINE_KERNEL(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc\n v_cmp_lt_f32 vcc, %5, %4\n v_cndmask_b32 %7, %7, %6, vcc\n")
DEFINE_KERNEL(k_cnd_indep, "v_cndmask_b32 %0, %0, %0, vcc\n v_cndmask_b32 %1, %1, %1, vcc\n v_cndmask_b32 %2, %2, %2, vcc\n v_cndmask_b32 %3, %3, %3, vcc\n v_cndmask_b32 %4, %4, %4, vcc\n v_cndmask_b32 %5, %5, %5, vcc\n v_cndmask_b32 %6, %6, %6, vcc\n v_cndmask_b32 %7, %7, %7, vcc\n")
DEFINE_KERNEL(k_min_chain, "v_min_f32 %0, %0, %1\n v_min_f32 %0, %0, %2\n v_min_f32 %0, %0, %3\n v_min_f32 %0, %0, %4\n v_min_f32 %0, %0, %5\n v_min_f32 %0, %0, %6\n v_min_f32 %0, %0, %7\n v_min_f32 %0, %0, %1\n")
DEFINE_KERNEL(k_fma_chain, "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n")

// packed: operate on register pairs
#define DEFINE_PK(NAME, ASM)                                                              \
    __global__ void NAME(unsigned long long *out, float seed) {                           \
        typedef float f2 __attribute__((ext_vector_type(2)));                             \
        f2 v0 = {seed, seed + 1}, v1 = {seed + 2, seed}, v2 = {seed + 3, seed}, v3 = {seed + 4, seed};  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                             \
        for (int i = 0; i < 256; i++) {                                                   \
            asm volatile(REP16(ASM) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));            \
        }                                                                                 \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                             \
        if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                  \
        if (v0.x + v1.x + v2.x + v3.x == 123.456f) out[0] = 0;                            \
    }
DEFINE_PK(k_pk_fma, "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n")
DEFINE_PK(k_pk_mul, "v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n")

typedef void (*kern_t)(unsigned long long *, float);

int main() {
    unsigned long long *d, h[1024];
    hipMalloc(&d, sizeof(h));
    struct { const char *name; kern_t k; int per_group; } tests[] = {
        {"v_fma_f32", k_fma, 8}, {"v_mul_f32", k_mul, 8}, {"v_add_f32", k_add, 8}, {"v_min_f32", k_min, 8},
        {"v_exp_f32", k_exp, 8}, {"v_rcp_f32", k_rcp, 8}, {"v_mov_b32", k_mov, 8}, {"v_cndmask_b32", k_cndmask, 8},
        {"v_cmp_lt_f32", k_cmp, 8}, {"v_readlane_b32", k_readlane, 8}, {"v_fma_f32 (sgpr src)", k_fma_sgpr, 8},
        {"v_sub_f32 (sgpr src)", k_sub_sgpr, 8}, {"v_add_f32_dpp row_shr", k_dpp_add, 8},
        {"v_permlane32_swap", k_permlane32, 8}, {"v_add_u32", k_iadd, 8}, {"v_cndmask_e64 (sgpr mask)", k_cndmask_e64, 8}, {"v_cmp+v_cndmask pairs", k_cmp_cnd, 8},
        {"v_cndmask same-reg", k_cnd_indep, 8}, {"v_min_f32 dependent chain", k_min_chain, 8}, {"v_fma_f32 dependent chain", k_fma_chain, 8}, {"v_pk_fma_f32", k_pk_fma, 4},
        {"v_pk_mul_f32", k_pk_mul, 4}};
    for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 4) {
        printf("--- %d wave(s) per SIMD (blocks of %d threads, 256 blocks)\n", waves_per_simd, 256 * waves_per_simd);
        for (auto &t : tests) {
            for (int rep = 0; rep < 2; rep++) {
                hipLaunchKernelGGL(t.k, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 1.0f);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d, sizeof(unsigned long long) * 256, hipMemcpyDeviceToHost);
            double avg = 0;
            for (int i = 0; i < 256; i++) avg += (double)h[i];
            avg /= 256;
            const double n = 256.0 * 16 * t.per_group;
            printf("%-26s %7.2f cycles / instruction / wave   (issue share with %d waves: %.2f)\n", t.name, avg / n,
                   waves_per_simd, avg / n / waves_per_simd);
        }
    }
    return 0;
}
