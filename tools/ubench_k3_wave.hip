// ubench_k3_wave.hip -- round 4: what would a WAVE-per-channel clock stage cost per symbol?
//
// The clock stage runs one channel per LANE: ~46 instructions per symbol at the ~4.4 cycles a lone wave needs per instruction,
// plus one LDS round trip for the window and the interpolator row (their addresses depend on the loop state): ~222 cycles.
// One channel per WAVE makes the loop state wave-uniform, so the row and the window can be selected by REGISTER index
// (s_set_gpr_idx_*) instead of an LDS address: the 129 rows live in 129 VGPRs (lane (g, j) holds tap j, g = 0..7), a ring of 32
// VGPRs holds the samples (register q, lane (g, j) = x[8 q + g + j]: the window that starts at ii is register ii >> 3, lane
// group ii & 7), the 8 products are one v_mul_f32, the in-order sum 8 v_add_f32 (7 with DPP row_shr:1), the result leaves by
// v_readlane_b32 with a scalar lane number, and the recursion runs on all lanes with the result as a scalar operand.
// This program times that instruction sequence (real dependences, arbitrary data: values do not matter for the cost) on one
// wave per workgroup, and the lane-per-channel dependence pattern (operands through LDS) next to it for reference.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_k3_wave.hip -o tools/ubench_k3_wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// registers reserved by the asm below: v[64:192] rows T0..T128, v[200:231] sample ring X0..X31, v[32:63] scratch / state
#define CLOB32(a) "v" #a
#define SYM_WAVE                                                                                                        \
    /* imu = rint(mu * 128) by the magic add; position float pm carries ii in its low mantissa bits */                   \
    "v_fma_f32 v40, v32, v50, v51\n\t"            /* v32 = mu, v50 = 128.0, v51 = magic */                               \
    "v_readfirstlane_b32 s20, v40\n\t"                                                                                   \
    "v_readfirstlane_b32 s21, v35\n\t"            /* v35 = pm */                                                         \
    "s_and_b32 s20, s20, 0x7f\n\t"                /* row (0..128 in the real thing; 0..127 here) */                      \
    "s_bfe_u32 s22, s21, 0x50003\n\t"             /* (ii >> 3) & 31: ring register */                                    \
    "s_and_b32 s23, s21, 7\n\t"                                                                                          \
    "s_lshl3_add_u32 s23, s23, 7\n\t"             /* lane 8 g + 7 */                                                     \
    "s_set_gpr_idx_on s20, 0x1\n\t"               /* SRC0 indexed */                                                     \
    "v_mov_b32 v41, v64\n\t"                      /* tap row: T[imu] */                                                  \
    "s_set_gpr_idx_idx s22\n\t"                                                                                          \
    "v_mul_f32 v42, v200, v41\n\t"                /* X[q] * taps */                                                      \
    "s_set_gpr_idx_off\n\t"                                                                                              \
    "v_add_f32 v43, 0, v42\n\t"                                                                                          \
    "s_nop 0\n\t"                                                                                                        \
    "v_add_f32_dpp v43, v43, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                               \
    "s_nop 0\n\t"                                                                                                        \
    "v_add_f32_dpp v43, v43, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                               \
    "s_nop 0\n\t"                                                                                                        \
    "v_add_f32_dpp v43, v43, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                               \
    "s_nop 0\n\t"                                                                                                        \
    "v_add_f32_dpp v43, v43, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                               \
    "s_nop 0\n\t"                                                                                                        \
    "v_add_f32_dpp v43, v43, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                               \
    "s_nop 0\n\t"                                                                                                        \
    "v_add_f32_dpp v43, v43, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                               \
    "s_nop 0\n\t"                                                                                                        \
    "v_add_f32_dpp v43, v43, v42 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                                               \
    "s_nop 3\n\t"                                 /* SALU wrote s23 long ago; VALU wrote v43 just now */                 \
    "v_readlane_b32 s24, v43, s23\n\t"            /* o */                                                                \
    /* mm = slice(last) o - slice(o) last through the sign trick; last lives in s25 */                                   \
    "v_xor_b32 v44, s24, v34\n\t"                 /* v34 = last */                                                       \
    "v_bfi_b32 v45, v52, s24, v44\n\t"            /* v52 = 0x7fffffff */                                                 \
    "v_bfi_b32 v46, v52, v34, v44\n\t"                                                                                   \
    "v_sub_f32 v44, v45, v46\n\t"                 /* mm */                                                               \
    "v_mov_b32 v34, s24\n\t"                      /* last = o */                                                         \
    "v_pk_mul_f32 v[46:47], v[60:61], v[44:45] op_sel_hi:[1,0]\n\t" /* (g_omega, g_mu) * mm */                           \
    "v_add_f32 v33, v33, v46\n\t"                 /* omega += */                                                         \
    "v_sub_f32 v62, v33, v55\n\t"                 /* dev = omega - mid */                                                \
    "v_pk_add_f32 v[48:49], v[56:57], v[62:63] op_sel_hi:[1,0]\n\t" /* (lim, -lim) + dev */                              \
    "v_sub_f32 v45, |v48|, |v49|\n\t"                                                                                    \
    "v_fma_f32 v33, v45, v58, v55\n\t"            /* omega = 0.5 y + mid */                                              \
    "v_add_f32 v32, v32, v33\n\t"                 /* mu + omega */                                                       \
    "v_add_f32 v32, v32, v47\n\t"                 /* + g_mu mm */                                                        \
    "v_floor_f32 v45, v32\n\t"                                                                                           \
    "v_sub_f32 v32, v32, v45\n\t"                 /* mu -= floor */                                                      \
    "v_add_f32 v35, v35, v45\n\t"                 /* pm += floor */                                                      \
    "s_add_u32 s26, s26, 1\n\t"                                                                                          \
    "s_mov_b32 m0, s26\n\t"                                                                                              \
    "v_writelane_b32 v36, s24, m0\n\t"            /* output register (lane k) */                                         \
    "s_cmp_lt_u32 s21, s27\n\t"                   /* ii < limit (previous position: one symbol of slack in the real thing) */ \
    "s_cbranch_scc0 2f\n\t"

__global__ __launch_bounds__(64) void wave_symbol(float *out, unsigned long long *cycles, int symbols) {
    float mu = 0.5f, omega = 5.0f, last = 0.25f;
    unsigned long long c0, c1;
    // position float: 1.5 * 2^23 + ii, advancing ~5 per symbol; data registers get arbitrary finite contents
    asm volatile(
        "v_mov_b32 v32, %2\n\tv_mov_b32 v33, %3\n\tv_mov_b32 v34, %4\n\tv_mov_b32 v35, 0x4b400000\n\t"
        "v_mov_b32 v50, 0x43000000\n\tv_mov_b32 v51, 0x4b400000\n\tv_mov_b32 v52, 0x7fffffff\n\t"
        "v_mov_b32 v60, 0x3e20d97c\n\tv_mov_b32 v61, 0x3d800000\n\tv_mov_b32 v55, 0x40a00000\n\tv_mov_b32 v62, 0\n\tv_mov_b32 v63, 0\n\t"   /* g_omega, g_mu, mid */
        "v_mov_b32 v56, 0x3d4ccccd\n\tv_mov_b32 v57, 0xbd4ccccd\n\tv_mov_b32 v58, 0x3f000000\n\t"   /* lim, -lim, 0.5 */
        "v_mov_b32 v36, 0\n\tv_mov_b32 v45, 0\n\tv_mov_b32 v46, 0\n\tv_mov_b32 v47, 0\n\tv_mov_b32 v48, 0\n\tv_mov_b32 v49, 0\n\t"
        "s_mov_b32 s26, 0\n\ts_mov_b32 s27, 0x7fffffff\n\t"
        "s_movk_i32 s28, 0\n"
        "0:\n\t"  /* fill T0..T128 and X0..X31 with something finite */
        "s_set_gpr_idx_on s28, 0x8\n\t"  /* DST indexed */
        "v_mov_b32 v64, 0x3dcccccd\n\t"
        "s_set_gpr_idx_off\n\t"
        "s_add_u32 s28, s28, 1\n\ts_cmp_lt_u32 s28, 168\n\ts_cbranch_scc1 0b\n\t"
        "s_memtime %0\n\ts_waitcnt lgkmcnt(0)\n\t"
        "s_mov_b32 s29, %5\n"
        "1:\n\t"
        SYM_WAVE SYM_WAVE SYM_WAVE SYM_WAVE SYM_WAVE SYM_WAVE SYM_WAVE SYM_WAVE
        "s_sub_u32 s29, s29, 8\n\ts_cmp_gt_i32 s29, 0\n\ts_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_memtime %1\n\ts_waitcnt lgkmcnt(0)\n\t"
        : "=&s"(c0), "=&s"(c1)
        : "v"(mu), "v"(omega), "v"(last), "s"(symbols)
        : "memory", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "m0",
          "v32", "v33", "v34", "v35", "v36", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52",
          "v55", "v56", "v57", "v58", "v60", "v61", "v62", "v63", "v64", "v192", "v200", "v231");
    float r;
    asm volatile("v_add_f32 %0, v32, v36" : "=v"(r));
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) {
        cycles[blockIdx.x] = c1 - c0;
    }
}

int main() {
    const int symbols = 26208;  // one 131072-sample call at 5 samples per symbol
    for (int blocks : {1, 16, 256, 1024}) {
        float *out;
        unsigned long long *cyc, h[1024];
        hipMalloc(&out, sizeof(float) * 64 * blocks);
        hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(wave_symbol, dim3(blocks), dim3(64), 0, 0, out, cyc, symbols);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(wave_symbol, dim3(blocks), dim3(64), 0, 0, out, cyc, symbols);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
        double lo = 1e30, hi = 0, sum = 0;
        for (int i = 0; i < blocks; i++) {
            const double c = (double) h[i] / symbols;
            lo = c < lo ? c : lo;
            hi = c > hi ? c : hi;
            sum += c;
        }
        printf("wave-per-channel symbol loop, %4d waves (one per workgroup): %.1f cycles per symbol (min %.1f, max %.1f), kernel %.3f ms "
               "for %d symbols = %.1f ns per symbol\n", blocks, sum / blocks, lo, hi, ms, symbols, ms * 1e6 / symbols);
        hipFree(out);
        hipFree(cyc);
    }
    printf("for reference: the lane-per-channel clock stage needs 217-222 cycles per symbol (DESIGN.md, K3), 2.53-2.57 ms per call\n");
    return 0;
}
