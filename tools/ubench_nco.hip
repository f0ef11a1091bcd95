// ubench_nco.hip -- issue cost of the NCO phase step (sig_source.c:47-53 restated branch-free) in the forms a lone wave can
// run it, with and without the LDS hand-over.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define STEP_A "v_add_f32 %0, %1, %0\n\tv_sub_f32 v40, %0, %2\n\tv_cmp_gt_f32_e64 vcc, |%0|, %3\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, v40, vcc\n\t"
#define STEP_B "v_add_f32 %0, %1, %0\n\tv_cmp_gt_f32_e64 vcc, |%0|, %3\n\tv_sub_f32 v40, %0, %2\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, v40, vcc\n\t"
#define STEP_C "v_add_f32 %0, %1, %0\n\tv_cmpx_gt_f32_e64 exec, |%0|, %3\n\tv_sub_f32 %0, %0, %2\n\ts_mov_b64 exec, -1\n\t"
#define STEP_D "v_add_f32 %0, %1, %0\n\tv_sub_f32 v40, %3, %0\n\tv_sub_f32 v41, %0, %3\n\tv_ashrrev_i32 v40, 31, v40\n\tv_bfi_b32 %0, v40, v41, %0\n\t"
// positive-step domain, 4-byte encodings: 2pi < q
#define STEP_E "v_add_f32 %0, %1, %0\n\tv_cmp_lt_f32 vcc, %3, %0\n\tv_subrev_f32 v40, %3, %0\n\ts_nop 0\n\tv_cndmask_b32 %0, %0, v40, vcc\n\t"
#define STEP_F "v_add_f32 %0, %1, %0\n\tv_cmpx_lt_f32 exec, %3, %0\n\tv_subrev_f32 %0, %3, %0\n\ts_mov_b64 exec, -1\n\t"

// no SGPR mask at all: ind = clamp(BIG * (q - 2 pi)) is exactly 0 or 1 (one ulp of q beyond 2 pi times 2^24 is >= 8), the
// wrap is fma(ind, -w, q): one rounding of q - w when ind = 1, q itself when ind = 0.  %4 = BIG with the step's sign, %5 = -w
#define STEP_G "v_add_f32 %0, %1, %0\n\tv_fma_f32 v40, %0, %4, %6 clamp\n\tv_fma_f32 %0, v40, %5, %0\n\t"
#define STEP_H "v_add_f32 %0, %2, %0\n\tv_fma_f32 v40, %0, %5, %7 clamp\n\tv_fma_f32 %0, v40, %6, %0\n\t"  // STEP_G, mode 17's operand numbers

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *stamps, int reps, float *big) {
    __shared__ float lds[64 * 68];
    float p = threadIdx.x * 0.01f, step = 0.3f + threadIdx.x * 1e-3f, w = 6.2831855f;
    const float two_pi = 6.2831855f;
    const float bigs = 16777216.0f, negw = -w, cbig = -6.2831855f * 16777216.0f;
    unsigned addr = threadIdx.x * 68 * 4;
    unsigned goff = threadIdx.x * 524288u;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; r++) {
        if (MODE == 0) asm volatile(".rept 256\n\t" STEP_A ".endr" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi) : "v40", "v41", "vcc");
        if (MODE == 1) asm volatile(".rept 256\n\t" STEP_B ".endr" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi) : "v40", "v41", "vcc");
        if (MODE == 2) asm volatile(".rept 256\n\t" STEP_C ".endr" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi) : "v40", "v41", "vcc");
        if (MODE == 3) asm volatile(".rept 256\n\t" STEP_D ".endr" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi) : "v40", "v41", "vcc");
        if (MODE == 4) asm volatile(".rept 256\n\t" STEP_E ".endr" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi) : "v40", "v41", "vcc");
        if (MODE == 5) asm volatile(".rept 256\n\t" STEP_F ".endr" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi) : "v40", "v41", "vcc");
        // with the block hand-over: one ds_write_b128 per 4 samples (values: the running phase four times, cost is the same)
        if (MODE == 6) asm volatile(".rept 64\n\t" STEP_A STEP_A STEP_A STEP_A "v_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tds_write_b128 %4, v[42:45]\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
        if (MODE == 7) asm volatile(".rept 64\n\t" STEP_A STEP_A STEP_A STEP_A "ds_write_b128 %4, v[42:45]\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
        if (MODE == 8) asm volatile(".rept 64\n\t" STEP_C STEP_C STEP_C STEP_C "ds_write_b128 %4, v[42:45]\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
        if (MODE == 9) asm volatile(".rept 64\n\t" STEP_C STEP_C STEP_C STEP_C "ds_write_b32 %4, v42\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
        if (MODE == 10) asm volatile(".rept 64\n\t" STEP_B STEP_B STEP_B STEP_B "ds_write_b128 %4, v[42:45]\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
        if (MODE == 11) asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffff\n\t.rept 64\n\t" STEP_B STEP_B STEP_B STEP_B "ds_write_b128 %4, v[42:45]\n\t.endr\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b64 exec, s[20:21]" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc", "s20", "s21");
        if (MODE == 12) asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffff\n\t.rept 256\n\t" STEP_B ".endr\n\ts_mov_b64 exec, s[20:21]" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi) : "v40", "v41", "vcc", "s20", "s21");
        if (MODE == 13) asm volatile(".rept 256\n\t" STEP_G ".endr" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(bigs), "v"(negw), "s"(cbig) : "v40", "v41", "vcc");
        if (MODE == 14) asm volatile(".rept 64\n\t" STEP_G STEP_G STEP_G STEP_G "ds_write_b128 %7, v[42:45]\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(bigs), "v"(negw), "s"(cbig), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
        if (MODE == 15) asm volatile(".rept 64\n\t" STEP_G STEP_G STEP_G STEP_G "ds_write_b64 %7, v[42:43]\n\tds_write_b64 %7, v[44:45] offset:8\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(bigs), "v"(negw), "s"(cbig), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
        if (MODE == 16) asm volatile(".rept 32\n\t" STEP_G STEP_G STEP_G STEP_G STEP_G STEP_G STEP_G STEP_G "ds_write_b128 %7, v[42:45]\n\tds_write_b128 %7, v[46:49] offset:16\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(p) : "v"(step), "v"(w), "s"(two_pi), "v"(bigs), "v"(negw), "s"(cbig), "v"(addr) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "vcc");
        // straight to global memory, lane = channel row (512 KB apart): no LDS, no store wave
        if (MODE == 17) asm volatile(".rept 64\n\t" STEP_H STEP_H STEP_H STEP_H "global_store_dwordx4 %1, v[42:45], %8\n\tv_add_u32 %1, 16, %1\n\t.endr\n\t" : "+v"(p), "+v"(goff) : "v"(step), "v"(w), "s"(two_pi), "v"(bigs), "v"(negw), "s"(cbig), "s"(big) : "v40", "v41", "v42", "v43", "v44", "v45", "vcc");
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = p + lds[threadIdx.x];
    if (threadIdx.x == 0) {
        stamps[0] = c1 - c0;
        stamps[1] = r1 - r0;
    }
}

int main() {
    float *out;
    unsigned long long *st, h[2];
    hipMalloc(&out, 4096);
    hipMalloc(&st, 64);
    float *big;
    hipMalloc(&big, 64u * 524288u + (1u << 20));
    const int reps = 64;
    const char *names[] = {"add, sub, cmp |q|, s_nop 1, cndmask (compiler's order)", "add, cmp |q|, sub, s_nop 0, cndmask",
                           "add, cmpx |q| -> exec, masked sub, s_mov exec", "add, sub, sub, ashr, bfi (no SGPR)",
                           "positive-step domain, 4-byte: add, cmp, subrev, s_nop 0, cndmask", "positive-step domain, 4-byte: add, cmpx, masked subrev, s_mov exec",
                           "compiler's order + 4 v_mov + ds_write_b128 per 4 samples", "compiler's order + ds_write_b128 per 4 samples",
                           "cmpx form + ds_write_b128 per 4 samples", "cmpx form + ds_write_b32 per 4 samples",
                           "hand order + ds_write_b128 per 4 samples, 64 lanes", "hand order + ds_write_b128 per 4 samples, 16 lanes", "hand order, 16 lanes, no write",
                           "add, fma clamp, fma (no mask)", "no-mask form + ds_write_b128 per 4 samples", "no-mask form + 2 ds_write_b64 per 4 samples",
                           "no-mask form + 2 ds_write_b128 per 8 samples", "no-mask form + global_store_dwordx4 per 4 samples, lane = row"};
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, st, reps, big); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, st, reps, big); hipDeviceSynchronize(); \
    hipMemcpy(h, st, 16, hipMemcpyDeviceToHost); printf("mode %d: %6.2f cycles per sample at %4.0f MHz  (%s)\n", M, (double) h[0] / (reps * 256.0), (double) h[0] / (double) h[1] * 100.0, names[M]);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17)
    return 0;
}
