// ubench_chain.hip -- latency of the dependent chains the sequential stages are made of, and the clock the chip holds
// while only a few waves are resident.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// MODE 0: 1024 dependent v_add_f32.  MODE 1: 1024 dependent v_add_f32_dpp wave_shr:1 (+ s_nop 1).
// MODE 2: 1024 dependent v_add_f32_dpp row_shr:1.  MODE 3: 256 dependent LDS round trips (ds_read_b32, address from data)
template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *stamps, int reps) {
    __shared__ int lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 7 + 3) & 1023;
    __syncthreads();
    float s = threadIdx.x * 0.001f, t = 1.0f + threadIdx.x * 1e-6f;
    int idx = threadIdx.x;
    float t2 = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pk = {s, t}, pk1 = {1.0f, 1.0f};
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; r++) {
        if (MODE == 0) {
            asm volatile(".rept 1024\n\tv_add_f32 %0, %0, %1\n\t.endr" : "+v"(s) : "v"(t));
        } else if (MODE == 1) {
            asm volatile(".rept 1024\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+v"(s) : "v"(t));
        } else if (MODE == 2) {
            asm volatile(".rept 1024\n\ts_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+v"(s) : "v"(t));
        } else if (MODE == 3) {
#pragma unroll 1
            for (int i = 0; i < 256; i++) idx = lds[idx];
        } else if (MODE == 4) {  // same chain with only lanes 0..31 enabled: does a half-empty wave issue faster?
            asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffffffff\n\t.rept 1024\n\tv_add_f32 %0, %0, %1\n\t.endr\n\ts_mov_b64 exec, s[20:21]" : "+v"(s) : "v"(t) : "s20", "s21");
        } else if (MODE == 5) {  // lanes 0..15 only
            asm volatile("s_mov_b64 s[20:21], exec\n\ts_mov_b64 exec, 0xffff\n\t.rept 1024\n\tv_add_f32 %0, %0, %1\n\t.endr\n\ts_mov_b64 exec, s[20:21]" : "+v"(s) : "v"(t) : "s20", "s21");
        } else if (MODE == 6) {  // two independent chains interleaved: issue rate of one wave
            asm volatile(".rept 512\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t.endr" : "+v"(s), "+v"(t2) : "v"(t));
        } else if (MODE == 7) {  // dependent v_pk_mul_f32
            asm volatile(".rept 1024\n\tv_pk_mul_f32 %0, %0, %1\n\t.endr" : "+v"(pk) : "v"(pk1));
        } else if (MODE == 9) {  // 8-byte encodings (VOP3 form of the same add): is a single wave fetch-limited?
            asm volatile(".rept 512\n\tv_add_f32_e64 %0, %0, %2\n\tv_add_f32_e64 %1, %1, %2\n\t.endr" : "+v"(s), "+v"(t2) : "v"(t));
        } else if (MODE == 10) {  // 16 trips of a 64-instruction loop of 4-byte adds: cost of the taken branch
            asm volatile("s_mov_b32 s20, 16\n1:\n\t.rept 32\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t.endr\n\t"
                         "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b" : "+v"(s), "+v"(t2) : "v"(t) : "s20", "scc");
        } else if (MODE == 11) {  // the same loop with 8-byte adds
            asm volatile("s_mov_b32 s20, 16\n1:\n\t.rept 32\n\tv_add_f32_e64 %0, %0, %2\n\tv_add_f32_e64 %1, %1, %2\n\t.endr\n\t"
                         "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b" : "+v"(s), "+v"(t2) : "v"(t) : "s20", "scc");
        } else if (MODE == 12) {  // s_waitcnt that never waits, between adds
            asm volatile(".rept 512\n\tv_add_f32 %0, %0, %2\n\ts_waitcnt lgkmcnt(0)\n\t.endr" : "+v"(s), "+v"(t2) : "v"(t));
        } else if (MODE == 13) {  // ds_read_b64 issue cost: one independent LDS read per add
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %2\n\tds_read_b64 v[40:41], %3\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s), "+v"(t2) : "v"(t), "v"(idx * 8) : "v40", "v41");
        } else if (MODE == 14) {
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %2\n\tds_read_b32 v40, %3\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s), "+v"(t2) : "v"(t), "v"(idx * 16) : "v40", "v41");
        } else if (MODE == 15) {
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %2\n\tds_read_b128 v[40:43], %3\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s), "+v"(t2) : "v"(t), "v"(idx * 16) : "v40", "v41", "v42", "v43");
        } else if (MODE == 16) {
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %2\n\tds_read2_b32 v[40:41], %3 offset0:1 offset1:2\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s), "+v"(t2) : "v"(t), "v"(idx * 16) : "v40", "v41");
        } else if (MODE == 17) {
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %2\n\tds_read2_b64 v[40:43], %3 offset0:1 offset1:2\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s), "+v"(t2) : "v"(t), "v"(idx * 16) : "v40", "v41", "v42", "v43");
        } else if (MODE == 18) {  // 4 adds per LDS read: does the read's cost hide behind VALU work?
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %0, %0, %2\n\tds_read_b64 v[40:41], %3\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s), "+v"(t2) : "v"(t), "v"(idx * 16) : "v40", "v41");
        } else if (MODE == 22) {  // quad-broadcast DPP add, dependent through the NON-DPP operand (the DPP source is old)
            asm volatile(".rept 256\n\tv_add_f32_dpp %0, %1, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, %1, %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, %1, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t.endr" : "+v"(s) : "v"(t));
        } else if (MODE == 23) {  // the dot product of the clock stage in quad form: pk_mul, s_nop 1, 8 broadcast adds
            asm volatile(".rept 102\n\tv_pk_mul_f32 v[40:41], %1, %2\n\ts_nop 1\n\t"
                         "v_add_f32_dpp %0, v40, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, v40, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, v40, %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, v40, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, v41, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, v41, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, v41, %0 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %0, v41, %0 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t.endr" : "+v"(s) : "v"(pk), "v"(pk1) : "v40", "v41");
        } else if (MODE == 24) {  // the same in today's form: 4 pk_mul, 8 plain adds (12 instructions)
            asm volatile(".rept 85\n\tv_pk_mul_f32 v[40:41], %1, %2\n\tv_pk_mul_f32 v[42:43], %1, %2\n\t"
                         "v_add_f32 %0, 0, v40\n\tv_add_f32 %0, v41, %0\n\tv_add_f32 %0, v42, %0\n\tv_add_f32 %0, v43, %0\n\t"
                         "v_pk_mul_f32 v[44:45], %1, %2\n\tv_pk_mul_f32 v[46:47], %1, %2\n\t"
                         "v_add_f32 %0, v44, %0\n\tv_add_f32 %0, v45, %0\n\tv_add_f32 %0, v46, %0\n\tv_add_f32 %0, v47, %0\n\t.endr"
                         : "+v"(s) : "v"(pk), "v"(pk1) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
        } else if (MODE == 19) {  // byte store per 8 adds
            asm volatile(".rept 128\n\t.rept 8\n\tv_add_f32 %0, %0, %2\n\t.endr\n\tglobal_store_byte %3, %0, %4\n\t.endr" : "+v"(s), "+v"(t2) : "v"(t), "v"(idx * 4096), "s"(out) : "memory");
        } else if (MODE == 20) {  // two interleaved in-order DPP chains (each gives the other its wait states)
            asm volatile(".rept 512\n\tv_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t.endr" : "+v"(s), "+v"(t2) : "v"(t));
        } else if (MODE == 21) {  // four interleaved chains, no nops
            asm volatile(".rept 256\n\tv_add_f32_dpp %0, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                         "v_add_f32_dpp %2, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %3, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t.endr" : "+v"(s), "+v"(t2), "+v"(pk.x), "+v"(pk.y) : "v"(t));
        } else if (MODE == 8) {  // v_cmp -> v_cndmask dependent pair (+ the add closing the chain)
            asm volatile(".rept 512\n\tv_cmp_gt_f32 vcc, 0, %0\n\tv_cndmask_b32 %0, %0, %1, vcc\n\t.endr" : "+v"(s) : "v"(t) : "vcc");
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 64 + threadIdx.x] = s + idx + t2 + pk.x + pk.y;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char *name, int blocks, int reps, int steps_per_rep) {
    float *d;
    unsigned long long *st, *h = (unsigned long long *) malloc(16 * blocks);
    hipMalloc(&d, 256 * blocks);
    hipMalloc(&st, 16 * blocks);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, st, reps);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(64), 0, 0, d, st, reps);
    hipDeviceSynchronize();
    hipMemcpy(h, st, 16 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0;
    for (int i = 0; i < blocks; i++) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
    cyc /= blocks; rt /= blocks;
    double steps = (double) reps * steps_per_rep;
    printf("%-22s waves %5d: %7.2f shader-cycles/step, %7.2f ns/step, clock %.0f MHz\n", name, blocks, cyc / steps,
           rt * 10.0 / steps, cyc / (rt * 10.0) * 1000.0);
    hipFree(d); hipFree(st); free(h);
}

int main() {
    for (int blocks : {4}) {
        run<0>("dependent v_add_f32", blocks, 200, 1024);
        run<1>("dpp wave_shr chain", blocks, 200, 1024);
        run<2>("dpp row_shr chain", blocks, 200, 1024);
        run<3>("dependent ds_read_b32", blocks, 200, 256);
        run<4>("v_add chain, 32 lanes", blocks, 200, 1024);
        run<5>("v_add chain, 16 lanes", blocks, 200, 1024);
        run<6>("2 indep v_add chains", blocks, 200, 1024);
        run<7>("dependent v_pk_mul_f32", blocks, 200, 1024);
        run<8>("v_cmp+v_cndmask chain", blocks, 200, 1024);
        run<9>("2 chains, 8-byte adds", blocks, 200, 1024);
        run<10>("64-instr loop, 4-byte", blocks, 200, 1024);
        run<11>("64-instr loop, 8-byte", blocks, 200, 1024);
        run<12>("add + idle s_waitcnt", blocks, 200, 512);
        run<22>("quad-bcast dpp add chain", blocks, 200, 1024);
        run<23>("dot8 quad form (11 instr)", blocks, 200, 102);
        run<24>("dot8 today (12 instr)", blocks, 200, 85);
        run<20>("2 DPP chains interleaved", blocks, 200, 512);
        run<21>("4 DPP chains interleaved", blocks, 200, 256);
        run<13>("add + ds_read_b64", blocks, 200, 256);
        run<14>("add + ds_read_b32", blocks, 200, 256);
        run<15>("add + ds_read_b128", blocks, 200, 256);
        run<16>("add + ds_read2_b32", blocks, 200, 256);
        run<17>("add + ds_read2_b64", blocks, 200, 256);
        run<18>("4 adds + ds_read_b64", blocks, 200, 256);
        run<19>("8 adds + store_byte", blocks, 20, 128);
    }
    return 0;
}
