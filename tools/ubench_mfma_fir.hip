// ubench_mfma_fir.hip -- round 4: can the exact-mode FIR's multiplies leave the vector pipe after all?
//
// Round 2 tried v_mfma_f32_4x4x1_16b_f32 as a multiplier (tools/ubench_mfma_mul.hip): exact, but that 8-cycle instruction
// holds the SIMD's vector issue for its whole life, so nothing ran beside it.  The larger K = 1 shapes are OUTER PRODUCTS --
// v_mfma_f32_16x16x1_4b_f32: four blocks of (16 samples) x (16 taps), 1024 separately rounded products in 32 cycles -- and
// an MFMA of that length holds the vector issue for only part of it (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost').
// An outer product is every sample times every tap, which is exactly the set of products a FIR needs; what the outer product
// loses is the sliding structure, so the sums must travel across lanes: with taps on the lanes, a sample step is
// "acc[p] = acc[p - 1] + product[p]" -- a shift register over the tap positions.  Laid out as position p = s + 8 * lane over
// eight registers s, that step is seven plain v_add_f32 (register renaming does the shift) and ONE v_add_f32 with a DPP
// row_shr:1 (bound_ctrl: a +0 enters at position 0), every lane busy.
//
// This program measures whether that mix runs at the rate the two pipes allow together:
//   mode 0  the present scheme: v_pk_mul_f32 + v_pk_add_f32 per two MACs (reference point)
//   mode 1  adds alone: 8 v_add_f32 (1 with DPP) per sample step
//   mode 2  MFMA 16x16x1_4b alone: 2 per 4 sample steps
//   mode 3  both, software-pipelined in ONE wave (the adds of step group g consume the products of group g - 1)
//   mode 4  mode 3 with v_mfma_f32_32x32x1_2b_f32 (one 64-cycle instruction instead of two 32-cycle ones)
// each with 1, 2 and 4 waves per SIMD, whole chip, and checks the products and the shift-register sums of mode 3 against plain
// C arithmetic (separately rounded multiply, then add, in tap order).
// MACs are counted as "component multiply-adds a FIR would need": 4 steps x 128 positions x 64 / 16 ... see macs_per_iter.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_mfma_fir.hip -o tools/ubench_mfma_fir
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define ADD(dst, a, b) asm volatile("v_add_f32 %0, %1, %2" : "=v"(dst) : "v"(a), "v"(b))
// dst = (a shifted one lane up within its row of 16, +0 entering at lane 0) + b
#define ADD_SHR(dst, a, b) asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(dst) : "v"(a), "v"(b))

// one sample step of the shift register: positions p = s + 8 l; acc[s] <- acc[s - 1] + d[s], acc[0] <- shr(acc[7]) + d[0].
// Written with explicit temporaries so that all eight adds read the OLD values.
#define STEP(acc, d0, d1, d2, d3, d4, d5, d6, d7)                                         \
    do {                                                                                  \
        float n0, n1, n2, n3, n4, n5, n6, n7;                                             \
        ADD(n1, acc[0], d1); ADD(n2, acc[1], d2); ADD(n3, acc[2], d3); ADD(n4, acc[3], d4); \
        ADD(n5, acc[4], d5); ADD(n6, acc[5], d6); ADD(n7, acc[6], d7);                    \
        ADD_SHR(n0, acc[7], d0); /* last: the DPP source (the previous step's n7) was written eight instructions ago */ \
        acc[0] = n0; acc[1] = n1; acc[2] = n2; acc[3] = n3; acc[4] = n4; acc[5] = n5; acc[6] = n6; acc[7] = n7; \
    } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    float a = 1.0f + lane * 1e-3f, b0 = 0.999f - lane * 1e-4f, b1 = 1.001f + lane * 1e-4f;
    float acc[8];
    for (int i = 0; i < 8; i++) acc[i] = 0.0f;
    f16v zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32v zero32;
    for (int i = 0; i < 32; i++) zero32[i] = 0.0f;
    f16v dA = zero16, dB = zero16;  // products of the previous group of four sample steps (residues 0-3 / 4-7)
    f32v dW = zero32;
    f2 pa[8], px = {a, b0}, ph = {b1, a};
    for (int i = 0; i < 8; i++) pa[i] = {0.0f, 0.0f};
    for (int it = 0; it < iters; it++) {
        if (MODE == 0) {
            // the same number of component-MACs as one group of modes 1-4: 4 steps x 8 registers x 64 lanes = 2048 per wave,
            // = 16 (pk_mul + pk_add) pairs
#pragma unroll
            for (int u = 0; u < 16; u++) {
                f2 p;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(px), "v"(ph));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pa[u & 7]) : "v"(p));
            }
        } else {
            f16v nA = dA, nB = dB;
            f32v nW = dW;
            if (MODE == 2 || MODE == 3) {
                nA = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b0, zero16, 0, 0, 0);
                nB = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b1, zero16, 0, 0, 0);
            }
            if (MODE == 4) {
                nW = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b0, zero32, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 3) {
                // register 4 b + q of a 16x16x1_4b result: block b (tap residue), row group = lane / 16, q = sample step
                STEP(acc, dA[0], dA[4], dA[8], dA[12], dB[0], dB[4], dB[8], dB[12]);
                STEP(acc, dA[1], dA[5], dA[9], dA[13], dB[1], dB[5], dB[9], dB[13]);
                STEP(acc, dA[2], dA[6], dA[10], dA[14], dB[2], dB[6], dB[10], dB[14]);
                STEP(acc, dA[3], dA[7], dA[11], dA[15], dB[3], dB[7], dB[11], dB[15]);
            }
            if (MODE == 4) {
                STEP(acc, dW[0], dW[4], dW[8], dW[12], dW[16], dW[20], dW[24], dW[28]);
                STEP(acc, dW[1], dW[5], dW[9], dW[13], dW[17], dW[21], dW[25], dW[29]);
                STEP(acc, dW[2], dW[6], dW[10], dW[14], dW[18], dW[22], dW[26], dW[30]);
                STEP(acc, dW[3], dW[7], dW[11], dW[15], dW[19], dW[23], dW[27], dW[31]);
            }
            if (MODE == 2) {
                asm volatile("" : "+v"(nA), "+v"(nB));
            }
            dA = nA;
            dB = nB;
            dW = nW;
            a += 1e-6f;
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i] + pa[i].x + pa[i].y;
    s += dA[0] + dB[5] + dW[7];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- correctness of the scheme on a small FIR: 2 streams x {re, im} in the four row groups, 128 tap positions (117 real
// taps + zero padding), samples fed four at a time; output k of a stream = sum_j x[k + j] * h[j], j ascending, every product
// and every sum rounded separately.  Host reference below.
#define NTAPS 117
#define NSAMP 512  // per row group
__global__ __launch_bounds__(64) void fir_check(const float *x /* [4][NSAMP] */, const float *h /* [128] */, float *y /* [4][NSAMP] */) {
    const int lane = threadIdx.x;
    const int blk = lane >> 4, row = lane & 15;  // A operand: block, row within block
    const int group = row >> 2, q = row & 3;     // row 4 g + q = sample step q of row group g
    const int col = lane & 15;                   // B operand / result: column
    // B: block b holds tap residue s: tap position p = s + 8 * col
    const float hb0 = h[(blk) + 8 * col], hb1 = h[(4 + blk) + 8 * col];
    float acc[8];
    for (int i = 0; i < 8; i++) acc[i] = 0.0f;
    const f16v zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int out_group = lane >> 4;  // results: row group of this lane
    for (int m0 = 0; m0 < NSAMP; m0 += 4) {
        const float a = x[group * NSAMP + m0 + q];  // the same value in all four blocks
        const f16v dA = __builtin_amdgcn_mfma_f32_16x16x1f32(a, hb0, zero16, 0, 0, 0);
        const f16v dB = __builtin_amdgcn_mfma_f32_16x16x1f32(a, hb1, zero16, 0, 0, 0);
#define EMIT(qq)                                                                                        \
        do {                                                                                            \
            /* position 116 = residue 4, lane 14: the finished sum of output k = m - 116 sits in acc[4] */ \
            const int m = m0 + (qq);                                                                    \
            if (col == 14 && m >= NTAPS - 1) y[out_group * NSAMP + m - (NTAPS - 1)] = acc[4];           \
        } while (0)
        STEP(acc, dA[0], dA[4], dA[8], dA[12], dB[0], dB[4], dB[8], dB[12]); EMIT(0);
        STEP(acc, dA[1], dA[5], dA[9], dA[13], dB[1], dB[5], dB[9], dB[13]); EMIT(1);
        STEP(acc, dA[2], dA[6], dA[10], dA[14], dB[2], dB[6], dB[10], dB[14]); EMIT(2);
        STEP(acc, dA[3], dA[7], dA[11], dA[15], dB[3], dB[7], dB[11], dB[15]); EMIT(3);
    }
}

template <int MODE>
static void run_rate(const char *name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;  // 256 threads = one wave per SIMD per block
    float *d;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double) blocks * 4 * iters * 2048.0;  // component-MACs (or their adds / products alone) per wave and iteration
    printf("%-58s %d wave(s)/SIMD  %8.3f ms  %6.1f T/s  (%5.1f cycles per group of 2048 at 2.4 GHz)\n", name, waves_per_simd, ms,
           macs / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / iters / waves_per_simd);
    hipFree(d);
}

int main() {
    // ---- correctness
    float *hx = (float *) malloc(sizeof(float) * 4 * NSAMP), *hh = (float *) calloc(128, sizeof(float));
    float *hy = (float *) malloc(sizeof(float) * 4 * NSAMP), *ref = (float *) calloc(4 * NSAMP, sizeof(float));
    srand(7);
    for (int i = 0; i < 4 * NSAMP; i++) hx[i] = (float) ((rand() % 20001) - 10000) * 1e-4f * (i % 97 == 0 ? 1e-30f : 1.0f);
    for (int j = 0; j < NTAPS; j++) hh[j] = (float) (sin(0.3 * (j - 58)) / (0.3 * (j - 58) + 1e-9) * (0.54 - 0.46 * cos(6.2831853 * j / 116.0))) * 0.05f;
    for (int g = 0; g < 4; g++) {
        for (int k = 0; k + NTAPS <= NSAMP; k++) {
            volatile float acc = 0.0f;
            for (int j = 0; j < NTAPS; j++) {
                volatile float p = hx[g * NSAMP + k + j] * hh[j];
                acc = acc + p;
            }
            ref[g * NSAMP + k] = acc;
        }
    }
    float *dx, *dh, *dy;
    hipMalloc(&dx, sizeof(float) * 4 * NSAMP);
    hipMalloc(&dh, sizeof(float) * 128);
    hipMalloc(&dy, sizeof(float) * 4 * NSAMP);
    hipMemcpy(dx, hx, sizeof(float) * 4 * NSAMP, hipMemcpyHostToDevice);
    hipMemcpy(dh, hh, sizeof(float) * 128, hipMemcpyHostToDevice);
    hipMemset(dy, 0, sizeof(float) * 4 * NSAMP);
    hipLaunchKernelGGL(fir_check, dim3(1), dim3(64), 0, 0, dx, dh, dy);
    hipDeviceSynchronize();
    hipMemcpy(hy, dy, sizeof(float) * 4 * NSAMP, hipMemcpyDeviceToHost);
    long bad = 0, checked = 0;
    for (int g = 0; g < 4; g++) {
        for (int k = 0; k + NTAPS <= NSAMP; k++) {
            checked++;
            if (memcmp(&hy[g * NSAMP + k], &ref[g * NSAMP + k], 4) != 0) {
                if (bad < 5) printf("  mismatch group %d output %d: device %.9g host %.9g\n", g, k, hy[g * NSAMP + k], ref[g * NSAMP + k]);
                bad++;
            }
        }
    }
    printf("shift-register FIR on MFMA products (117 taps, 4 streams x %d outputs): %ld of %ld outputs differ from the separately rounded "
           "multiply-then-add in tap order\n", NSAMP - NTAPS + 1, bad, checked);
    // ---- rates
    for (int w = 1; w <= 4; w *= 2) {
        run_rate<0>("0: v_pk_mul_f32 + v_pk_add_f32 (present scheme)", w);
        run_rate<1>("1: adds alone (7 v_add_f32 + 1 v_add_f32_dpp per step)", w);
        run_rate<2>("2: MFMA 16x16x1_4b alone (2 per 4 steps)", w);
        run_rate<3>("3: MFMA 16x16x1_4b + the adds, pipelined in one wave", w);
        run_rate<4>("4: MFMA 32x32x1_2b + the adds, pipelined in one wave", w);
    }
    return bad ? 1 : 0;
}
