// ubench_mfma_mul.hip -- can the matrix pipe do the exact-mode FIR's multiplies?  v_mfma_f32_4x4x1_16b_f32 with K = 1 and
// C = -0 is ONE product per output element (lane l, element i: a[lane 4*(l/4)+i] * b[lane l] + (-0)), i.e. fl(a*b) if the
// unit rounds once and keeps denormals -- checked here against v_mul_f32 bit for bit -- and it runs beside the VALU.
// Prints: mismatches of the products, and the rate of "4 products + their 4 separately rounded adds" per lane in three
// forms: VALU only (2 v_pk_mul + 2 v_pk_add), MFMA + 2 v_pk_add, and the adds alone.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_mfma_mul.hip -o tools/ubench_mfma_mul
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ void check(const float *a, const float *b, float *out_mfma, float *out_mul, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const f4 negz = {-0.0f, -0.0f, -0.0f, -0.0f};
    f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t], b[t], negz, 0, 0, 0);
    const int blk = t & ~3;
    for (int i = 0; i < 4; i++) {
        out_mfma[(size_t) t * 4 + i] = d[i];
        out_mul[(size_t) t * 4 + i] = a[blk + i] * b[t];
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(float *out, int iters) {
    f2 x = {1.0f + threadIdx.x * 1e-3f, 0.5f}, h = {0.999f, 1.001f};
    f2 acc[8];
    for (int i = 0; i < 8; i++) acc[i] = {0.0f, 0.0f};
    const f4 negz = {-0.0f, -0.0f, -0.0f, -0.0f};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (MODE == 0) {  // VALU only: 4 products + 4 adds = 2 pk_mul + 2 pk_add
                f2 p0, p1;
                asm volatile("v_pk_mul_f32 %0, %2, %3\n\tv_pk_mul_f32 %1, %2, %3" : "=&v"(p0), "=&v"(p1) : "v"(x), "v"(h));
                asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3" : "+v"(acc[2 * u]), "+v"(acc[2 * u + 1]) : "v"(p0), "v"(p1));
            } else if (MODE == 1) {  // matrix pipe for the products
                f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, h.x, negz, 0, 0, 0);
                f2 p0 = {d[0], d[1]}, p1 = {d[2], d[3]};
                asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %3" : "+v"(acc[2 * u]), "+v"(acc[2 * u + 1]) : "v"(p0), "v"(p1));
            } else if (MODE == 3) {  // the products alone, on the matrix pipe (results folded into acc so they stay live)
                f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, h.x, negz, 0, 0, 0);
                asm volatile("" : "+v"(d));
                if (it == iters + 7) { acc[u].x += d[0] + d[1] + d[2] + d[3]; }
            } else if (MODE == 4) {  // products on the matrix pipe, INDEPENDENT adds on the vector pipe (no data dependence)
                f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(x.x, h.x, negz, 0, 0, 0);
                asm volatile("" : "+v"(d));
                asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2" : "+v"(acc[2 * u]), "+v"(acc[2 * u + 1]) : "v"(x));
                if (it == iters + 7) { acc[u].x += d[0] + d[1] + d[2] + d[3]; }
            } else {  // the adds alone
                asm volatile("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2" : "+v"(acc[2 * u]), "+v"(acc[2 * u + 1]) : "v"(x));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run_rate(const char *name) {
    const int blocks = 256 * 4 * 2, iters = 4096;
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
    const double macs = (double) blocks * 256 * iters * 4 * 4;  // 4 groups of 4 products-and-adds per lane and iteration
    printf("%-44s %8.3f ms, %6.1f T product-and-add per second\n", name, ms, macs / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main() {
    const int n = 1 << 20;
    float *ha = (float *) malloc(4 * n), *hb = (float *) malloc(4 * n);
    srand(7);
    for (int i = 0; i < n; i++) {
        uint32_t ua = ((uint32_t) rand() << 16) ^ (uint32_t) rand(), ub = ((uint32_t) rand() << 16) ^ (uint32_t) rand();
        if (i % 5 == 0) ua &= 0x807fffffu;        // denormal a
        if (i % 7 == 0) ub = (ub & 0x80ffffffu) | 0x00800000u;  // tiny normal b: denormal / underflowing products
        if (i % 11 == 0) ua = (ua & 0x80000000u); // signed zero
        if (i % 3 == 0) { ua = (ua & 0x81ffffffu) | 0x3e000000u; ub = (ub & 0x81ffffffu) | 0x3e000000u; }  // ordinary magnitudes
        memcpy(&ha[i], &ua, 4);
        memcpy(&hb[i], &ub, 4);
    }
    float *da, *db, *d1, *d2;
    hipMalloc(&da, 4 * n); hipMalloc(&db, 4 * n); hipMalloc(&d1, 16 * (size_t) n); hipMalloc(&d2, 16 * (size_t) n);
    hipMemcpy(da, ha, 4 * n, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, 4 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, da, db, d1, d2, n);
    uint32_t *h1 = (uint32_t *) malloc(16 * (size_t) n), *h2 = (uint32_t *) malloc(16 * (size_t) n);
    hipMemcpy(h1, d1, 16 * (size_t) n, hipMemcpyDeviceToHost);
    hipMemcpy(h2, d2, 16 * (size_t) n, hipMemcpyDeviceToHost);
    size_t bad = 0, bad_nan = 0, shown = 0, denorm_out = 0;
    for (size_t i = 0; i < (size_t) n * 4; i++) {
        const bool nan1 = (h1[i] & 0x7fffffffu) > 0x7f800000u, nan2 = (h2[i] & 0x7fffffffu) > 0x7f800000u;
        if ((h2[i] & 0x7f800000u) == 0 && (h2[i] & 0x7fffffu) != 0) denorm_out++;
        if (nan1 && nan2) continue;
        if (h1[i] != h2[i]) {
            bad++;
            if (nan1 != nan2) bad_nan++;
            if (shown++ < 6) printf("  mismatch at %zu: mfma %08x  v_mul %08x\n", i, h1[i], h2[i]);
        }
    }
    printf("products checked: %zu (of them %zu denormal results), mismatches mfma(a,b,-0) vs v_mul_f32: %zu (NaN-ness differs: %zu)\n",
           (size_t) n * 4, denorm_out, bad, bad_nan);
    run_rate<0>("VALU: 2 v_pk_mul + 2 v_pk_add per 4");
    run_rate<1>("MFMA 4x4x1 (products) + 2 v_pk_add per 4");
    run_rate<2>("2 v_pk_add per 4 (adds alone)");
    run_rate<3>("MFMA 4x4x1 alone");
    run_rate<4>("MFMA 4x4x1 + 2 independent v_pk_add per 4");
    return 0;
}
