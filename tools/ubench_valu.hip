// ubench_valu.hip -- how fast does gfx950 issue the fp32 VALU forms the exact-mode FIR can be built from?
// (packed vs scalar mul/add/fma).  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b) {
    v2f acc[16];
    v2f x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        acc[i] = (v2f){(float) threadIdx.x * 1e-3f + i, 1.0f + i};
        x[i] = (v2f){a + i, b - i};
    }
    v2f t = (v2f){a, a};
    for (int it = 0; it < iters; it++) {
        // keep the operands opaque: otherwise x*t is loop-invariant and the "mul+add" modes only time the adds
#pragma unroll
        for (int i = 0; i < 16; i++) {
            asm volatile("" : "+v"(x[i]));
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (MODE == 0) {  // packed mul + packed add (exact mode, complex sample x real tap)
                v2f p = x[i] * t;
                acc[i] = acc[i] + p;
            } else if (MODE == 1) {  // scalar mul + add on each half
                float p0 = x[i].x * t.x, p1 = x[i].y * t.x;
                asm volatile("" : "+v"(p0), "+v"(p1));
                acc[i].x = acc[i].x + p0;
                acc[i].y = acc[i].y + p1;
            } else if (MODE == 2) {  // packed fma
                acc[i] = __builtin_elementwise_fma(x[i], t, acc[i]);
            } else {  // scalar fma
                float f0 = __builtin_fmaf(x[i].x, t.x, acc[i].x);
                asm volatile("" : "+v"(f0));
                float f1 = __builtin_fmaf(x[i].y, t.x, acc[i].y);
                asm volatile("" : "+v"(f1));
                acc[i].x = f0;
                acc[i].y = f1;
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            asm volatile("" : "+v"(acc[i]));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int blocks, int iters) {
    float *d;
    hipMalloc(&d, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per iteration per lane: 16 complex mul-adds = 32 mul + 32 add (or 32 fma)
    double lane_macs = (double) blocks * 256 * iters * 32.0;
    printf("%-28s blocks %5d  %8.3f ms  %8.2f T complex-component MAC/s (x2 = flop/s %.1f T)\n", name, blocks, ms,
           lane_macs / ms / 1e9, 2 * lane_macs / ms / 1e9);
    hipFree(d);
}

int main() {
    for (int blocks : {256, 1024, 2048}) {
        run<0>("pk_mul + pk_add (exact)", blocks, 20000);
        run<1>("mul + add scalar (exact)", blocks, 20000);
        run<2>("pk_fma", blocks, 20000);
        run<3>("fma scalar", blocks, 20000);
    }
    return 0;
}
