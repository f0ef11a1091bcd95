// ubench_fir.hip -- what limits the register-blocked exact-mode complex FIR inner loop of K1?
// Variants: window reloaded from LDS each step or not; taps from s_load (SGPR) or fixed registers; U = 4 or 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int R, int U, bool LOADW, bool TAPS_MEM, bool TAPV = false, int WPS = 4>
__global__ __launch_bounds__(256, WPS) void k(const float *__restrict__ taps, float *out, int nchunks) {
    __shared__ v2f xs[256 * (WPS > 4 ? 7 : R) + 640];
    for (int i = threadIdx.x; i < 256 * (WPS > 4 ? 7 : R) + 640; i += 256) xs[i] = (v2f){i * 1e-3f, -i * 1e-3f};
    __syncthreads();
    v2f acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = (v2f){0.f, 0.f};
    const v2f *base = xs + threadIdx.x * R;
    v2f w[R + U - 1];
#pragma unroll
    for (int k2 = 0; k2 < R + U - 1; k2++) w[k2] = base[k2];
    float tp[U];
#pragma unroll
    for (int u = 0; u < U; u++) tp[u] = taps[u];
    for (int c = 0; c < nchunks; c++) {
        if (LOADW) {
#pragma unroll
            for (int k2 = 0; k2 < R + U - 1; k2++) w[k2] = base[c * U + k2];
        }
        if (TAPS_MEM) {
#pragma unroll
            for (int u = 0; u < U; u++) tp[u] = taps[c * U + u];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            v2f tv = (v2f){tp[u], tp[u]};
            if (TAPV) asm volatile("" : "+v"(tv));  // tap pair lives in VGPRs
#pragma unroll
            for (int r = 0; r < R; r++) {
                v2f p = TAPV ? w[r + u] * tv : w[r + u] * tp[u];
                acc[r] = acc[r] + p;
            }
        }
        if (!LOADW) {
#pragma unroll
            for (int k2 = 0; k2 < R + U - 1; k2++) asm volatile("" : "+v"(w[k2]));
        }
    }
    v2f s = (v2f){0.f, 0.f};
#pragma unroll
    for (int r = 0; r < R; r++) s += acc[r];
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

template <int R, int U, bool LOADW, bool TAPS_MEM, bool TAPV = false, int WPS = 4>
static void run(const char *name, int blocks) {
    float *taps, *out;
    hipMalloc(&taps, 4096 * 4);
    hipMemset(taps, 0, 4096 * 4);
    hipMalloc(&out, blocks * 256 * 4);
    int nchunks = 480 / U;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<R, U, LOADW, TAPS_MEM, TAPV, WPS>), dim3(blocks), dim3(256), 0, 0, taps, out, nchunks);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, U, LOADW, TAPS_MEM, TAPV, WPS>), dim3(blocks), dim3(256), 0, 0, taps, out, nchunks);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double macs = (double) blocks * 256 * R * 480 * 2.0;  // component MACs
    printf("%-44s %8.3f ms  %6.2f T component-MAC/s\n", name, ms, macs / ms / 1e9);
    hipFree(taps);
    hipFree(out);
}

int main() {
    const int B = 8192;
    run<15, 4, true, true>("R15 U4 window from LDS, taps s_load", B);
    run<15, 4, true, false>("R15 U4 window from LDS, taps fixed", B);
    run<15, 4, false, true>("R15 U4 window fixed, taps s_load", B);
    run<15, 4, false, false>("R15 U4 window fixed, taps fixed", B);
    run<15, 8, true, true>("R15 U8 window from LDS, taps s_load", B);
    run<15, 8, false, false>("R15 U8 window fixed, taps fixed", B);
    run<15, 4, true, true, true>("R15 U4 window from LDS, taps s_load -> VGPR pair", B);
    run<15, 4, false, false, true>("R15 U4 window fixed, taps fixed VGPR pair", B);
    run<7, 4, true, true, false, 8>("R7 U4 LDS+s_load, 8 waves/SIMD", B);
    run<7, 4, false, false, false, 8>("R7 U4 fixed/fixed, 8 waves/SIMD", B);
    run<16, 1, false, false, true, 4>("R16 U1 fixed/fixed VGPR tap (like ubench_valu)", B);
    run<7, 8, true, true>("R7 U8 window from LDS, taps s_load", B);
    run<11, 6, true, true>("R11 U6 window from LDS, taps s_load", B);
    return 0;
}
