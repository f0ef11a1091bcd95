// ubench_dcchain.hip -- design numbers for the lane-dense DC blocker (round 2): one CHAIN wave per workgroup runs the
// running sums of 4 stages x 16 channels (lane = stage, channel), 64 time steps per block, operands from LDS rows that are
// contiguous in time (ds_read_b128 / ds_write_b128); HELPER waves of the same workgroup do the pointwise work between two
// barriers (here: a stand-in with the same instruction mix per row: LDS read, 4 VALU, LDS write, LDS read, VALU, LDS write).
// Prints shader cycles per 64-step iteration for several helper counts and grid sizes.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_dcchain.hip -o tools/ubench_dcchain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define ROWS 64          // (stage, channel) rows
#define BLK 64           // steps per block
#define PITCH 132        // floats per row: two block buffers + one 16-byte unit of padding (33 units: odd)

template <int HELPERS, int ROWOPS, int CMODE>
__global__ __launch_bounds__(64 * (HELPERS + 1)) void k(float *out, unsigned long long *stamps, int iters) {
    __shared__ __attribute__((aligned(16))) float ts[ROWS * PITCH];
    __shared__ __attribute__((aligned(16))) float ring[3 * 16 * 256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < ROWS * PITCH; i += blockDim.x) ts[i] = 1e-3f * (i & 15);
    for (int i = threadIdx.x; i < 3 * 16 * 256; i += blockDim.x) ring[i] = 0.5f;
    __syncthreads();
    float acc = lane * 0.25f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (wave == 0) {
        __builtin_amdgcn_s_setprio(3);
        typedef float f4 __attribute__((ext_vector_type(4)));
        for (int it = 0; it < iters; it++) {
            f4 *row = reinterpret_cast<f4 *>(ts + lane * PITCH + (it & 1) * BLK);
            f4 v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = row[q];
            // CMODE 0: full (reads, adds, writes); 1: reads + adds; 2: reads only; 3: reads + writes, no adds
#pragma unroll
            for (int q = 0; q < 16; q++) {
                // strictly sequential: s[n] = s[n-1] + t[n], one rounding each, sums written over the terms
                if (CMODE == 0 || CMODE == 1) {
                    v[q].x = acc + v[q].x;
                    v[q].y = v[q].x + v[q].y;
                    v[q].z = v[q].y + v[q].z;
                    v[q].w = v[q].z + v[q].w;
                    acc = v[q].w;
                } else {
                    asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
                }
                if (CMODE == 0 || CMODE == 3) {
                    row[q] = v[q];
                }
            }
            __syncthreads();
        }
    } else {
        const int h = wave - 1;
        for (int it = 0; it < iters; it++) {
            for (int r = h; ROWOPS >= 0 && r < ROWS; r += HELPERS) {
                float *row = ts + r * PITCH + ((it + 1) & 1) * BLK;
                float *rg = ring + (r % 48) * 256;
                float s = row[lane];
                float v = s;
#pragma unroll
                for (int o = 0; o < ROWOPS; o++) {
                    v = __builtin_fmaf(v, 0.999f, 1e-3f);
                }
                rg[(it * 64 + lane) & 255] = v;
                float ud = rg[(it * 64 + lane - 160) & 255];
                row[lane] = (v - ud) * 1e-3f;
            }
            __syncthreads();
        }
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && wave == 0) {
        stamps[blockIdx.x] = c1 - c0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + ts[threadIdx.x];
}

template <int HELPERS, int ROWOPS, int CMODE>
static void run(int blocks, int iters) {
    float *out;
    unsigned long long *st;
    hipMalloc(&out, sizeof(float) * blocks * 64 * (HELPERS + 1));
    hipMalloc(&st, sizeof(unsigned long long) * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<HELPERS, ROWOPS, CMODE>), dim3(blocks), dim3(64 * (HELPERS + 1)), 0, 0, out, st, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<HELPERS, ROWOPS, CMODE>), dim3(blocks), dim3(64 * (HELPERS + 1)), 0, 0, out, st, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long *h = (unsigned long long *) malloc(sizeof(unsigned long long) * blocks);
    hipMemcpy(h, st, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (int i = 0; i < blocks; i++) {
        sum += (double) h[i];
        mx = h[i] > mx ? (double) h[i] : mx;
    }
    printf("cmode %d helpers %2d rowops %2d blocks %4d: %8.1f cycles/iteration (max %8.1f), %.3f ms for %d iterations\n", CMODE, HELPERS, ROWOPS, blocks,
           sum / blocks / iters, mx / iters, ms, iters);
    free(h);
    hipFree(out);
    hipFree(st);
}

int main() {
    const int iters = 2048;
    for (int blocks : {16, 256}) {
        run<1, -1, 0>(blocks, iters);
        run<1, -1, 1>(blocks, iters);
        run<1, -1, 2>(blocks, iters);
        run<1, -1, 3>(blocks, iters);
        run<3, -1, 0>(blocks, iters);
        run<15, -1, 0>(blocks, iters);
        run<15, 4, 2>(blocks, iters);
        run<15, 4, 0>(blocks, iters);
        run<7, 4, 0>(blocks, iters);
    }
    return 0;
}
