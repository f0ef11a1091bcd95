// ubench_lds_unaligned.hip -- does gfx950 serve a 16-byte LDS read that is only 4-byte aligned, and at what price?
// (the clock stage's 8-sample window starts at an arbitrary sample; its ring stores pair elements to keep every
// window 8-byte aligned, at twice the LDS).  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, unsigned long long *stamps, int reps, int skew) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 132 + 64];
    for (int i = threadIdx.x; i < 64 * 132 + 64; i += 64) lds[i] = (float) i;
    __syncthreads();
    // lane stride 132 floats (4 mod 64 banks... 132 = 2*64 + 4): aligned b128 reads are conflict-free
    unsigned addr = (unsigned) (size_t) (const __attribute__((address_space(3))) float *) lds + threadIdx.x * 132 * 4 + skew * 4 * (MODE == 2 ? (threadIdx.x & 3) : 1);
    float s = 0.0f;
    unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < reps; r++) {
        if (MODE == 0 || MODE == 2)
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %0\n\tds_read_b128 v[40:43], %1\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s) : "v"(addr) : "v40", "v41", "v42", "v43", "memory");
        if (MODE == 1)
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %0\n\tds_read2_b64 v[40:43], %1 offset0:0 offset1:1\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s) : "v"(addr) : "v40", "v41", "v42", "v43", "memory");
        if (MODE == 3)
            asm volatile(".rept 256\n\tv_add_f32 %0, %0, %0\n\tds_read_b96 v[40:42], %1\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "+v"(s) : "v"(addr) : "v40", "v41", "v42", "v43", "memory");
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float v0, v1, v2, v3;
    asm volatile("ds_read_b128 v[40:43], %4\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32 %0, v40\n\tv_mov_b32 %1, v41\n\tv_mov_b32 %2, v42\n\tv_mov_b32 %3, v43"
                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(addr) : "v40", "v41", "v42", "v43", "memory");
    out[threadIdx.x * 4 + 0] = v0 + s * 0.0f;
    out[threadIdx.x * 4 + 1] = v1;
    out[threadIdx.x * 4 + 2] = v2;
    out[threadIdx.x * 4 + 3] = v3;
    if (threadIdx.x == 0) stamps[0] = c1 - c0;
}

int main() {
    float *out, h_out[256];
    unsigned long long *st, h;
    hipMalloc(&out, 4096);
    hipMalloc(&st, 64);
    const int reps = 32;
#define RUN(M, SK, NAME) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, st, reps, SK); hipDeviceSynchronize(); \
    hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost); hipMemcpy(h_out, out, 1024, hipMemcpyDeviceToHost); \
    printf("%-58s %6.2f cycles per (v_add + read);  lane 1 got %.0f %.0f %.0f %.0f (want %d..)\n", NAME, (double) h / (reps * 256.0), h_out[4], h_out[5], h_out[6], h_out[7], 132 + SK * (M == 2 ? 1 : 1));
    RUN(0, 0, "ds_read_b128, 16-byte aligned")
    RUN(0, 1, "ds_read_b128, +4 bytes")
    RUN(0, 2, "ds_read_b128, +8 bytes")
    RUN(0, 3, "ds_read_b128, +12 bytes")
    RUN(2, 1, "ds_read_b128, +4*(lane&3) bytes (mixed alignments)")
    RUN(1, 0, "ds_read2_b64, 8-byte aligned")
    RUN(1, 1, "ds_read2_b64, +4 bytes")
    RUN(3, 1, "ds_read_b96, +4 bytes")
    return 0;
}
