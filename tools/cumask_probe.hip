// cumask_probe.hip -- does hipExtStreamCreateWithCUMask partition the chip here, and which mask bit is which CU?
// Finding (MI355X, ROCm 7.2): bit i of the mask is CU (i / 8) of XCD (i % 8); an XCD whose bits are all clear runs on ALL
// of its CUs.  A uniform subset of every XCD is therefore a set of whole BYTES.
// hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe && tools/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__global__ void where(unsigned *hits, int spin) {
    // HW_REG_HW_ID (4): cu_id [11:8], sh_id [12], se_id [15:13]; HW_REG_XCC_ID (20): xcc_id [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    float a = threadIdx.x;
    for (int i = 0; i < spin; i++) {
        a = a * 1.0001f + 0.5f;
    }
    if (threadIdx.x == 0) {
        atomicAdd(&hits[((xcc * 8 + se) * 2 + sh) * 16 + cu], 1u + (a == 12345.0f));
    }
}

static void report(const char *name, hipStream_t s, unsigned *d_hits) {
    const int slots = 16 * 8 * 2 * 16;
    hipMemsetAsync(d_hits, 0, slots * 4, s);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(where, dim3(256 * 64), dim3(256), 0, s, d_hits, 20000);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(slots);
    hipMemcpy(h.data(), d_hits, slots * 4, hipMemcpyDeviceToHost);
    int used = 0;
    printf("%s: %.3f ms;", name, ms);
    for (int x = 0; x < 16; x++) {
        int in_x = 0;
        for (int k = 0; k < 8 * 2 * 16; k++) in_x += h[x * 256 + k] != 0;
        if (in_x) printf(" xcc%d:%d", x, in_x);
        used += in_x;
    }
    printf("  -> %d CUs used\n", used);
    if (getenv("VERBOSE")) {
        for (int i = 0; i < slots; i++) if (h[i]) printf("  xcc %d se %d sh %d cu %d: %u\n", i / 256, (i / 32) % 8, (i / 16) % 2, i % 16, h[i]);
    }
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs\n", p.name, p.multiProcessorCount);
    unsigned *d_hits;
    hipMalloc(&d_hits, 16 * 8 * 2 * 16 * 4);
    hipStream_t s0;
    hipStreamCreate(&s0);
    report("no mask", s0, d_hits);
    const int words = (p.multiProcessorCount + 31) / 32;
    struct { const char *name; uint32_t pat; } masks[] = {
        {"even bits", 0x55555555u}, {"low half of each word", 0x0000ffffu}, {"3 of 4 bytes", 0x00ffffffu}, {"1 of 4 bytes", 0xff000000u},
        {"low byte of each word", 0x000000ffu}};
    for (auto &m : masks) {
        std::vector<uint32_t> mask(words, m.pat);
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, words, mask.data());
        if (e != hipSuccess) {
            printf("%s: hipExtStreamCreateWithCUMask -> %s\n", m.name, hipGetErrorString(e));
            continue;
        }
        report(m.name, s, d_hits);
        hipStreamDestroy(s);
    }
    // first words only: which XCDs do the first 64 bits cover?
    {
        std::vector<uint32_t> mask(words, 0u);
        mask[0] = 0xffffffffu;
        mask[1] = 0xffffffffu;
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, words, mask.data()) == hipSuccess) {
            report("bits 0-63 only", s, d_hits);
            hipStreamDestroy(s);
        }
    }
    // two complementary masks at once: do the two kernels overlap in time without sharing CUs?
    {
        std::vector<uint32_t> a(words, 0x00ffffffu), b(words, 0xff000000u);
        hipStream_t sa, sb;
        if (hipExtStreamCreateWithCUMask(&sa, words, a.data()) == hipSuccess && hipExtStreamCreateWithCUMask(&sb, words, b.data()) == hipSuccess) {
            unsigned *d2;
            hipMalloc(&d2, 16 * 8 * 2 * 16 * 4);
            hipMemset(d2, 0, 16 * 8 * 2 * 16 * 4);
            hipMemset(d_hits, 0, 16 * 8 * 2 * 16 * 4);
            hipEvent_t e0, e1, e2;
            hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
            hipDeviceSynchronize();
            hipEventRecord(e0, sa);
            hipLaunchKernelGGL(where, dim3(256 * 64), dim3(256), 0, sa, d_hits, 20000);
            hipLaunchKernelGGL(where, dim3(64 * 64), dim3(256), 0, sb, d2, 20000);
            hipEventRecord(e1, sa);
            hipEventRecord(e2, sb);
            hipDeviceSynchronize();
            float ma = 0, mb = 0;
            hipEventElapsedTime(&ma, e0, e1);
            hipEventElapsedTime(&mb, e0, e2);
            std::vector<unsigned> ha(4096), hb(4096);
            hipMemcpy(ha.data(), d_hits, 4096 * 4, hipMemcpyDeviceToHost);
            hipMemcpy(hb.data(), d2, 4096 * 4, hipMemcpyDeviceToHost);
            int both = 0, na = 0, nb = 0;
            for (int i = 0; i < 4096; i++) { na += ha[i] != 0; nb += hb[i] != 0; both += ha[i] && hb[i]; }
            printf("concurrent 3/4 + 1/4 masks: A %.3f ms on %d CUs, B (quarter of the work) done at %.3f ms on %d CUs, %d CUs shared\n", ma, na, mb, nb, both);
        }
    }
    return 0;
}
