// dc_div_sweep.cpp -- exhaustive check of sdrm_boxcar_out_fast (csrc/sdrm_core.h): for every supported boxcar length L
// (32..3968) and EVERY fp32 significand, the three-instruction quotient equals the IEEE division whenever it does not
// raise its `unsafe` flag; for a set of lengths additionally every exponent (denormal inputs, quotients at the edge of
// the normal range, the largest finite values, infinities and NaN).  Also counts how often `unsafe` is raised.
// Build: g++ -O2 -mfma -ffp-contract=off -I sdr-modem_amd/csrc tools/dc_div_sweep.cpp -o tools/dc_div_sweep -lm -lpthread
// Run:   tools/dc_div_sweep [first_L last_L]      (8 threads; ~1 minute for the whole range)
#include <math.h>
#include <pthread.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdrm_core.h"

static uint64_t g_bad, g_unsafe, g_total;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

static void check(uint32_t bits, float lf, float inv, uint64_t *bad, uint64_t *uns) {
    float a;
    memcpy(&a, &bits, 4);
    bool unsafe;
    volatile float want = a / lf;
    float got = sdrm_boxcar_out_fast(a, lf, inv, &unsafe);
    if (unsafe) {
        (*uns)++;
        return;
    }
    float w = want;
    if (memcmp(&got, &w, 4) != 0 && !(got != got && w != w)) {
        if ((*bad)++ < 5) fprintf(stderr, "MISMATCH L=%g a=%a (0x%08x): fast %a, division %a\n", lf, a, bits, got, w);
    }
}

struct job { int l0, l1; };

static void *run(void *arg) {
    struct job *j = (struct job *) arg;
    uint64_t bad = 0, uns = 0, tot = 0;
    for (int L = j->l0; L <= j->l1; L++) {
        const float lf = (float) L, inv = 1.0f / lf;
        // every significand at two exponents (the arithmetic is scale-invariant while the quotient stays normal), both signs
        for (uint32_t m = 0; m < (1u << 23); m++) {
            check((127u << 23) | m, lf, inv, &bad, &uns);
            check(0x80000000u | (90u << 23) | m, lf, inv, &bad, &uns);
            tot += 2;
        }
        // every exponent for a sparse set of significands (including 0, all ones, and a pseudo-random walk)
        uint32_t m = 0x2545F4u;
        for (int k = 0; k < 4096; k++) {
            m = (m * 1664525u + 1013904223u);
            const uint32_t sig = (k == 0) ? 0u : (k == 1) ? 0x7fffffu : (m >> 9);
            for (uint32_t e = 0; e < 256; e++) {
                check((e << 23) | sig, lf, inv, &bad, &uns);
                check(0x80000000u | (e << 23) | sig, lf, inv, &bad, &uns);
                tot += 2;
            }
        }
    }
    pthread_mutex_lock(&g_mu);
    g_bad += bad;
    g_unsafe += uns;
    g_total += tot;
    pthread_mutex_unlock(&g_mu);
    return NULL;
}

int main(int argc, char **argv) {
    int first = argc > 2 ? atoi(argv[1]) : 32, last = argc > 2 ? atoi(argv[2]) : 3968;
    enum { T = 8 };
    pthread_t th[T];
    struct job jobs[T];
    int per = (last - first + T) / T;
    for (int t = 0; t < T; t++) {
        jobs[t].l0 = first + t * per;
        jobs[t].l1 = jobs[t].l0 + per - 1 < last ? jobs[t].l0 + per - 1 : last;
        pthread_create(&th[t], NULL, run, &jobs[t]);
    }
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    // denormal inputs and quotients, exhaustively, for a few lengths: all 2^23 denormals and the 13 binades above them
    static uint64_t lb[T], lu[T], lt[T];
    static const int some[T] = {32, 80, 154, 160, 400, 800, 1280, 3968};
    for (int t = 0; t < T; t++) {
        pthread_create(&th[t], NULL, [](void *arg) -> void * {
            const int t = (int) (intptr_t) arg;
            const float lf = (float) some[t], inv = 1.0f / lf;
            for (uint32_t bits = 0; bits < (14u << 23); bits++) {
                check(bits, lf, inv, &lb[t], &lu[t]);
                lt[t]++;
            }
            return NULL;
        }, (void *) (intptr_t) t);
    }
    uint64_t bad = 0, uns = 0, tot = 0;
    for (int t = 0; t < T; t++) {
        pthread_join(th[t], NULL);
        bad += lb[t];
        uns += lu[t];
        tot += lt[t];
    }
    printf("lengths %d..%d: %llu quotients checked, %llu mismatches, %llu flagged unsafe (all significands x 2 exponents x 2 signs, "
           "4096 significands x every exponent)\n", first, last, (unsigned long long) g_total, (unsigned long long) g_bad,
           (unsigned long long) g_unsafe);
    printf("lowest 14 binades (denormal inputs and quotients) for 8 lengths: %llu checked, %llu mismatches, %llu flagged unsafe\n",
           (unsigned long long) tot, (unsigned long long) bad, (unsigned long long) uns);
    return (g_bad + bad) ? 1 : 0;
}
