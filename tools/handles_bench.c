/* handles_bench.c -- the reference's own usage at scale: T client threads, each with ITS OWN fsk_demod handle, each
 * calling fsk_demod_process once per buffer (src/dsp_worker.c:44-106).  Run it twice: as is (every handle a private
 * batch of one: one small launch sequence per buffer) and with SDRM_SHARED_SLOTS=T (the handles share one batcher).
 * Build: gcc -O2 -pthread tools/handles_bench.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip
 *            -Wl,-rpath,'$ORIGIN/../sdr-modem_amd/csrc' -lm -o tools/handles_bench
 * Run:   tools/handles_bench [threads] [buffer samples] [buffers per thread] */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "sdrmodem_hip.h"

static size_t n_buf = 4096, n_calls = 50;
static float *iq;
static pthread_barrier_t go;
static unsigned long long symbols;
static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

static void *client(void *arg) {
    fsk_demod *d = (fsk_demod *) arg;
    unsigned long long mine = 0;
    pthread_barrier_wait(&go);
    for (size_t k = 0; k < n_calls; k++) {
        int8_t *soft = NULL;
        size_t n = 0;
        fsk_demod_process((const sdrm_cf32 *) iq, n_buf, &soft, &n, d);
        mine += n;
    }
    pthread_mutex_lock(&lock);
    symbols += mine;
    pthread_mutex_unlock(&lock);
    return NULL;
}

int main(int argc, char **argv) {
    int threads = argc > 1 ? atoi(argv[1]) : 64;
    if (argc > 2) n_buf = (size_t) atol(argv[2]);
    if (argc > 3) n_calls = (size_t) atol(argv[3]);
    iq = malloc(sizeof(float) * 2 * n_buf);
    double ph = 0.0;
    unsigned lfsr = 0xACE1u;
    int bit = 1;
    for (size_t i = 0; i < n_buf; i++) {
        if (i % 5 == 0) {
            lfsr = (lfsr >> 1) ^ (-(lfsr & 1u) & 0xB400u);
            bit = (lfsr & 1u) ? 1 : -1;
        }
        ph += 2.0 * M_PI * 2400.0 * bit / 48000.0;
        iq[2 * i] = (float) cos(ph);
        iq[2 * i + 1] = (float) sin(ph);
    }
    fsk_demod **d = calloc((size_t) threads, sizeof(*d));
    for (int i = 0; i < threads; i++) {
        if (fsk_demod_create(48000, 9600, 5000, 1, 2000, true, (uint32_t) n_buf, &d[i]) != 0) {
            fprintf(stderr, "fsk_demod_create failed\n");
            return 1;
        }
    }
    pthread_t *t = calloc((size_t) threads, sizeof(*t));
    pthread_barrier_init(&go, NULL, (unsigned) threads + 1);
    for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, client, d[i]);
    pthread_barrier_wait(&go);
    const double t0 = now();
    for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
    const double dt = now() - t0;
    printf("%d handles x %zu buffers of %zu samples (%s): %.1f ms, %.1f Msamples/s, %.0f symbols per buffer\n", threads, n_calls,
           n_buf, getenv("SDRM_SHARED_SLOTS") ? "shared batcher" : "private batches", dt * 1e3,
           (double) threads * (double) n_calls * (double) n_buf / dt / 1e6, (double) symbols / ((double) threads * (double) n_calls));
    for (int i = 0; i < threads; i++) fsk_demod_destroy(d[i]);
    return 0;
}
