/* handles_bench.c -- the reference's own usage at scale: T client threads, each with ITS OWN demodulator, each blocking on one call
 * per buffer (src/dsp_worker.c:188 thread per client, :75 the blocking call, src/tcp_server.c:659 one worker per request,
 * src/resources/config.conf:11 buffer_size 131072).
 *
 *   tools/handles_bench [-w] [-q] [-s] [-W warm_up_buffers] threads buffer_samples buffers_per_thread [input.cf32 ...]
 *
 * Default mode: every thread owns an fsk_demod handle and calls fsk_demod_process once per buffer.  -w: every thread owns a
 * dsp_worker (private handle, file sink into a scratch directory) and feeds it with dsp_worker_put; the worker's own thread
 * demodulates.  Input: the given .cf32 files (thread i streams file i % n from its start, `buffers_per_thread` buffers of
 * `buffer_samples`: the file must hold that many) or, without files, one synthetic buffer repeated.
 * Output: per input file ("class") the number of handles, whether every handle of the class produced the same stream, its
 * symbol count and FNV-1a hash -- tests/test_gpu_handles.py compares those with the oracle's stream for the same file --, the
 * number of handles in the sticky error state, the device's hand-off ledger (sdrm_handoff_stats) and the wall time of the
 * buffers behind the first `warm_up_buffers` of every thread (default 2: a process's first calls load code objects, create streams
 * and grow the runtime's pools; they are part of every stream and of its hash, not of the time).
 * Run it with SDRM_HANDOFF=0 for the comparison, with SDRM_SHARED_SLOTS=T -- or -s, which calls sdrm_fsk_demod_share(T, 1000) instead
 * -- for handles that share one batcher.
 * Build: gcc -O2 -pthread tools/handles_bench.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip
 *            -Wl,-rpath,'$ORIGIN/../sdr-modem_amd/csrc' -lm -o tools/handles_bench */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "sdrmodem_hip.h"

#define MAX_FILES 64

static size_t n_buf = 131072, n_calls = 20, n_warm = 2;
static int n_files = 0, worker_mode = 0;
static float *file_iq[MAX_FILES];
static float *synth_iq;
static pthread_barrier_t go, warm;
static char scratch[256];

struct client {
    int index;
    fsk_demod *demod;
    dsp_worker *worker;
    unsigned long long symbols, hash;
    int error;
};

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

static unsigned long long fnv(unsigned long long h, const void *data, size_t n) {
    const unsigned char *p = (const unsigned char *) data;
    for (size_t i = 0; i < n; i++) {
        h = (h ^ p[i]) * 0x100000001b3ull;
    }
    return h;
}

static const float *buffer_of(const struct client *c, size_t k) {
    if (n_files == 0) {
        return synth_iq;
    }
    return file_iq[c->index % n_files] + 2 * k * n_buf;
}

static void *client_main(void *arg) {
    struct client *c = (struct client *) arg;
    c->hash = 0xcbf29ce484222325ull;
    pthread_barrier_wait(&go);
    for (size_t k = 0; k < n_calls; k++) {
        if (k == n_warm) {
            pthread_barrier_wait(&warm);  /* everybody's warm-up buffers are through: the clock starts (main) */
        }
        if (worker_mode) {
            dsp_worker_put((sdrm_cf32 *) buffer_of(c, k), n_buf, c->worker);  /* blocks while the queue is full (file source) */
            continue;
        }
        int8_t *soft = NULL;
        size_t n = 0;
        fsk_demod_process((const sdrm_cf32 *) buffer_of(c, k), n_buf, &soft, &n, c->demod);
        c->symbols += n;
        c->hash = fnv(c->hash, soft, n);
    }
    if (!worker_mode) {
        c->error = sdrm_fsk_demod_error(c->demod);
    }
    return NULL;
}

int main(int argc, char **argv) {
    int quiet = 0, share = 0, a = 1;
    for (; a < argc && argv[a][0] == '-'; a++) {
        if (strcmp(argv[a], "-w") == 0) worker_mode = 1;
        if (strcmp(argv[a], "-q") == 0) quiet = 1;
        if (strcmp(argv[a], "-s") == 0) share = 1;
        if (strcmp(argv[a], "-W") == 0 && a + 1 < argc) n_warm = (size_t) atol(argv[++a]);
    }
    int threads = a < argc ? atoi(argv[a++]) : 64;
    if (a < argc) n_buf = (size_t) atol(argv[a++]);
    if (a < argc) n_calls = (size_t) atol(argv[a++]);
    for (; a < argc && n_files < MAX_FILES; a++) {
        FILE *f = fopen(argv[a], "rb");
        const size_t want = 2 * n_buf * n_calls;
        file_iq[n_files] = malloc(sizeof(float) * want);
        if (f == NULL || file_iq[n_files] == NULL || fread(file_iq[n_files], sizeof(float), want, f) != want) {
            fprintf(stderr, "%s: cannot read %zu complex samples\n", argv[a], want / 2);
            return 2;
        }
        fclose(f);
        n_files++;
    }
    if (n_files == 0) {
        synth_iq = malloc(sizeof(float) * 2 * n_buf);
        double ph = 0.0;
        unsigned lfsr = 0xACE1u;
        int bit = 1;
        for (size_t i = 0; i < n_buf; i++) {
            if (i % 5 == 0) {
                lfsr = (lfsr >> 1) ^ (-(lfsr & 1u) & 0xB400u);
                bit = (lfsr & 1u) ? 1 : -1;
            }
            ph += 2.0 * M_PI * 2400.0 * bit / 48000.0;
            synth_iq[2 * i] = (float) cos(ph);
            synth_iq[2 * i + 1] = (float) sin(ph);
        }
    }
    struct client *cl = calloc((size_t) threads, sizeof(*cl));
    if (share && sdrm_fsk_demod_share((size_t) threads, 1000) != 0) {
        fprintf(stderr, "sdrm_fsk_demod_share failed\n");
        return 1;
    }
    if (worker_mode) {
        snprintf(scratch, sizeof(scratch), "/tmp/handles_bench.%d", (int) getpid());
        mkdir(scratch, 0700);
    }
    for (int i = 0; i < threads; i++) {
        cl[i].index = i;
        if (worker_mode) {
            sdrm_worker_config wc;
            memset(&wc, 0, sizeof(wc));
            wc.rx_sampling_freq = 48000;
            wc.demod_baud_rate = 9600;
            wc.demod_fsk_deviation = 5000;
            wc.demod_decimation = 1;
            wc.demod_fsk_transition_width = 2000;
            wc.demod_fsk_use_dc_block = true;
            wc.demod_destination = 0; /* FILE (api.proto DemodDestination) */
            wc.buffer_size = (uint32_t) n_buf;
            wc.queue_size = 4;
            wc.rx_file_source = true;
            wc.base_path = scratch;
            if (dsp_worker_create((uint32_t) i, -1, &wc, &cl[i].worker) != 0) {
                fprintf(stderr, "dsp_worker_create failed for client %d\n", i);
                return 1;
            }
        } else if (fsk_demod_create(48000, 9600, 5000, 1, 2000, true, (uint32_t) n_buf, &cl[i].demod) != 0) {
            fprintf(stderr, "fsk_demod_create failed for client %d\n", i);
            return 1;
        }
    }
    pthread_t *t = calloc((size_t) threads, sizeof(*t));
    if (n_warm >= n_calls) n_warm = 0;
    pthread_barrier_init(&go, NULL, (unsigned) threads + 1);
    pthread_barrier_init(&warm, NULL, (unsigned) threads + 1);
    for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, client_main, &cl[i]);
    pthread_barrier_wait(&go);
    pthread_barrier_wait(&warm);
    const double t0 = now();
    for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
    if (worker_mode) {
        /* the workers drain their queues and stop (the reference's shutdown: src/dsp_worker.c:191-215), then their files are read back */
        for (int i = 0; i < threads; i++) dsp_worker_destroy(cl[i].worker);
    }
    const double dt = now() - t0;
    if (worker_mode) {
        for (int i = 0; i < threads; i++) {
            char path[320];
            snprintf(path, sizeof(path), "%s/rx.demod2client.%d.s8", scratch, i);
            FILE *f = fopen(path, "rb");
            cl[i].hash = 0xcbf29ce484222325ull;
            if (f == NULL) {
                cl[i].error = -1;
                continue;
            }
            unsigned char buf[65536];
            size_t n;
            while ((n = fread(buf, 1, sizeof(buf), f)) > 0) {
                cl[i].symbols += n;
                cl[i].hash = fnv(cl[i].hash, buf, n);
            }
            fclose(f);
            unlink(path);
        }
        rmdir(scratch);
    }
    int errors = 0;
    unsigned long long symbols = 0;
    for (int i = 0; i < threads; i++) {
        errors += cl[i].error != 0;
        symbols += cl[i].symbols;
    }
    const int classes = n_files ? n_files : 1;
    for (int k = 0; k < classes && k < threads; k++) {
        int members = 0, agree = 1;
        for (int i = k; i < threads; i += classes) {
            members++;
            agree = agree && cl[i].hash == cl[k].hash && cl[i].symbols == cl[k].symbols;
        }
        if (!quiet || !agree) {
            printf("class %d: handles %d symbols %llu fnv %016llx agree %s\n", k, members, cl[k].symbols, cl[k].hash, agree ? "yes" : "NO");
        }
    }
    uint64_t taken = 0, refused = 0;
    uint32_t peak = 0;
    sdrm_handoff_stats(-1, &taken, &refused, &peak);
    const char *hand = getenv("SDRM_HANDOFF");
    printf("%d %s x %zu buffers of %zu samples (%s%s): %.1f ms, %.1f Msamples/s, %.0f symbols per buffer, errors %d, "
           "hand-off taken %llu refused %llu peak waiting %u\n",
           threads, worker_mode ? "workers" : "handles", n_calls - n_warm, n_buf,
           (share || getenv("SDRM_SHARED_SLOTS")) ? "shared batcher" : "private batches", (hand && atoi(hand) == 0) ? ", SDRM_HANDOFF=0" : "", dt * 1e3,
           (double) threads * (double) (n_calls - n_warm) * (double) n_buf / dt / 1e6, (double) symbols / ((double) threads * (double) n_calls), errors,
           (unsigned long long) taken, (unsigned long long) refused, peak);
    if (!worker_mode) {
        for (int i = 0; i < threads; i++) fsk_demod_destroy(cl[i].demod);
    }
    return errors ? 3 : 0;
}
