/* batcher_bench.c -- end-to-end rate of the worker surface: P producer threads copy IQ buffers of C clients into the
 * batcher (the memcpy of queue_put), the batcher runs one batched call per round, Q consumer threads take the soft bits.
 * Everything the reference's per-client path does between the SDR callback and the socket write is inside the timing:
 * host memcpy into pinned memory, host->device copy, kernels, device->host copy, hand-over to the consumer thread.
 * Build: gcc -O2 -pthread tools/batcher_bench.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip -Wl,-rpath,'$ORIGIN/../sdr-modem_amd/csrc' -lm -o tools/batcher_bench
 * Run:   tools/batcher_bench [channels] [chunk] [rounds] [producer threads] [consumer threads] [batchers]
 * batchers > 0: the clients are placed by a node front door (sdrm_node_*) over that many batchers -- one per visible GPU,
 * wrapping round on a box with fewer: the one-process, many-GPU layout of INTEGRATION.md section 3c.
 * Environment (node mode): BB_SLOTS=<slots per batcher> (default: just enough for the clients) -- a server's batcher is sized for
 * its busiest hour, most slots wait for a client; BB_KINDS=3 -- the clients are BASELINE configs[4]'s three kinds in turn
 * (48 kHz / 9600 baud, 240 kHz / 19200 baud / decimation 5, 48 kHz / 1200 baud / decimation 8) instead of the first alone. */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sdrmodem_hip.h"

static sdrm_batcher *bt;
static sdrm_node *node;
static sdrm_node_slot *slots; /* node mode: where each client was placed */
static int n_batchers = 0;
static size_t n_ch = 256, chunk = 131072, rounds = 16;
static int n_prod = 8, n_cons = 4;
static float *iq; /* one synthetic FM buffer, shared read-only */
static unsigned long long symbols;
static pthread_mutex_t sym_lock = PTHREAD_MUTEX_INITIALIZER;

static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double) t.tv_sec + 1e-9 * (double) t.tv_nsec;
}

static void *producer(void *arg) {
    const int id = (int) (size_t) arg;
    const size_t lo = n_ch * (size_t) id / (size_t) n_prod, hi = n_ch * (size_t) (id + 1) / (size_t) n_prod;
    for (size_t k = 0; k < rounds; k++) {
        for (size_t c = lo; c < hi; c++) {
            if (slots != NULL) {
                sdrm_batcher_put(slots[c].batcher, slots[c].channel, (const sdrm_cf32 *) iq, chunk);
            } else {
                sdrm_batcher_put(bt, c, (const sdrm_cf32 *) iq, chunk);
            }
        }
    }
    return NULL;
}

static void *consumer(void *arg) {
    const int id = (int) (size_t) arg;
    const size_t lo = n_ch * (size_t) id / (size_t) n_cons, hi = n_ch * (size_t) (id + 1) / (size_t) n_cons;
    unsigned long long mine = 0;
    for (size_t k = 0; k < rounds; k++) {
        for (size_t c = lo; c < hi; c++) {
            int8_t *soft = NULL;
            size_t n = 0;
            sdrm_batcher *b = slots != NULL ? slots[c].batcher : bt;
            const size_t ch = slots != NULL ? slots[c].channel : c;
            sdrm_batcher_take(b, ch, &soft, &n);
            if (soft == NULL) {
                fprintf(stderr, "unexpected end of stream on channel %zu\n", c);
                return NULL;
            }
            mine += n;
            sdrm_batcher_complete(b, ch);
        }
    }
    pthread_mutex_lock(&sym_lock);
    symbols += mine;
    pthread_mutex_unlock(&sym_lock);
    return NULL;
}

int main(int argc, char **argv) {
    if (argc > 1) n_ch = (size_t) atol(argv[1]);
    if (argc > 2) chunk = (size_t) atol(argv[2]);
    if (argc > 3) rounds = (size_t) atol(argv[3]);
    if (argc > 4) n_prod = atoi(argv[4]);
    if (argc > 5) n_cons = atoi(argv[5]);
    if (argc > 6) n_batchers = atoi(argv[6]);
    /* 9600-baud MSK-like test signal at 48 kHz: random +-1 symbols, 5 samples each, deviation 2400 Hz */
    iq = malloc(sizeof(float) * 2 * chunk);
    double ph = 0.0;
    unsigned lfsr = 0xACE1u;
    int bit = 1;
    for (size_t i = 0; i < chunk; i++) {
        if (i % 5 == 0) {
            lfsr = (lfsr >> 1) ^ (-(lfsr & 1u) & 0xB400u);
            bit = (lfsr & 1u) ? 1 : -1;
        }
        ph += 2.0 * M_PI * 2400.0 * bit / 48000.0;
        iq[2 * i] = (float) cos(ph);
        iq[2 * i + 1] = (float) sin(ph);
    }
    sdrm_fsk_config *cfg = calloc(n_ch, sizeof(*cfg));
    for (size_t c = 0; c < n_ch; c++) {
        cfg[c].sampling_freq = 48000;
        cfg[c].baud_rate = 9600;
        cfg[c].deviation = 5000;
        cfg[c].decimation = 1;
        cfg[c].transition_width = 2000;
        cfg[c].use_dc_block = true;
        cfg[c].max_input_buffer_length = (uint32_t) chunk;
        if (getenv("BB_KINDS") != NULL && atoi(getenv("BB_KINDS")) == 3 && c % 3 == 1) {
            cfg[c].sampling_freq = 240000;
            cfg[c].baud_rate = 19200;
            cfg[c].decimation = 5;
        } else if (getenv("BB_KINDS") != NULL && atoi(getenv("BB_KINDS")) == 3 && c % 3 == 2) {
            cfg[c].baud_rate = 1200;
            cfg[c].decimation = 8;
        }
    }
    sdrm_batcher_config bc = {6, 100000, true};
    int code = 0;
    if (n_batchers > 0) {
        sdrm_node_config nc;
        memset(&nc, 0, sizeof(nc));
        nc.n_batchers = (size_t) n_batchers;
        nc.slots_per_batcher = (n_ch + (size_t) n_batchers - 1) / (size_t) n_batchers;
        if (getenv("BB_SLOTS") != NULL && (size_t) atol(getenv("BB_SLOTS")) > nc.slots_per_batcher) {
            nc.slots_per_batcher = (size_t) atol(getenv("BB_SLOTS"));
        }
        nc.geometry = cfg[0];
        nc.batcher = bc;
        code = sdrm_node_create(&nc, &node);
        if (code != 0) {
            fprintf(stderr, "sdrm_node_create failed: %d\n", code);
            return 1;
        }
        slots = calloc(n_ch, sizeof(*slots));
        for (size_t c = 0; c < n_ch; c++) {
            if (sdrm_node_attach(node, &cfg[c], 1 + c % 8, &slots[c]) != 0 ||
                sdrm_batcher_reset_channel(slots[c].batcher, slots[c].channel, &cfg[c]) != 0) {
                fprintf(stderr, "client %zu could not be placed\n", c);
                return 1;
            }
        }
    } else {
        code = sdrm_batcher_create(cfg, n_ch, -1, &bc, &bt);
        if (code != 0) {
            fprintf(stderr, "sdrm_batcher_create failed: %d\n", code);
            return 1;
        }
    }
    pthread_t *tp = calloc((size_t) n_prod, sizeof(pthread_t)), *tc = calloc((size_t) n_cons, sizeof(pthread_t));
    /* warm-up: two rounds */
    size_t keep = rounds;
    rounds = 2;
    for (int i = 0; i < n_prod; i++) pthread_create(&tp[i], NULL, producer, (void *) (size_t) i);
    for (int i = 0; i < n_cons; i++) pthread_create(&tc[i], NULL, consumer, (void *) (size_t) i);
    for (int i = 0; i < n_prod; i++) pthread_join(tp[i], NULL);
    for (int i = 0; i < n_cons; i++) pthread_join(tc[i], NULL);
    rounds = keep;
    symbols = 0;
    uint64_t r0 = 0;
    if (node != NULL) {
        for (size_t i = 0; i < sdrm_node_batchers(node); i++) {
            sdrm_node_stat st;
            sdrm_node_stat_read(node, i, &st);
            r0 += sdrm_batcher_rounds(st.batcher);
        }
    } else {
        r0 = sdrm_batcher_rounds(bt);
    }
    const double t0 = now();
    for (int i = 0; i < n_prod; i++) pthread_create(&tp[i], NULL, producer, (void *) (size_t) i);
    for (int i = 0; i < n_cons; i++) pthread_create(&tc[i], NULL, consumer, (void *) (size_t) i);
    for (int i = 0; i < n_prod; i++) pthread_join(tp[i], NULL);
    for (int i = 0; i < n_cons; i++) pthread_join(tc[i], NULL);
    const double dt = now() - t0;
    const double samples = (double) n_ch * (double) chunk * (double) rounds;
    uint64_t r1 = 0;
    if (node != NULL) {
        for (size_t i = 0; i < sdrm_node_batchers(node); i++) {
            sdrm_node_stat st;
            sdrm_node_stat_read(node, i, &st);
            r1 += sdrm_batcher_rounds(st.batcher);
            printf("  batcher %zu on device %d: %zu clients\n", i, st.device, st.clients);
        }
    } else {
        r1 = sdrm_batcher_rounds(bt);
    }
    printf("%s end to end: %zu clients x %zu samples x %zu buffers, %d producer / %d consumer threads: %.1f ms per round, "
           "%.0f Msamples/s, %llu device calls, %.0f symbols per buffer\n", node != NULL ? "node" : "batcher",
           n_ch, chunk, rounds, n_prod, n_cons, dt / (double) rounds * 1e3, samples / dt / 1e6,
           (unsigned long long) (r1 - r0), (double) symbols / ((double) n_ch * (double) rounds));
    if (node != NULL) {
        for (size_t c = 0; c < n_ch; c++) {
            sdrm_batcher_interrupt(slots[c].batcher, slots[c].channel);
            sdrm_node_detach(node, &slots[c]);
        }
        sdrm_node_destroy(node);
    } else {
        sdrm_batcher_destroy(bt);
    }
    return 0;
}
