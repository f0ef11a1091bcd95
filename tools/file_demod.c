/* file_demod.c -- file-source harness for the worker surface.
 * Reads a raw .cf32 recording (little-endian interleaved fp32 I,Q, no header) in buffer_size chunks exactly like the
 * reference's file source does (src/sdr/file_source.c:93-130: one fread of max_output_buffer_length samples per
 * sdr_process_rx call), pushes every chunk to a dsp_worker (sdr_worker.c:25-29) and lets the worker write
 * <base_path>/rx.demod2client.<id>.s8 (and rx.sdr2demod.<id>.cf32 with -d), the files sdr-modem itself produces for a
 * FILE destination.  With -n N the same recording feeds N workers that share one per-GPU batcher; with -g G they are
 * placed by a node front door over G batchers instead (sdrm_node_*: one per visible GPU, wrapping round when G exceeds the
 * GPUs of the box -- the one-process, many-GPU layout of INTEGRATION.md section 3b).  -o HZ applies the file source's
 * frequency offset (RxRequest.rx_offset, src/sdr/file_source.c:120-128) on the device in front of the demodulator.
 * Build: gcc -O2 -pthread tools/file_demod.c -Iinclude -Lsdr-modem_amd/csrc -lsdrmodem_hip
 *            -Wl,-rpath,'$ORIGIN/../sdr-modem_amd/csrc' -o tools/file_demod
 * Usage: file_demod [-d] [-n workers] [-g batchers] [-o rx_offset_hz] [-b buffer_size] <in.cf32> <base_path> <fs> <baud>
 *                   <deviation> <decim> <tw> <dc 0|1> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdrmodem_hip.h"

int main(int argc, char **argv) {
    int dump = 0, n_workers = 1, n_batchers = 0, a = 1;
    long long rx_offset = 0;
    uint32_t buffer_size = 4096;
    while (a < argc && argv[a][0] == '-') {
        if (strcmp(argv[a], "-d") == 0) {
            dump = 1;
        } else if (strcmp(argv[a], "-n") == 0 && a + 1 < argc) {
            n_workers = atoi(argv[++a]);
        } else if (strcmp(argv[a], "-g") == 0 && a + 1 < argc) {
            n_batchers = atoi(argv[++a]);
        } else if (strcmp(argv[a], "-o") == 0 && a + 1 < argc) {
            rx_offset = atoll(argv[++a]);
        } else if (strcmp(argv[a], "-b") == 0 && a + 1 < argc) {
            buffer_size = (uint32_t) atol(argv[++a]);
        }
        a++;
    }
    if (argc - a < 8 || n_workers < 1) {
        fprintf(stderr, "usage: %s [-d] [-n workers] [-g batchers] [-o rx_offset_hz] [-b buffer_size] in.cf32 base_path fs baud "
                        "deviation decim tw dc\n", argv[0]);
        return 2;
    }
    const char *in_path = argv[a], *base_path = argv[a + 1];
    sdrm_worker_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.rx_sampling_freq = (uint64_t) atoll(argv[a + 2]);
    cfg.demod_baud_rate = (uint32_t) atol(argv[a + 3]);
    cfg.demod_fsk_deviation = atoll(argv[a + 4]);
    cfg.demod_decimation = (uint32_t) atol(argv[a + 5]);
    cfg.demod_fsk_transition_width = (uint32_t) atol(argv[a + 6]);
    cfg.demod_fsk_use_dc_block = atoi(argv[a + 7]) != 0;
    cfg.rx_dump_file = dump != 0;
    cfg.demod_destination = 0; /* FILE */
    cfg.buffer_size = buffer_size;
    cfg.queue_size = 16;
    cfg.rx_file_source = true; /* blocking queue: a file must not lose data (src/dsp_worker.c:176-179) */
    cfg.base_path = base_path;
    cfg.rx_offset_hz = rx_offset;

    FILE *in = fopen(in_path, "rb");
    if (in == NULL) {
        fprintf(stderr, "<3>unable to open file for input: %s\n", in_path);
        return 1;
    }
    sdrm_batcher *bt = NULL;
    sdrm_node *node = NULL;
    if (n_batchers > 0) {
        /* one process, several devices: the node places every worker (least-loaded batcher first) */
        sdrm_node_config nc;
        memset(&nc, 0, sizeof(nc));
        nc.n_batchers = (size_t) n_batchers;
        nc.slots_per_batcher = (size_t) ((n_workers + n_batchers - 1) / n_batchers);
        nc.geometry.sampling_freq = cfg.rx_sampling_freq;
        nc.geometry.baud_rate = cfg.demod_baud_rate;
        nc.geometry.deviation = cfg.demod_fsk_deviation;
        nc.geometry.decimation = (uint8_t) cfg.demod_decimation;
        nc.geometry.transition_width = cfg.demod_fsk_transition_width;
        nc.geometry.use_dc_block = cfg.demod_fsk_use_dc_block;
        nc.geometry.max_input_buffer_length = buffer_size;
        nc.batcher.slots = 6;
        nc.batcher.max_wait_us = 20000;
        nc.batcher.blocking = true;
        int code = sdrm_node_create(&nc, &node);
        if (code != 0) {
            fprintf(stderr, "<3>unable to create the node: %d\n", code);
            return 1;
        }
    } else if (n_workers > 1) {
        sdrm_fsk_config *fc = calloc((size_t) n_workers, sizeof(*fc));
        for (int i = 0; i < n_workers; i++) {
            fc[i].sampling_freq = cfg.rx_sampling_freq;
            fc[i].baud_rate = cfg.demod_baud_rate;
            fc[i].deviation = cfg.demod_fsk_deviation;
            fc[i].decimation = (uint8_t) cfg.demod_decimation;
            fc[i].transition_width = cfg.demod_fsk_transition_width;
            fc[i].use_dc_block = cfg.demod_fsk_use_dc_block;
            fc[i].max_input_buffer_length = buffer_size;
        }
        sdrm_batcher_config bc = {6, 20000, true};
        int code = sdrm_batcher_create(fc, (size_t) n_workers, -1, &bc, &bt);
        free(fc);
        if (code != 0) {
            fprintf(stderr, "<3>unable to create the batcher: %d\n", code);
            return 1;
        }
    }
    dsp_worker **w = calloc((size_t) n_workers, sizeof(*w));
    for (int i = 0; i < n_workers; i++) {
        cfg.batcher = bt;
        cfg.batcher_channel = (size_t) i;
        cfg.node = node;
        cfg.source_id = 1; /* one recording = one source */
        int code = dsp_worker_create((uint32_t) i, -1, &cfg, &w[i]);
        if (code != 0) {
            fprintf(stderr, "<3>unable to create worker %d: %d\n", i, code);
            return 1;
        }
        cfg.rx_dump_file = false; /* one copy of the input is enough */
    }
    sdrm_cf32 *buf = malloc(sizeof(sdrm_cf32) * buffer_size);
    size_t total = 0;
    for (;;) {
        size_t got = fread(buf, sizeof(sdrm_cf32), buffer_size, in); /* file_source.c:101 */
        if (got == 0) {
            break;
        }
        total += got;
        for (int i = 0; i < n_workers; i++) {
            dsp_worker_put(buf, got, w[i]); /* sdr_worker.c:25-29 */
        }
    }
    for (int i = 0; i < n_workers; i++) {
        dsp_worker_destroy(w[i]); /* poison pill after the queued buffers, joins the thread, closes the files */
    }
    if (node != NULL) {
        for (size_t i = 0; i < sdrm_node_batchers(node); i++) {
            sdrm_node_stat st;
            if (sdrm_node_stat_read(node, i, &st) == 0) {
                fprintf(stderr, "batcher %zu on device %d served %llu worker(s)\n", i, st.device, (unsigned long long) st.attached);
            }
        }
        sdrm_node_destroy(node);
    }
    if (bt != NULL) {
        sdrm_batcher_destroy(bt);
    }
    fclose(in);
    free(buf);
    free(w);
    fprintf(stderr, "%zu samples demodulated by %d worker(s)\n", total, n_workers);
    return 0;
}
