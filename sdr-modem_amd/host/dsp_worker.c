/*
 * dsp_worker.c -- per-client DSP thread: pulls IQ buffers from its queue, optionally dumps them, demodulates
 * them on the GPU through fsk_demod_process(), and pushes the soft bits to a file and/or the client socket.
 *
 * Mirror of the reference's src/dsp_worker.{h,c} (public API src/dsp_worker.h:14-22, thread body
 * src/dsp_worker.c:44-106, construction :108-197, teardown :199-227) with the protobuf RxRequest and the
 * libconfig server_config replaced by the plain sdrm_worker_config (those headers need protobuf-c / libiio,
 * which are outside this path).  Same call order, same file names (rx.sdr2demod.<id>.cf32,
 * rx.demod2client.<id>.s8), same error returns and "<3>" messages.  Doppler pre-correction (src/dsp/doppler.c) is
 * driven by a per-second shift callback (the SGP4 orbit model stays with the caller) and runs on the GPU in front of
 * the demodulator.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/sdrmodem_hip.h"

enum { DEST_FILE = 0, DEST_SOCKET = 1, DEST_BOTH = 2 }; /* api.proto DemodDestination */

struct dsp_worker_t {
    uint32_t id;
    int client_socket;
    fsk_demod *demod;
    sdrm_batch *corrected; /* batch of one channel, used instead of `demod` when Doppler pre-correction is on */
    sdrm_doppler *doppler;
    sdrm_batcher *batcher; /* shared per-GPU batcher (borrowed) used instead of demod/corrected + inbox */
    size_t channel;
    sdrm_node *node;       /* the node that placed this client on `batcher` (borrowed), NULL otherwise */
    sdrm_node_slot slot;   /* ... and the slot to give back */
    int64_t rx_offset_hz;  /* the file source's frequency offset, what the built-in shift callback returns */
    void (*doppler_release)(void *);  /* frees the caller's shift callback state when the worker goes */
    void *doppler_user;
    bool dump_failed;      /* batcher path: the IQ dump could not be written, the client has been ended */
    queue *inbox;
    pthread_t thread;
    bool thread_started;
    FILE *iq_dump;
    FILE *soft_dump;
    int destination;
};

bool dsp_worker_find_by_id(void *id, void *data) {
    const dsp_worker *w = (const dsp_worker *) data;
    return w->id == *(uint32_t *) id;
}

void dsp_worker_put(sdrm_cf32 *output, size_t output_len, dsp_worker *worker) {
    if (worker->batcher != NULL) {
        /* the IQ goes straight into the batcher's pinned arena; the dump the reference writes from its DSP thread
         * (src/dsp_worker.c:59-64) is written here, by the source thread, because no other thread sees the samples */
        if (worker->dump_failed) {
            return;
        }
        if (worker->iq_dump != NULL && fwrite(output, sizeof(sdrm_cf32), output_len, worker->iq_dump) < output_len) {
            /* the reference's DSP thread stops at this point (src/dsp_worker.c:56-64: message, break): the client is ended --
             * the buffer that could not be dumped is not demodulated, what was put before it is still delivered */
            fprintf(stderr, "<3>[%d] unable to write sdr data\n", worker->id);
            worker->dump_failed = true;
            sdrm_batcher_interrupt(worker->batcher, worker->channel);
            return;
        }
        sdrm_batcher_put(worker->batcher, worker->channel, output, output_len);
        return;
    }
    queue_put(output, output_len, worker->inbox);
}

void dsp_worker_shutdown(void *arg, void *data) {
    (void) arg;
    dsp_worker *w = (dsp_worker *) data;
    if (w->batcher != NULL) {
        sdrm_batcher_interrupt(w->batcher, w->channel);
        return;
    }
    interrupt_waiting_the_data(w->inbox);
}

/* reference src/tcp_utils.c:7-17 */
static int write_fully(const uint8_t *bytes, size_t n, int fd) {
    size_t done = 0;
    while (done < n) {
        ssize_t w = write(fd, bytes + done, n - done);
        if (w < 0) {
            return -1;
        }
        done += (size_t) w;
    }
    return 0;
}

/* thread body when the worker sits on a shared batcher: the soft bits arrive already demodulated */
static void *batched_main(void *arg) {
    dsp_worker *w = (dsp_worker *) arg;
    int gave_up = 0;
    fprintf(stdout, "[%d] dsp_worker is starting\n", w->id);
    for (;;) {
        int8_t *soft = NULL;
        size_t soft_len = 0;
        sdrm_batcher_take(w->batcher, w->channel, &soft, &soft_len);
        if (soft == NULL) {
            /* poison pill (everything put before it has been delivered) -- or the device behind the batcher failed, which
             * ends every client at once: say so, as the private-handle path does through sdrm_fsk_demod_error */
            const int dev = sdrm_batcher_error(w->batcher);
            if (dev != 0) {
                fprintf(stderr, "<3>[%d] the demodulator's device call failed (%d); the client is ended\n", w->id, dev);
            }
            break;
        }
        if (w->soft_dump != NULL && fwrite(soft, sizeof(int8_t), soft_len, w->soft_dump) < soft_len) {
            sdrm_batcher_complete(w->batcher, w->channel);
            fprintf(stderr, "<3>[%d] unable to write demod data\n", w->id);
            gave_up = 1;
            break;
        }
        int code = 0;
        if (w->destination == DEST_SOCKET || w->destination == DEST_BOTH) {
            code = write_fully((const uint8_t *) soft, soft_len, w->client_socket);
        }
        sdrm_batcher_complete(w->batcher, w->channel);
        if (code != 0) {
            gave_up = 1;
            break;
        }
    }
    if (gave_up) {
        /* nobody takes this client's results any more: let the shared rounds go (the reference's worker leaves its own
         * queue to fill up, src/dsp_worker.c:56-64,83-101; here the rounds belong to every client).  Not after the
         * poison pill: that path has drained the channel, and the slot may already have been handed to the next client
         * (sdrm_batcher_reset_channel) by the time this thread is joined -- a late abandon would close ITS channel. */
        sdrm_batcher_abandon(w->batcher, w->channel);
    }
    printf("[%d] dsp_worker stopped\n", w->id);
    return NULL;
}

static void *worker_main(void *arg) {
    dsp_worker *w = (dsp_worker *) arg;
    fprintf(stdout, "[%d] dsp_worker is starting\n", w->id);
    for (;;) {
        sdrm_cf32 *iq = NULL;
        size_t iq_len = 0;
        take_buffer_for_processing(&iq, &iq_len, w->inbox);
        if (iq == NULL) {
            break; /* poison pill */
        }
        if (w->iq_dump != NULL && fwrite(iq, sizeof(sdrm_cf32), iq_len, w->iq_dump) < iq_len) {
            complete_buffer_processing(w->inbox);
            fprintf(stderr, "<3>[%d] unable to write sdr data\n", w->id);
            break;
        }
        int8_t *soft = NULL;
        size_t soft_len = 0;
        if (w->corrected != NULL) {
            /* reference order: [the file source's offset oscillator, file_source.c:120-128 -- the batch's pre-offset,] then
             * doppler_process_rx, then fsk_demod_process (src/dsp_worker.c:65-76); here one call */
            sdrm_nco_segment segs[64];
            size_t n_segs = w->doppler != NULL ? sdrm_doppler_plan(w->doppler, 0, iq_len, segs, 64) : 0;
            const sdrm_cf32 *ins[1] = {iq};
            size_t lens[1] = {iq_len};
            int8_t *outs[1] = {NULL};
            size_t olens[1] = {0};
            const int failed = n_segs > 0 ? sdrm_batch_process_nco(w->corrected, ins, lens, segs, n_segs, outs, olens)
                                          : sdrm_batch_process(w->corrected, ins, lens, outs, olens);
            if (failed != 0) {
                complete_buffer_processing(w->inbox);
                fprintf(stderr, "<3>[%d] demodulation failed on the device\n", w->id);
                break;
            }
            soft = outs[0];
            soft_len = olens[0];
        } else if (w->demod != NULL) {
            fsk_demod_process(iq, iq_len, &soft, &soft_len, w->demod);
            if (sdrm_fsk_demod_error(w->demod) != 0) {
                /* the device path failed under this client's handle: end this client, like a socket or disk error
                 * does in the reference (src/dsp_worker.c:56-64, 83-101); the others keep running */
                complete_buffer_processing(w->inbox);
                fprintf(stderr, "<3>[%d] demodulation failed on the device\n", w->id);
                break;
            }
        }
        if (soft == NULL) {
            complete_buffer_processing(w->inbox);
            continue;
        }
        if (w->soft_dump != NULL && fwrite(soft, sizeof(int8_t), soft_len, w->soft_dump) < soft_len) {
            complete_buffer_processing(w->inbox);
            fprintf(stderr, "<3>[%d] unable to write demod data\n", w->id);
            break;
        }
        int code = 0;
        if (w->destination == DEST_SOCKET || w->destination == DEST_BOTH) {
            code = write_fully((const uint8_t *) soft, soft_len, w->client_socket);
        }
        complete_buffer_processing(w->inbox);
        if (code != 0) {
            break;
        }
    }
    printf("[%d] dsp_worker stopped\n", w->id);
    return NULL;
}

int dsp_worker_create(uint32_t id, int client_socket, const sdrm_worker_config *cfg, dsp_worker **result) {
    return sdrm_dsp_worker_create(id, client_socket, cfg, result);
}

/* a worker that never started: what dsp_worker_destroy would free of it */
static void discard(dsp_worker *w) {
    if (w->doppler_release != NULL) {
        w->doppler_release(w->doppler_user);
    }
    free(w);
}

int sdrm_dsp_worker_create(uint32_t id, int client_socket, const sdrm_worker_config *cfg, dsp_worker **result) {
    dsp_worker *w = calloc(1, sizeof(*w));
    if (w == NULL) {
        return -ENOMEM;
    }
    w->id = id;
    w->client_socket = client_socket;
    w->rx_offset_hz = cfg->rx_offset_hz;
    w->doppler_release = cfg->doppler_release;
    w->doppler_user = cfg->doppler_user;
    int code = 0;
    /* The file source's offset (file_source.c:120-128: sig_source_multiply(freq_offset, ...), one oscillator for the life of the
     * stream) is the batch's pre-offset; the Doppler correction (src/dsp_worker.c:65-71) runs behind it with an oscillator of
     * its own -- in series, every sample rounded to fp32 in between, as the reference has them. */
    sdrm_doppler_shift_fn shift_fn = cfg->doppler_shift;
    void *shift_user = cfg->doppler_user;
    sdrm_fsk_config fc = {cfg->rx_sampling_freq, cfg->demod_baud_rate, cfg->demod_fsk_deviation,
                          (uint8_t) cfg->demod_decimation, cfg->demod_fsk_transition_width,
                          cfg->demod_fsk_use_dc_block, cfg->buffer_size};
    if (cfg->batcher != NULL || cfg->node != NULL) {
        if (cfg->batcher != NULL) {
            if (cfg->batcher_channel >= sdrm_batcher_channels(cfg->batcher)) {
                fprintf(stderr, "<3>[%d] batcher has no channel %zu\n", w->id, cfg->batcher_channel);
                discard(w);
                return -1;
            }
            w->batcher = cfg->batcher;
            w->channel = cfg->batcher_channel;
            /* the slot starts a new stream with this client's parameters (fsk_demod_create's part, src/dsp_worker.c:138-144) */
            code = sdrm_batcher_reset_channel_offset(w->batcher, w->channel, &fc, cfg->rx_offset_hz);
        } else {
            /* the node places the client (least-loaded healthy device); a device that fails between the placement and the
             * slot's reset is skipped next time round, so at most one attempt per batcher */
            const size_t attempts = sdrm_node_batchers(cfg->node) + 1;
            code = -EBUSY;
            for (size_t k = 0; k < attempts; k++) {
                code = sdrm_node_attach(cfg->node, &fc, cfg->source_id, &w->slot);
                if (code != 0) {
                    break;
                }
                code = sdrm_batcher_reset_channel_offset(w->slot.batcher, w->slot.channel, &fc, cfg->rx_offset_hz);
                if (code == 0) {
                    w->node = cfg->node;
                    w->batcher = w->slot.batcher;
                    w->channel = w->slot.channel;
                    break;
                }
                const int dead = sdrm_batcher_error(w->slot.batcher);
                sdrm_node_detach(cfg->node, &w->slot);
                if (dead == 0) {
                    break; /* the client's own parameters were refused (-1, -ENOTSUP): another device would refuse them too */
                }
            }
        }
        if (code != 0) {
            fprintf(stderr, "<3>[%d] unable to create demodulator\n", w->id);
            discard(w);
            return code;
        }
        if (shift_fn != NULL) {
            code = sdrm_doppler_create(cfg->rx_sampling_freq, shift_fn, shift_user, &w->doppler);
            if (code == 0) {
                code = sdrm_batcher_set_doppler(w->batcher, w->channel, w->doppler);
            }
        }
    } else if (shift_fn != NULL || cfg->rx_offset_hz != 0) {
        if (shift_fn != NULL) {
            code = sdrm_doppler_create(cfg->rx_sampling_freq, shift_fn, shift_user, &w->doppler);
            if (code != 0) {
                fprintf(stderr, "<3>[%d] unable to create doppler correction block\n", w->id);
                dsp_worker_destroy(w);
                return code;
            }
        }
        code = sdrm_batch_create(&fc, 1, -1, 0, &w->corrected);
        if (code == 0 && cfg->rx_offset_hz != 0) {
            code = sdrm_batch_set_pre_offset(w->corrected, 0, cfg->rx_offset_hz);
        }
    } else {
        code = fsk_demod_create(cfg->rx_sampling_freq, cfg->demod_baud_rate, cfg->demod_fsk_deviation,
                                (uint8_t) cfg->demod_decimation, cfg->demod_fsk_transition_width,
                                cfg->demod_fsk_use_dc_block, cfg->buffer_size, &w->demod);
    }
    if (code != 0) {
        fprintf(stderr, "<3>[%d] unable to create demodulator\n", w->id);
        dsp_worker_destroy(w);
        return code;
    }
    char path[4096];
    if (cfg->rx_dump_file) {
        snprintf(path, sizeof(path), "%s/rx.sdr2demod.%d.cf32", cfg->base_path, id);
        w->iq_dump = fopen(path, "wb");
        if (w->iq_dump == NULL) {
            fprintf(stderr, "<3>[%d] unable to open file for sdr input: %s\n", w->id, path);
            dsp_worker_destroy(w);
            return -1;
        }
    }
    w->destination = cfg->demod_destination;
    if (cfg->demod_destination == DEST_FILE || cfg->demod_destination == DEST_BOTH) {
        snprintf(path, sizeof(path), "%s/rx.demod2client.%d.s8", cfg->base_path, id);
        w->soft_dump = fopen(path, "wb");
        if (w->soft_dump == NULL) {
            fprintf(stderr, "<3>[%d] unable to open file for demod output: %s\n", w->id, path);
            dsp_worker_destroy(w);
            return -1;
        }
    }
    /* a file source must not lose data => blocking queue (src/dsp_worker.c:176-179) */
    if (w->batcher == NULL) {
        code = create_queue(cfg->buffer_size, cfg->queue_size, cfg->rx_file_source, &w->inbox);
        if (code != 0) {
            dsp_worker_destroy(w);
            return code;
        }
    }
    if (pthread_create(&w->thread, NULL, w->batcher != NULL ? batched_main : worker_main, w) != 0) {
        dsp_worker_destroy(w);
        return -1;
    }
    w->thread_started = true;
    *result = w;
    return 0;
}

void dsp_worker_destroy(void *data) {
    if (data == NULL) {
        return;
    }
    dsp_worker *w = (dsp_worker *) data;
    fprintf(stdout, "[%d] dsp_worker is stopping\n", w->id);
    if (w->inbox != NULL) {
        interrupt_waiting_the_data(w->inbox);
    }
    if (w->batcher != NULL) {
        sdrm_batcher_interrupt(w->batcher, w->channel);
    }
    if (w->thread_started) {
        pthread_join(w->thread, NULL);
    }
    if (w->inbox != NULL) {
        destroy_queue(w->inbox);
    }
    if (w->iq_dump != NULL) {
        fclose(w->iq_dump);
    }
    if (w->soft_dump != NULL) {
        fclose(w->soft_dump);
    }
    if (w->demod != NULL) {
        fsk_demod_destroy(w->demod);
    }
    if (w->corrected != NULL) {
        sdrm_batch_destroy(w->corrected);
    }
    if (w->batcher != NULL && w->doppler != NULL) {
        sdrm_batcher_set_doppler(w->batcher, w->channel, NULL);
    }
    if (w->doppler != NULL) {
        sdrm_doppler_destroy(w->doppler);
    }
    if (w->node != NULL) {
        sdrm_node_detach(w->node, &w->slot); /* the slot serves the node's next client, on whichever device that is */
    }
    if (w->doppler_release != NULL) {
        w->doppler_release(w->doppler_user);
    }
    free(w);
}
