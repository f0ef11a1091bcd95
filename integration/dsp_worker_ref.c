/*
 * dsp_worker_ref.c -- dsp_worker_create() with the REFERENCE'S OWN SIGNATURE (/root/reference/src/dsp_worker.h:22) on top
 * of libsdrmodem_hip.so.  This is the one file a maintainer adds to sdr-modem in place of src/dsp_worker.c (together with
 * linking the library): tcp_server.c:659, sdr_worker.c:25-29,46,52 and the linked_list
 * callbacks keep calling dsp_worker_create / _put / _shutdown / _find_by_id / _destroy exactly as they do today.
 *
 * The library's constructor takes a plain struct (sdrm_worker_config: no protobuf-c, no libiio in its headers) and is
 * exported under two names; this file calls it as sdrm_dsp_worker_create, because the dsp_worker_create defined HERE is the
 * one the whole program sees.
 *
 * Build inside sdr-modem:   cc -DSDRM_REF_HEADERS -Isrc -I<repo>/include -c integration/dsp_worker_ref.c
 * Build for the test-suite: cc -I<repo>/include -c integration/dsp_worker_ref.c      (ref_fields.h restates the two types)
 */
#include <errno.h>
#include <stdio.h>

#ifdef SDRM_REF_HEADERS
#include "api.pb-c.h"
#include "server_config.h"
#else
#include "ref_fields.h"
#endif
#define SDRM_REFERENCE_DSP_WORKER_CREATE /* this file defines dsp_worker_create with the reference's parameter list */
#include "sdrmodem_hip.h"

/* Optional: workers become clients of ONE per-GPU batcher instead of owning a private
 * demodulator each.  `next_channel` hands out the batcher's channels; a server that recycles slots passes its own. */
static sdrm_batcher *g_batcher = NULL;
static size_t (*g_next_channel)(void *user) = NULL;
static void *g_next_channel_user = NULL;
void sdrm_ref_attach_batcher(sdrm_batcher *batcher, size_t (*next_channel)(void *user), void *user) {
    g_batcher = batcher;
    g_next_channel = next_channel;
    g_next_channel_user = user;
}

/* Optional: the server has several GPUs -- one node handle (sdrm_node_create: a batcher per device) places every new client on
 * the least-loaded healthy device; the worker gives its slot back when linked_list's destructor calls dsp_worker_destroy.
 * Clients that share an SDR source in the reference (same centre frequency and offset, sdr_worker.c:83-95) carry the same
 * source id, so the node keeps them on one device while that balances.  INTEGRATION.md section 3b. */
static sdrm_node *g_node = NULL;
void sdrm_ref_attach_node(sdrm_node *node) {
    g_node = node;
}

/* Optional: Doppler pre-correction.  The reference builds its SGP4 predictor from req->doppler (TLE + ground station,
 * src/dsp_worker.c:120-136, src/dsp/doppler.c:31-42); that orbit model stays on the host side of the boundary, so the
 * integrator supplies a factory that turns the request into "shift in Hz for second k of the pass". */
static int (*g_doppler_factory)(const struct RxRequest *req, const struct server_config *config, sdrm_doppler_shift_fn *fn,
                                void **user) = NULL;
static void (*g_doppler_release)(void *user) = NULL;
void sdrm_ref_set_doppler_factory(int (*factory)(const struct RxRequest *, const struct server_config *, sdrm_doppler_shift_fn *,
                                                 void **)) {
    g_doppler_factory = factory;
}
/* ... and what frees the state such a factory hands out, when the worker is destroyed (may stay NULL).  The shipped pair:
 * integration/doppler_factory_ref.c -- sdrm_ref_doppler_factory / sdrm_ref_doppler_close, on the reference's own src/sgpsdp */
void sdrm_ref_set_doppler_release(void (*release)(void *user)) {
    g_doppler_release = release;
}

int dsp_worker_create(uint32_t id, int client_socket, struct server_config *server_config, struct RxRequest *req,
                      dsp_worker **worker) {
    if (server_config == NULL || req == NULL || worker == NULL) {
        return -1;
    }
    if (req->demod_type != MODEM_TYPE__GMSK || req->fsk_settings == NULL) {
        /* the reference creates no demodulator for another modem type and then dereferences it (dsp_worker.c:138-144,75);
         * tcp_server.c:123-169 rejects such requests before they get here */
        fprintf(stderr, "<3>[%d] unable to create demodulator\n", (int) id);
        return -1;
    }
    sdrm_worker_config c = {
        .rx_sampling_freq = req->rx_sampling_freq,                                   /* dsp_worker.c:140 */
        .demod_baud_rate = req->demod_baud_rate,
        .demod_fsk_deviation = req->fsk_settings->demod_fsk_deviation,
        .demod_decimation = req->demod_decimation,                                   /* cast to uint8_t there, :141 */
        .demod_fsk_transition_width = req->fsk_settings->demod_fsk_transition_width,
        .demod_fsk_use_dc_block = req->fsk_settings->demod_fsk_use_dc_block != 0,
        .rx_dump_file = req->rx_dump_file != 0,                                      /* dsp_worker.c:152 */
        .demod_destination = (int) req->demod_destination,                           /* FILE 0, SOCKET 1, BOTH 2 */
        .buffer_size = server_config->buffer_size,                                   /* server_config.h */
        .queue_size = server_config->queue_size,
        .rx_file_source = server_config->rx_sdr_type == RX_SDR_TYPE_FILE,            /* dsp_worker.c:178 */
        .base_path = server_config->base_path,
    };
    if (req->doppler != NULL) {
        if (g_doppler_factory == NULL) {
            fprintf(stderr, "<3>[%d] unable to create doppler correction block\n", (int) id);  /* dsp_worker.c:131 */
            return -ENOTSUP;
        }
        int code = g_doppler_factory(req, server_config, &c.doppler_shift, &c.doppler_user);
        if (code != 0) {
            fprintf(stderr, "<3>[%d] unable to create doppler correction block\n", (int) id);
            return code;
        }
        c.doppler_release = g_doppler_release;
    }
    if (g_batcher != NULL) {
        c.batcher = g_batcher;
        c.batcher_channel = g_next_channel != NULL ? g_next_channel(g_next_channel_user) : 0;
    } else if (g_node != NULL) {
        c.node = g_node;
        /* one id per SDR source as sdr_worker_find_closest tells them apart (sdr_worker.c:88-91); never 0 ("no source").
         * rx_offset itself is NOT applied here: the reference's sources apply it upstream of dsp_worker_put
         * (tcp_server.c:438-458, file_source.c:120-128) */
        c.source_id = (req->rx_center_freq * 0x9E3779B97F4A7C15ull) ^ ((uint64_t) req->rx_offset * 0xC2B2AE3D27D4EB4Full) ^ 1ull;
        if (c.source_id == 0) {
            c.source_id = 1;
        }
    }
    return sdrm_dsp_worker_create(id, client_socket, &c, worker);
}
