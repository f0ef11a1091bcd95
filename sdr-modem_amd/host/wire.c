/*
 * wire.c -- the slice of sdr-modem's wire protocol that touches the RX hot path: the
 * 6-byte message header, the success/failure Response the server sends before soft bits start to flow on the same socket,
 * and the fields of an RxRequest the DSP worker reads.  Enough for a caller to put the GPU worker behind a socket the way
 * the reference's tcp_server does; the server itself (accept loop, devices, TX) stays out of scope.
 *
 * reference: header src/api.h:23-27 (packed: u8 version, u8 type, u32 length in network order); message types :8-15;
 * Response api.proto:69-72 written by src/api_utils.c:82-108; RxRequest api.proto:35-49, its fields read at
 * src/dsp_worker.c:120-163; soft bits are written raw after the Response (src/tcp_server.c:677, src/dsp_worker.c:93-95).
 * protobuf-c is not in this image: the two messages are small enough to encode / decode by hand (proto2 varints and
 * length-delimited sub-messages; unknown fields are skipped as the format prescribes).
 */
#include <arpa/inet.h>
#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "../../include/sdrmodem_hip.h"

static int write_all(int fd, const uint8_t *p, size_t n) {
    while (n > 0) {
        ssize_t w = write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) {
                continue;
            }
            return -1;
        }
        p += w;
        n -= (size_t) w;
    }
    return 0;
}

static int read_all(int fd, uint8_t *p, size_t n) {
    while (n > 0) {
        ssize_t r = read(fd, p, n);
        if (r == 0) {
            return -1; /* peer closed */
        }
        if (r < 0) {
            if (errno == EINTR) {
                continue;
            }
            return -1;
        }
        p += r;
        n -= (size_t) r;
    }
    return 0;
}

static size_t put_varint(uint8_t *p, uint64_t v) {
    size_t n = 0;
    do {
        uint8_t b = (uint8_t) (v & 0x7f);
        v >>= 7;
        p[n++] = (uint8_t) (b | (v ? 0x80 : 0));
    } while (v);
    return n;
}

/* header + Response{status, details}; both fields are `required` and therefore always on the wire (api_utils.c:82-108) */
int sdrm_wire_write_response(int socket, uint32_t status, uint32_t details) {
    uint8_t buf[6 + 2 + 10 + 10];
    size_t n = 6;
    buf[n++] = 0x08; /* field 1, varint */
    n += put_varint(buf + n, status);
    buf[n++] = 0x10; /* field 2, varint */
    n += put_varint(buf + n, details);
    const uint32_t len = htonl((uint32_t) (n - 6));
    buf[0] = SDRM_WIRE_PROTOCOL_VERSION;
    buf[1] = SDRM_WIRE_TYPE_RESPONSE;
    memcpy(buf + 2, &len, 4);
    return write_all(socket, buf, n);
}

int sdrm_wire_read_header(int socket, uint8_t *type, uint32_t *message_length) {
    uint8_t h[6];
    if (read_all(socket, h, sizeof(h)) != 0) {
        return -1;
    }
    if (h[0] != SDRM_WIRE_PROTOCOL_VERSION) {
        return -2;
    }
    uint32_t len;
    memcpy(&len, h + 2, 4);
    len = ntohl(len);
    if (len > SDRM_WIRE_MAX_MESSAGE) {
        return -3; /* no message of this protocol's RX side comes near it: do not let a peer size the caller's buffer */
    }
    *type = h[1];
    *message_length = len;
    return 0;
}

struct cursor {
    const uint8_t *p, *end;
};

static int get_varint(struct cursor *c, uint64_t *v) {
    uint64_t out = 0;
    for (int shift = 0; shift < 64 && c->p < c->end; shift += 7) {
        const uint8_t b = *c->p++;
        out |= (uint64_t) (b & 0x7f) << shift;
        if (!(b & 0x80)) {
            *v = out;
            return 0;
        }
    }
    return -1;
}

/* a field key: (field number << 3) | wire type.  Field numbers are 1 .. 2^29 - 1 (protobuf encoding rules); a key with
 * higher bits set must not alias onto a known field when truncated */
static int get_key(struct cursor *c, unsigned *field, unsigned *wire_type) {
    uint64_t key;
    if (get_varint(c, &key) != 0 || key > 0xffffffffull || (key >> 3) == 0) {
        return -1;
    }
    *field = (unsigned) (key >> 3);
    *wire_type = (unsigned) (key & 7);
    return 0;
}

static int skip_field(struct cursor *c, unsigned wire_type) {
    uint64_t v;
    switch (wire_type) {
        case 0:
            return get_varint(c, &v);
        case 1:
            if (c->end - c->p < 8) return -1;
            c->p += 8;
            return 0;
        case 2:
            if (get_varint(c, &v) != 0 || (uint64_t) (c->end - c->p) < v) return -1;
            c->p += v;
            return 0;
        case 5:
            if (c->end - c->p < 4) return -1;
            c->p += 4;
            return 0;
        default:
            return -1;
    }
}

/* fsk_demodulation_settings (api.proto:21-25) */
static int decode_fsk_settings(struct cursor c, sdrm_worker_config *cfg) {
    while (c.p < c.end) {
        uint64_t v;
        unsigned field, wt;
        if (get_key(&c, &field, &wt) != 0) return -1;
        if (wt == 0 && field >= 1 && field <= 3) {
            if (get_varint(&c, &v) != 0) return -1;
            if (field == 1) cfg->demod_fsk_deviation = (int64_t) v; /* int64: two's complement varint */
            if (field == 2) cfg->demod_fsk_transition_width = (uint32_t) v;
            if (field == 3) cfg->demod_fsk_use_dc_block = v != 0;
        } else if (skip_field(&c, wt) != 0) {
            return -1;
        }
    }
    return 0;
}

/* RxRequest body -> the request half of sdrm_worker_config (the server_config half -- buffer_size, queue_size, base_path,
 * rx_file_source -- and the Doppler callback are the caller's).  *has_doppler tells whether the request carries doppler
 * settings (the orbit model behind them, SGP4, stays with the caller).  0, or -1 for a malformed / incomplete message. */
int sdrm_wire_decode_rx_request(const uint8_t *body, size_t len, sdrm_worker_config *cfg, int *has_doppler) {
    struct cursor c = {body, body + len};
    unsigned seen = 0;
    int doppler = 0;
    while (c.p < c.end) {
        uint64_t v;
        unsigned field, wt;
        if (get_key(&c, &field, &wt) != 0) return -1;
        if (wt == 0 && field >= 1 && field <= 8) {
            if (get_varint(&c, &v) != 0) return -1;
            seen |= 1u << field;
            switch (field) {
                case 2: cfg->rx_sampling_freq = v; break;
                case 3: cfg->rx_dump_file = v != 0; break;
                case 5: if (v != 1) return -1; break; /* modem_type GMSK = 1 is all the reference has */
                case 6: cfg->demod_baud_rate = (uint32_t) v; break;
                case 7: cfg->demod_decimation = (uint32_t) v; break;
                case 8:
                    if (v > 2) return -1; /* demod_destination: FILE = 0, SOCKET = 1, BOTH = 2 (api.proto:29-33) */
                    cfg->demod_destination = (int) v;
                    break;
                default: break; /* rx_center_freq, rx_offset: the device's business */
            }
        } else if (wt == 2 && (field == 9 || field == 10 || field == 11)) {
            if (get_varint(&c, &v) != 0 || (uint64_t) (c.end - c.p) < v) return -1;
            struct cursor sub = {c.p, c.p + v};
            c.p += v;
            if (field == 9) doppler = 1;
            if (field == 10 && decode_fsk_settings(sub, cfg) != 0) return -1;
        } else if (skip_field(&c, wt) != 0) {
            return -1;
        }
    }
    if (has_doppler != NULL) {
        *has_doppler = doppler;
    }
    return (seen & 0x1fe) == 0x1fe ? 0 : -1; /* fields 1..8 are required */
}
