/*
 * ref_fields.h -- the two reference types dsp_worker_create() takes, restated for builds that do not have protobuf-c and
 * libiio installed (this repository's test-suite).  Inside sdr-modem itself, compile integration/dsp_worker_ref.c with
 * -DSDRM_REF_HEADERS instead: it then includes the reference's own "api.pb-c.h" and "server_config.h".
 *
 * Layout-faithful restatement (field order and types) of
 *   struct RxRequest, FskDemodulationSettings, DopplerSettings, FileSettings   /root/reference/src/api.pb-c.h:44-121
 *   struct server_config                                                       /root/reference/src/server_config.h:16-40
 * Only the fields marked [read] are looked at by the adapter (the ones src/dsp_worker.c:108-197 reads).
 */
#ifndef SDRM_REF_FIELDS_H
#define SDRM_REF_FIELDS_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

/* what every protobuf-c message starts with (protobuf-c.h, struct ProtobufCMessage) */
typedef struct {
    const void *descriptor;
    unsigned n_unknown_fields;
    void *unknown_fields;
} ProtobufCMessage;
typedef int protobuf_c_boolean;

typedef enum { MODEM_TYPE__GMSK = 1 } ModemType;                                                  /* api.proto:4-6 */
typedef enum { DEMOD_DESTINATION__FILE = 0, DEMOD_DESTINATION__SOCKET = 1, DEMOD_DESTINATION__BOTH = 2 } DemodDestination;

typedef struct {
    ProtobufCMessage base;
    size_t n_tle;
    char **tle;
    uint32_t latitude, longitude, altitude;
} DopplerSettings;

typedef struct {
    ProtobufCMessage base;
    int64_t demod_fsk_deviation;              /* [read] */
    uint32_t demod_fsk_transition_width;      /* [read] */
    protobuf_c_boolean demod_fsk_use_dc_block; /* [read] */
} FskDemodulationSettings;

typedef struct {
    ProtobufCMessage base;
    char *filename;
    uint64_t start_time_seconds;
} FileSettings;

struct RxRequest {
    ProtobufCMessage base;
    uint64_t rx_center_freq;
    uint64_t rx_sampling_freq;                /* [read] */
    protobuf_c_boolean rx_dump_file;          /* [read] */
    int64_t rx_offset;
    ModemType demod_type;                     /* [read] */
    uint32_t demod_baud_rate;                 /* [read] */
    uint32_t demod_decimation;                /* [read] */
    DemodDestination demod_destination;       /* [read] */
    DopplerSettings *doppler;                 /* [read]: != NULL asks for Doppler pre-correction */
    FskDemodulationSettings *fsk_settings;    /* [read] */
    FileSettings *file_settings;
};

#define RX_SDR_TYPE_SDR_SERVER 0
#define RX_SDR_TYPE_PLUTOSDR 1
#define RX_SDR_TYPE_FILE 2

struct server_config {
    char *bind_address;
    uint16_t port;
    int read_timeout_seconds;
    uint32_t buffer_size;                     /* [read] */
    uint16_t queue_size;                      /* [read] */
    uint8_t rx_sdr_type;                      /* [read] */
    char *rx_sdr_server_address;
    int rx_sdr_server_port;
    char *base_path;                          /* [read] */
    char *rx_file_base_path;
    char *tx_file_base_path;
    uint8_t tx_sdr_type;
    double tx_plutosdr_gain;
    double rx_plutosdr_gain;
    unsigned int tx_plutosdr_timeout_millis;
    void *iio;                                /* iio_lib * in the reference */
};

#endif
