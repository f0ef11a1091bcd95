/*
 * doppler_factory_ref.c -- the Doppler factory integration/dsp_worker_ref.c asks for (sdrm_ref_set_doppler_factory), built
 * on the REFERENCE'S OWN orbit model: its vendored SGP4/SDP4 sources (/root/reference/src/sgpsdp, part of sdr-modem's tree).
 * With this file linked in and
 *     sdrm_ref_set_doppler_factory(sdrm_ref_doppler_factory);
 * called once at start-up, a request with RxRequest.doppler set (TLE + ground station) gets the reference's Doppler
 * pre-correction -- src/dsp_worker.c:120-136 builds the predictor, src/dsp/doppler.c:31-42 evaluates the shift, :151-172 steps
 * the time by one update interval per evaluation -- with the oscillator and the demodulator behind it on the GPU.  The orbit
 * model itself (one evaluation per second and client) stays on the host, as SURVEY.md section 2.1 scopes it.
 *
 * Build inside sdr-modem:   cc -DSDRM_REF_HEADERS -Isrc -Isrc/sgpsdp -I<repo>/include -c integration/doppler_factory_ref.c
 * Build for the test-suite (build container only: the reference tree must be present; oracle/Makefile, target ref):
 *                           cc -I/root/reference/src/sgpsdp -I<repo>/include -Iintegration ... $(REF_SGP)
 * Nothing of this file travels to the GPU box as a product: the library never links it.
 */
#define _POSIX_C_SOURCE 200809L
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sgp4sdp4.h" /* the reference's src/sgpsdp/sgp4sdp4.h */

#ifdef SDRM_REF_HEADERS
#include "api.pb-c.h"
#include "server_config.h"
#else
#include "ref_fields.h"
#endif
#include "sdrmodem_hip.h"

/* km/s, src/dsp/doppler.c:7 */
static const double SDRM_SPEED_OF_LIGHT = 2.99792458E5;

/* what doppler_create keeps of its arguments (src/dsp/doppler.c:9-29, 44-110), without the oscillator and the buffers */
typedef struct {
    geodetic_t ground_station;
    sat_t satellite;
    obs_set_t obs_set;
    double sampling_freq;
    int64_t center_freq;
    int64_t constant_offset;
    double jul_start_time;        /* 0.0: "now", taken when the first shift is asked for (doppler.c:150-157) */
    uint64_t update_interval_samples;
    uint64_t evaluated;           /* shifts handed out so far: the next one is for second `evaluated` */
    int direction;                /* +1: reception (doppler_process_rx, doppler.c:197-199) */
} sdrm_ref_doppler;

/* src/dsp/doppler.c:31-42, doppler_calculate_shift */
static double shift_now(sdrm_ref_doppler *d) {
    const double tsince = (d->satellite.jul_utc - d->satellite.jul_epoch) * xmnpda;
    if (d->satellite.flags & DEEP_SPACE_EPHEM_FLAG) {
        SDP4(&d->satellite, tsince);
    } else {
        SGP4(&d->satellite, tsince);
    }
    Convert_Sat_State(&d->satellite.pos, &d->satellite.vel);
    Calculate_Obs(d->satellite.jul_utc, &d->satellite.pos, &d->satellite.vel, &d->ground_station, &d->obs_set);
    return (d->direction * (d->center_freq - d->center_freq * (SDRM_SPEED_OF_LIGHT - d->obs_set.range_rate) / SDRM_SPEED_OF_LIGHT)) +
           d->constant_offset;
}

/* sdrm_doppler_shift_fn: the shift for second k of the pass.  The reference evaluates at the start time and then advances the
 * satellite's clock by update_interval / sampling_freq / secday per evaluation, accumulating in a double (doppler.c:158-163);
 * the planner asks for seconds 0, 1, 2, ... in order, so the same additions happen here.  Any other order restarts the sum. */
static double sdrm_ref_doppler_shift(void *user, uint64_t second) {
    sdrm_ref_doppler *d = (sdrm_ref_doppler *) user;
    if (d->jul_start_time == 0.0) {
        struct tm t;
        UTC_Calendar_Now(&t);
        d->jul_start_time = Julian_Date(&t);
    }
    if (second != d->evaluated || second == 0) {
        d->satellite.jul_utc = d->jul_start_time;
        for (uint64_t k = 0; k < second; k++) {
            d->satellite.jul_utc += (double) d->update_interval_samples / d->sampling_freq / secday;
        }
    } else {
        d->satellite.jul_utc += (double) d->update_interval_samples / d->sampling_freq / secday;
    }
    d->evaluated = second + 1;
    return shift_now(d);
}

/* doppler_create's parameter list (src/dsp/doppler.h; the buffer length has no meaning here).  0, -ENOMEM, -1 (bad TLE / time). */
int sdrm_ref_doppler_open(double latitude, double longitude, double altitude, uint64_t sampling_freq, uint64_t center_freq,
                          int64_t constant_offset, time_t start_time_seconds, char tle[3][80], sdrm_doppler_shift_fn *fn, void **user) {
    sdrm_ref_doppler *d = calloc(1, sizeof(*d));
    if (d == NULL) {
        return -ENOMEM;
    }
    d->ground_station.lat = Radians(latitude);   /* doppler.c:62-64 */
    d->ground_station.lon = Radians(longitude);
    d->ground_station.alt = altitude;
    d->ground_station.theta = 0.0;
    d->center_freq = (int64_t) center_freq;
    d->constant_offset = constant_offset;
    if (start_time_seconds == 0) {
        d->jul_start_time = 0.0;
    } else {
        struct tm cdate;
        if (gmtime_r(&start_time_seconds, &cdate) == NULL) {
            free(d);
            return -1;
        }
        cdate.tm_year += 1900;                   /* doppler.c:75-77 */
        cdate.tm_mon += 1;
        d->jul_start_time = Julian_Date(&cdate);
    }
    d->sampling_freq = (double) sampling_freq;
    d->update_interval_samples = sampling_freq;  /* one evaluation per second, doppler.c:82 */
    d->direction = 1;
    if (Get_Next_Tle_Set(tle, &d->satellite.tle) != 1) { /* "yes yes. 1 is for success", doppler.c:101-106 */
        fprintf(stderr, "<3>invalid tle configuration\n");
        free(d);
        return -1;
    }
    select_ephemeris(&d->satellite);
    d->satellite.jul_epoch = Julian_Date_of_Epoch(d->satellite.tle.epoch);
    *fn = sdrm_ref_doppler_shift;
    *user = d;
    return 0;
}

void sdrm_ref_doppler_close(void *user) {
    free(user);
}

/* the factory: the request's Doppler settings as src/dsp_worker.c:120-136 reads them */
int sdrm_ref_doppler_factory(const struct RxRequest *req, const struct server_config *config, sdrm_doppler_shift_fn *fn, void **user) {
    (void) config;
    if (req == NULL || req->doppler == NULL || req->doppler->n_tle < 3 || req->doppler->tle == NULL) {
        return -1;
    }
    char tle[3][80];
    for (int i = 0; i < 3; i++) { /* api_utils_convert_tle, src/api_utils.c:110-114 */
        memset(tle[i], 0, sizeof(tle[i]));
        strncpy(tle[i], req->doppler->tle[i], 79);
    }
    const time_t start = req->file_settings != NULL ? (time_t) req->file_settings->start_time_seconds : 0;
    return sdrm_ref_doppler_open(req->doppler->latitude / 10E6, req->doppler->longitude / 10E6, req->doppler->altitude / 10E3,
                                 req->rx_sampling_freq, req->rx_center_freq, 0, start, tle, fn, user);
}
