/*
 * ref_doppler_shifts.c -- BUILD-CONTAINER ONLY helper (test infrastructure): prints the per-second Doppler shifts the
 * reference computes for a pass, using the reference's OWN vendored SGP4/SDP4 sources (src/sgpsdp/ *.c, compiled
 * where they lie; they need nothing outside libc/libm).  The orbit model is out of scope for the GPU path (SURVEY.md
 * section 2.1 row 9: 1 evaluation per second per channel, stays on the host); this tool only produces the fixture
 * tests/golden/doppler_shifts_lucky7.json that pins the Doppler batching + NCO restatement against the reference's
 * golden file test/resources/lucky7.expected.cf32 (test/test_doppler.c:37-76).
 *
 * The shift formula and the time stepping follow the reference's src/dsp/doppler.c:31-42 (doppler_calculate_shift) and
 * :151-172; src/dsp/doppler.c itself cannot be compiled here (it pulls sig_source.c -> <volk/volk.h>).
 *
 * usage: ref_doppler_shifts lat lon alt sampling_freq center_freq constant_offset start_time n  "tle0" "tle1" "tle2"
 */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "sgp4sdp4.h"

static const double SPEED_OF_LIGHT_KM_S = 2.99792458E5;

static double shift_now(sat_t *sat, geodetic_t *gs, obs_set_t *obs, long long center, long long offset, int direction) {
    double tsince = (sat->jul_utc - sat->jul_epoch) * xmnpda;
    if (sat->flags & DEEP_SPACE_EPHEM_FLAG) {
        SDP4(sat, tsince);
    } else {
        SGP4(sat, tsince);
    }
    Convert_Sat_State(&sat->pos, &sat->vel);
    Calculate_Obs(sat->jul_utc, &sat->pos, &sat->vel, gs, obs);
    return (direction * (center - center * (SPEED_OF_LIGHT_KM_S - obs->range_rate) / SPEED_OF_LIGHT_KM_S)) + offset;
}

int main(int argc, char **argv) {
    if (argc != 12) {
        fprintf(stderr, "usage: see the header comment\n");
        return 2;
    }
    double lat = atof(argv[1]), lon = atof(argv[2]), alt = atof(argv[3]);
    double fs = atof(argv[4]);
    long long center = atoll(argv[5]), offset = atoll(argv[6]);
    time_t start = (time_t) atoll(argv[7]);
    int n = atoi(argv[8]);
    char tle[3][80];
    for (int i = 0; i < 3; i++) {
        memset(tle[i], 0, 80);
        strncpy(tle[i], argv[9 + i], 79);
    }
    geodetic_t gs;
    memset(&gs, 0, sizeof(gs));
    gs.lat = Radians(lat);
    gs.lon = Radians(lon);
    gs.alt = alt;
    gs.theta = 0.0;
    struct tm cdate;
    gmtime_r(&start, &cdate);
    cdate.tm_year += 1900;
    cdate.tm_mon += 1;
    double jul_start = Julian_Date(&cdate);
    sat_t sat;
    memset(&sat, 0, sizeof(sat));
    if (Get_Next_Tle_Set(tle, &sat.tle) != 1) {
        fprintf(stderr, "invalid tle\n");
        return 1;
    }
    select_ephemeris(&sat);
    sat.jul_epoch = Julian_Date_of_Epoch(sat.tle.epoch);
    obs_set_t obs;
    memset(&obs, 0, sizeof(obs));
    /* second k of the pass: doppler.c evaluates at jul_start, then advances by update_interval/fs/secday each second */
    sat.jul_utc = jul_start;
    printf("[");
    for (int k = 0; k < n; k++) {
        double s = shift_now(&sat, &gs, &obs, center, offset, 1);
        printf("%s%.17g", k ? ", " : "", s);
        sat.jul_utc += fs / fs / secday;
    }
    printf("]\n");
    return 0;
}
