/*
 * sdrm_oracle.h -- CPU ORACLE for the GMSK/FSK demodulation hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch, plain-C restatement of the algorithm behind the reference's
 * fsk_demod_process() (dernasherbrezon/sdr-modem, src/dsp/fsk_demod.c:80-110 and the stage files it
 * chains).  It exists to CHECK the HIP path and to serve as the timed CPU baseline; nothing in the
 * product library (sdr-modem_amd/) includes, links or calls it.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * Arithmetic contract (== the reference under VOLK_GENERIC=1, VOLK_ALIGNMENT=16, which is the only
 * configuration the reference's tests pin, test/resources/run_tests.sh:8-9): every fp32 operation is
 * rounded once, dot products are accumulated left to right starting from +0, no FMA contraction
 * (build with -ffp-contract=off), denormals kept.
 *
 * How the oracle is pinned (see oracle/README.md and tests/test_oracle_golden.py):
 *   - the four end-to-end golden .s8 files of test/test_fsk_demod.c (+-2 LSB is the reference's own
 *     tolerance, test/test_fsk_demod.c:47; observed <= 1 LSB),
 *   - every inline known-answer vector of test/test_{lpf_taps,lpf,quadrature_demod,dc_blocker,
 *     mmse_fir_interpolator,clock_recovery_mm,sig_source}.c (tests/golden/ref_unit_vectors.json),
 *   - bit-for-bit against oracle/_ref/libsdrm_ref.so = the reference's OWN lpf_taps.c, dc_blocker.c and
 *     fast_atan2f.c compiled unmodified (the only hot-path sources that build without libvolk; the rest
 *     of the path needs <volk/volk.h>, which this image lacks, and is therefore NOT built -- no stand-in
 *     is written for it).
 */
#ifndef SDRM_ORACLE_H
#define SDRM_ORACLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stage: low-pass tap design (reference src/dsp/lpf_taps.c:14-103) ---- */
int orc_lowpass_taps(float gain, uint64_t sampling_freq, uint64_t cutoff_freq, uint32_t transition_width,
                     float **taps, size_t *taps_len);

/* ---- stage: decimating streaming FIR, real taps, real or complex samples
 *      (reference src/dsp/fir_filter.c:35-159, src/dsp/lpf.c:12-40) ---- */
typedef struct orc_fir orc_fir;
/* width = 1 (float samples) or 2 (interleaved complex).  Takes ownership of nothing: taps are copied. */
int orc_fir_create(uint8_t decimation, const float *taps, size_t taps_len, size_t max_input_len, int width,
                   orc_fir **out);
/* returns borrowed output pointer (valid until next call) and output length in samples */
void orc_fir_process(orc_fir *f, const float *input, size_t input_len, float **output, size_t *output_len);
void orc_fir_destroy(orc_fir *f);
/* lpf = taps + fir, as lpf_create() does */
int orc_lpf_create(uint8_t decimation, uint64_t sampling_freq, uint64_t cutoff_freq, uint32_t transition_width,
                   size_t max_input_len, int width, orc_fir **out);

/* ---- stage: fast atan2 (reference src/math/fast_atan2f.c:87-157) ---- */
float orc_fast_atan2f(float y, float x);

/* ---- stage: quadrature demod (reference src/dsp/quadrature_demod.c:23-73) ---- */
typedef struct orc_quad orc_quad;
int orc_quad_create(float gain, uint32_t max_input_len, orc_quad **out);
void orc_quad_process(orc_quad *q, const float *iq, size_t input_len, float **output, size_t *output_len);
void orc_quad_destroy(orc_quad *q);

/* ---- stage: DC blocker (reference src/dsp/dc_blocker.c:35-119); in place like the reference ---- */
typedef struct orc_dc orc_dc;
int orc_dc_create(int length, orc_dc **out);
void orc_dc_process(orc_dc *d, float *inout, size_t len);
void orc_dc_destroy(orc_dc *d);

/* ---- stage: MMSE interpolator (reference src/dsp/mmse_fir_interpolator.c:188-191 + fir_filter.c:116-121).
 *      `base` is the 16-byte aligned working buffer, idx the position of the first of 8 samples. ---- */
float orc_mmse_interp(const float *base, size_t idx, float mu);

/* ---- stage: Mueller & Mueller clock recovery (reference src/dsp/clock_recovery_mm.c:28-139) ---- */
typedef struct orc_clock orc_clock;
int orc_clock_create(float omega, float gain_omega, float mu, float gain_mu, float omega_relative_limit,
                     size_t max_input_len, orc_clock **out);
void orc_clock_process(orc_clock *c, const float *input, size_t input_len, float **output, size_t *output_len);
void orc_clock_destroy(orc_clock *c);

/* ---- the operator (reference src/dsp/fsk_demod.h:11-15, src/dsp/fsk_demod.c:28-135) ---- */
typedef struct orc_fsk orc_fsk;
int orc_fsk_create(uint64_t sampling_freq, uint32_t baud_rate, int64_t deviation, uint8_t decimation,
                   uint32_t transition_width, bool use_dc_block, uint32_t max_input_buffer_length, orc_fsk **out);
/* input: interleaved re,im fp32.  *output borrowed until the next call. */
void orc_fsk_process(orc_fsk *d, const float *iq, size_t input_len, int8_t **output, size_t *output_len);
/* float soft bits of the LAST process call (the clock-recovery output before int8 quantisation) */
const float *orc_fsk_last_soft(const orc_fsk *d, size_t *len);
/* derived parameters, for parity checks of the product's parameter derivation */
typedef struct {
    uint32_t taps1_len, taps2_len, dc_length;
    float quad_gain, sps, gain_omega, gain_mu, omega_lim;
} orc_fsk_info;
void orc_fsk_get_info(const orc_fsk *d, orc_fsk_info *info, const float **taps1, const float **taps2);
void orc_fsk_destroy(orc_fsk *d);

/* ---- next row (f-1): NCO / sig_source (reference src/dsp/sig_source.c:22-75) ---- */
typedef struct orc_nco orc_nco;
int orc_nco_create(float amplitude, uint64_t sampling_freq, uint32_t max_len, orc_nco **out);
void orc_nco_process(orc_nco *s, int64_t freq, size_t n, float **iq_out, size_t *out_len);
void orc_nco_multiply(orc_nco *s, int64_t freq, const float *iq_in, size_t n, float **iq_out, size_t *out_len);
void orc_nco_destroy(orc_nco *s);

/* ---- next row (f-1): Doppler pre-correction = batching of the NCO at one-second boundaries with a linearly
 *      interpolated, integer-truncated shift (reference src/dsp/doppler.c:116-190).  The orbit model that produces the
 *      per-second shifts (SGP4, doppler.c:31-42) stays outside: shifts[k] is the shift at second k of the pass. ---- */
typedef struct orc_doppler orc_doppler;
int orc_doppler_create(uint64_t sampling_freq, const double *shifts, size_t n_shifts, uint32_t max_len, orc_doppler **out);
/* the (len, freq_hz) batches one call of doppler_process() would hand to sig_source_multiply(); returns the count */
size_t orc_doppler_plan(orc_doppler *d, size_t input_len, uint32_t *lens, int64_t *freqs, size_t cap);
void orc_doppler_process(orc_doppler *d, const float *iq, size_t input_len, float **iq_out, size_t *out_len);
void orc_doppler_destroy(orc_doppler *d);

/* ---- CPU baseline timing helper (bench.py cpu_baseline leg) ----
 * Runs `threads` independent demodulators (one per thread, like one dsp_worker per client,
 * reference src/dsp_worker.c:188) over the same cf32 buffer in chunks of `chunk` samples, looping over
 * the buffer until at least `min_seconds` of wall time elapsed.  Returns aggregate complex Msamples/s. */
double orc_bench_fsk(const float *iq, size_t total_samples, size_t chunk, uint64_t sampling_freq, uint32_t baud_rate,
                     int64_t deviation, uint8_t decimation, uint32_t transition_width, bool use_dc_block, int threads,
                     double min_seconds, double *seconds_out, uint64_t *samples_out);

#ifdef __cplusplus
}
#endif
#endif
