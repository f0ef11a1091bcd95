/*
 * sdrm_oracle.c -- CPU ORACLE (test infrastructure, see sdrm_oracle.h).  Plain C restatement of the
 * reference's GMSK/FSK demodulation path; every function cites the reference file:line it follows.
 * Build: gcc -std=gnu11 -O2 -ffp-contract=off (oracle/Makefile).  Not part of the product.
 */
#define _GNU_SOURCE
#include "sdrm_oracle.h"

#include <errno.h>
#include <inttypes.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "orc_tables.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* most samples the clock-recovery stage may carry across calls.  The reference sizes its buffer for 8
 * (clock_recovery_mm.c:58) and silently overruns it when samples-per-symbol >= 8 (carry < 1.01*sps + 6);
 * the oracle and the device path both provision 255 (samples-per-symbol up to 244) and truncate, keeping the newest,
 * beyond that. */
#define ORC_CLOCK_HCAP 255

/* float -> int32 the way the reference's x86-64 build does it (cvttss2si): out of range / NaN -> INT_MIN.
 * In C that conversion is undefined; the reference relies on it at fast_atan2f.c:112 and
 * clock_recovery_mm.c:110,122, so the oracle fixes the x86 behaviour explicitly. */
static int cvt_f32_i32(float v) {
    if (v >= -2147483648.0f && v < 2147483648.0f) {
        return (int) v;
    }
    return INT_MIN;
}

static int cvt_f64_i32(double v) {
    if (v > -2147483649.0 && v < 2147483648.0) {
        return (int) v;
    }
    return INT_MIN;
}

/* ------------------------------------------------------------------ taps */

/* reference src/dsp/lpf_taps.c:14-31 (argument checks), :33-40 (tap count), :42-53 (Hamming window),
 * :55-103 (windowed sinc + fp32 normalisation) */
int orc_lowpass_taps(float gain, uint64_t fs, uint64_t fc, uint32_t tw, float **taps_out, size_t *len_out) {
    if (fs == 0) {
        fprintf(stderr, "<3>sampling frequency should be positive\n");
        return -1;
    }
    if (fc == 0 || (double) fc > (double) fs / 2) {
        fprintf(stderr, "<3>cutoff frequency should be positive and less than sampling freq / 2. got: %" PRIu64 "\n", fc);
        return -1;
    }
    if (tw == 0) {
        fprintf(stderr, "<3>transition width should be positive\n");
        return -1;
    }
    int ntaps = (int) (53.0 * (double) fs / (22.0 * (double) (uint64_t) tw));
    if ((ntaps & 1) == 0) {
        ntaps += 1;
    }
    float *h = calloc((size_t) ntaps, sizeof(float));
    float *win = malloc(sizeof(float) * (size_t) ntaps);
    if (h == NULL || win == NULL) {
        free(h);
        free(win);
        return -ENOMEM;
    }
    int span = ntaps - 1;
    for (int n = 0; n < ntaps; n++) {
        win[n] = (float) (0.54 - 0.46 * cos((2 * M_PI * n) / span));
    }
    int half = span / 2;
    double w0 = 2 * M_PI * (double) fc / (double) fs;
    for (int n = -half; n <= half; n++) {
        if (n == 0) {
            h[half] = (float) (w0 / M_PI * win[half]);
        } else {
            h[n + half] = (float) (sin((double) n * w0) / (n * M_PI) * win[n + half]);
        }
    }
    free(win);
    /* DC gain, accumulated in fp32 from the centre outwards, then one fp32 divide */
    float dc = h[half];
    for (int n = 1; n <= half; n++) {
        dc += 2 * h[n + half];
    }
    gain /= dc;
    for (int i = 0; i < ntaps; i++) {
        h[i] *= gain;
    }
    *taps_out = h;
    *len_out = (size_t) ntaps;
    return 0;
}

/* ------------------------------------------------------------------ FIR */

#ifdef ORC_TUNED
/* TIMING ONLY: when the box has the real libvolk (the reference's kernel library, an un-vendored system package), the
 * tuned build can run the reference's own dot-product kernels behind the same loops (bench.py, cpu_baseline.libvolk).
 * VOLK exports every kernel as a function-POINTER variable (`volk_32f_x2_dot_prod_32f_u` ...) whose first call picks
 * the machine's best implementation, so calls go through the variable.  reference call sites: src/dsp/fir_filter.c:102,132 */
#include <dlfcn.h>
typedef void (*orc_volk_dot_fn)(float *result, const float *input, const float *taps, unsigned int n);
static void *g_volk_handle;
static orc_volk_dot_fn *g_volk_dot_f, *g_volk_dot_c;

int orc_volk_attach(const char *path) {
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (h == NULL) {
        return -1;
    }
    orc_volk_dot_fn *f = (orc_volk_dot_fn *) dlsym(h, "volk_32f_x2_dot_prod_32f_u");
    orc_volk_dot_fn *c = (orc_volk_dot_fn *) dlsym(h, "volk_32fc_32f_dot_prod_32fc_u");
    if (f == NULL || c == NULL || *f == NULL || *c == NULL) {
        dlclose(h);
        return -2;
    }
    g_volk_handle = h;
    g_volk_dot_f = f;
    g_volk_dot_c = c;
    return 0;
}

void orc_volk_detach(void) {
    g_volk_dot_f = g_volk_dot_c = NULL;
    if (g_volk_handle != NULL) {
        dlclose(g_volk_handle);
        g_volk_handle = NULL;
    }
}
#endif

struct orc_fir {
    int decim;
    size_t ntaps;
    float *rev;      /* taps reversed: rev[j] = h[T-1-j]  (fir_filter.c:25-28) */
    float *rev2;     /* ORC_TUNED build only: every reversed tap twice (re, im lanes) */
    int width;       /* floats per sample */
    size_t cap;      /* max input samples per call */
    float *work;     /* kept tail of earlier input followed by this call's input */
    size_t kept;     /* samples retained from earlier calls (starts at T-1 zeros, fir_filter.c:60,73) */
    float *out;
};

int orc_fir_create(uint8_t decimation, const float *taps, size_t taps_len, size_t max_input_len, int width,
                   orc_fir **out) {
    if (taps_len == 0 || decimation == 0 || (width != 1 && width != 2)) {
        return -1;
    }
    orc_fir *f = calloc(1, sizeof(*f));
    if (f == NULL) {
        return -ENOMEM;
    }
    f->decim = decimation;
    f->ntaps = taps_len;
    f->width = width;
    f->cap = max_input_len;
    f->kept = taps_len - 1;
    f->rev = malloc(sizeof(float) * taps_len);
    f->work = calloc((max_input_len + taps_len) * (size_t) width, sizeof(float));
    f->out = malloc(sizeof(float) * (size_t) width * (max_input_len + 1));
    if (f->rev == NULL || f->work == NULL || f->out == NULL) {
        orc_fir_destroy(f);
        return -ENOMEM;
    }
    for (size_t j = 0; j < taps_len; j++) {
        f->rev[j] = taps[taps_len - 1 - j];
    }
#ifdef ORC_TUNED
    f->rev2 = malloc(sizeof(float) * 2 * taps_len);
    if (f->rev2 == NULL) {
        orc_fir_destroy(f);
        return -ENOMEM;
    }
    for (size_t j = 0; j < taps_len; j++) {
        f->rev2[2 * j] = f->rev2[2 * j + 1] = f->rev[j];
    }
#endif
    *out = f;
    return 0;
}

/* reference src/dsp/fir_filter.c:93-114 (float), :123-144 (complex), guard :146-152.
 * One output per `decim` input positions; each output is the left-to-right fp32 dot product of T
 * consecutive samples with the reversed taps (VOLK generic volk_32f_x2_dot_prod_32f /
 * volk_32fc_32f_dot_prod_32fc: `acc += in[i] * taps[i]`, re and im accumulated separately). */
void orc_fir_process(orc_fir *f, const float *input, size_t n, float **output, size_t *output_len) {
    if (n > f->cap) {
        fprintf(stderr, "<3>requested buffer %zu is more than max: %zu\n", n, f->cap);
        *output = NULL;
        *output_len = 0;
        return;
    }
    const size_t w = (size_t) f->width;
    const size_t T = f->ntaps;
    memcpy(f->work + f->kept * w, input, n * w * sizeof(float));
    size_t total = f->kept + n;
    size_t pos = 0, made = 0;
    while (pos + T <= total) {
        const float *x = f->work + pos * w;
#ifdef ORC_TUNED
        /* TIMING STAND-IN ONLY (libsdrm_oracle_tuned.so, bench.py's second CPU figure): what libvolk's SIMD dot
         * products do instead of the generic kernels -- partial sums per vector lane, i.e. a different summation
         * order and therefore NOT the pinned arithmetic.  Never used as a checker. */
        if (g_volk_dot_f != NULL) { /* the real libvolk is attached: its kernels do the dot products */
            if (w == 1) {
                (*g_volk_dot_f)(&f->out[made], x, f->rev, (unsigned int) T);
            } else {
                (*g_volk_dot_c)(&f->out[2 * made], x, f->rev, (unsigned int) T);
            }
        } else if (w == 1) {
            float a[16] = {0};
            size_t j = 0;
            for (; j + 16 <= T; j += 16) {
                for (int k = 0; k < 16; k++) {
                    a[k] += x[j + k] * f->rev[j + k];
                }
            }
            float acc = 0.0f;
            for (int k = 0; k < 16; k++) {
                acc += a[k];
            }
            for (; j < T; j++) {
                acc += x[j] * f->rev[j];
            }
            f->out[made] = acc;
        } else {
            float a[16] = {0}; /* even entries: re, odd entries: im; rev2 holds every tap twice */
            size_t j = 0;
            for (; j + 16 <= 2 * T; j += 16) {
                for (int k = 0; k < 16; k++) {
                    a[k] += x[j + k] * f->rev2[j + k];
                }
            }
            float re = 0.0f, im = 0.0f;
            for (int k = 0; k < 16; k += 2) {
                re += a[k];
                im += a[k + 1];
            }
            for (; j < 2 * T; j += 2) {
                re += x[j] * f->rev2[j];
                im += x[j + 1] * f->rev2[j + 1];
            }
            f->out[2 * made] = re;
            f->out[2 * made + 1] = im;
        }
#else
        if (w == 1) {
            float acc = 0.0f;
            for (size_t j = 0; j < T; j++) {
                acc += x[j] * f->rev[j];
            }
            f->out[made] = acc;
        } else {
            float re = 0.0f, im = 0.0f;
            for (size_t j = 0; j < T; j++) {
                re += x[2 * j] * f->rev[j];
                im += x[2 * j + 1] * f->rev[j];
            }
            f->out[2 * made] = re;
            f->out[2 * made + 1] = im;
        }
#endif
        made++;
        pos += (size_t) f->decim;
    }
    /* `pos` may overshoot `total` by up to decim-1: that is the decimation phase carried to the next call
     * (fir_filter.c:107).  The reference underflows here when decim > T; such filters are rejected upstream. */
    if (pos > total) {
        pos = total;
    }
    f->kept = total - pos;
    if (pos > 0) {
        memmove(f->work, f->work + pos * w, f->kept * w * sizeof(float));
    }
    *output = f->out;
    *output_len = made;
}

void orc_fir_destroy(orc_fir *f) {
    if (f == NULL) {
        return;
    }
    free(f->rev);
    free(f->rev2);
    free(f->work);
    free(f->out);
    free(f);
}

/* reference src/dsp/lpf.c:12-35 */
int orc_lpf_create(uint8_t decimation, uint64_t fs, uint64_t fc, uint32_t tw, size_t max_input_len, int width,
                   orc_fir **out) {
    float *taps = NULL;
    size_t n = 0;
    int code = orc_lowpass_taps(1.0f, fs, fc, tw, &taps, &n);
    if (code != 0) {
        return code;
    }
    code = orc_fir_create(decimation, taps, n, max_input_len, width, out);
    free(taps);
    return code;
}

/* ------------------------------------------------------------------ atan2 */

/* reference src/math/fast_atan2f.c:87-157.  Octant reduction, 255-step table with linear
 * interpolation, small-angle shortcut below 0.003921569 (compared in double like the reference). */
float orc_fast_atan2f(float y, float x) {
    float ya = fabsf(y), xa = fabsf(x);
    if (!(ya > 0.0f || xa > 0.0f)) {
        return 0.0f;
    }
    float z = (ya < xa) ? ya / xa : xa / ya;
    float base;
    if (z < 0.003921569) {
        base = z;
    } else {
        float a = z * 255.0f;
        int idx = cvt_f32_i32(a) & 0xff;
        a -= (float) idx;
        base = orc_atan_tab[idx];
        base += (orc_atan_tab[idx + 1] - orc_atan_tab[idx]) * a;
    }
    const float pi_f = 3.14159265358979323846f;
    const float half_pi_f = 1.57079632679489661923f;
    if (xa > ya) {
        if (x >= 0.0) {
            return (y >= 0.0) ? base : -base;
        }
        return (y >= 0.0) ? pi_f - base : base - pi_f;
    }
    if (y >= 0.0) {
        return (x >= 0.0) ? half_pi_f - base : half_pi_f + base;
    }
    return (x >= 0.0) ? -half_pi_f + base : -half_pi_f - base;
}

/* ------------------------------------------------------------------ quadrature demod */

struct orc_quad {
    float gain;
    float prev_re, prev_im; /* x[n-1]; zero before the stream starts (quadrature_demod.c:44) */
    size_t cap;
    float *out;
};

int orc_quad_create(float gain, uint32_t max_input_len, orc_quad **out) {
    orc_quad *q = calloc(1, sizeof(*q));
    if (q == NULL) {
        return -ENOMEM;
    }
    q->gain = gain;
    q->cap = max_input_len;
    q->out = malloc(sizeof(float) * (size_t) max_input_len + sizeof(float));
    if (q->out == NULL) {
        free(q);
        return -ENOMEM;
    }
    *out = q;
    return 0;
}

/* reference src/dsp/quadrature_demod.c:57-73: t = x[n] * conj(x[n-1]) (VOLK generic
 * volk_32fc_x2_multiply_conjugate_32fc = C complex multiply), out = gain * fast_atan2f(im t, re t).
 * The product is written out the way gcc evaluates a C99 complex multiply for finite operands:
 * re = a*c + b*d, im = b*c - a*d, four roundings for the products and one for each sum. */
void orc_quad_process(orc_quad *q, const float *iq, size_t n, float **output, size_t *output_len) {
    if (n > q->cap) {
        fprintf(stderr, "<3>requested buffer %zu is more than max: %zu\n", n, q->cap);
        *output = NULL;
        *output_len = 0;
        return;
    }
    float c = q->prev_re, d = q->prev_im;
    for (size_t i = 0; i < n; i++) {
        float a = iq[2 * i], b = iq[2 * i + 1];
        float re = a * c + b * d;
        float im = b * c - a * d;
        q->out[i] = q->gain * orc_fast_atan2f(im, re);
        c = a;
        d = b;
    }
    q->prev_re = c;
    q->prev_im = d;
    *output = q->out;
    *output_len = n;
}

void orc_quad_destroy(orc_quad *q) {
    if (q == NULL) {
        return;
    }
    free(q->out);
    free(q);
}

/* ------------------------------------------------------------------ DC blocker */

/* One boxcar of length L as the reference builds it (dc_blocker.c:56-64): y[n] = (u[n] - u[n-L]) + y[n-1],
 * returned divided by (float)L.  The reference shifts a delay line with memmove per sample; the oracle
 * keeps the same L past inputs in a ring -- same values, same arithmetic. */
struct orc_boxcar {
    float *ring; /* last L inputs */
    int len;     /* L */
    int head;    /* slot holding u[n-L] */
    float acc;   /* y[n-1] */
};

struct orc_dc {
    int len; /* L */
    struct orc_boxcar stage[4];
    float *xring; /* last 2(L-1) stage-0 inputs; dc_blocker.c:112-114 taps x[n - 2(L-1)] */
    int xlen;
    int xhead;
};

int orc_dc_create(int length, orc_dc **out) {
    if (length < 2) {
        return -1;
    }
    orc_dc *d = calloc(1, sizeof(*d));
    if (d == NULL) {
        return -ENOMEM;
    }
    d->len = length;
    for (int s = 0; s < 4; s++) {
        d->stage[s].ring = calloc((size_t) length, sizeof(float));
        d->stage[s].len = length;
        if (d->stage[s].ring == NULL) {
            orc_dc_destroy(d);
            return -ENOMEM;
        }
    }
    d->xlen = 2 * (length - 1);
    d->xring = calloc((size_t) d->xlen, sizeof(float));
    if (d->xring == NULL) {
        orc_dc_destroy(d);
        return -ENOMEM;
    }
    *out = d;
    return 0;
}

static inline float boxcar_step(struct orc_boxcar *b, float u) {
    float old = b->ring[b->head];
    b->ring[b->head] = u;
    b->head = (b->head + 1 == b->len) ? 0 : b->head + 1;
    float y = u - old + b->acc;
    b->acc = y;
    return y / (float) b->len;
}

/* reference src/dsp/dc_blocker.c:105-119: four cascaded boxcars, output = x[n - 2(L-1)] - y4[n], in place */
void orc_dc_process(orc_dc *d, float *inout, size_t n) {
    for (size_t i = 0; i < n; i++) {
        float x = inout[i];
        float y = boxcar_step(&d->stage[0], x);
        y = boxcar_step(&d->stage[1], y);
        y = boxcar_step(&d->stage[2], y);
        y = boxcar_step(&d->stage[3], y);
        float delayed = d->xring[d->xhead];
        d->xring[d->xhead] = x;
        d->xhead = (d->xhead + 1 == d->xlen) ? 0 : d->xhead + 1;
        inout[i] = delayed - y;
    }
}

void orc_dc_destroy(orc_dc *d) {
    if (d == NULL) {
        return;
    }
    for (int s = 0; s < 4; s++) {
        free(d->stage[s].ring);
    }
    free(d->xring);
    free(d);
}

/* ------------------------------------------------------------------ MMSE interpolator */

/* reference src/dsp/mmse_fir_interpolator.c:188-191: bank row imu = rint(mu * 128) (mu*128 in fp32,
 * rint in double, half-to-even), applied reversed as an 8-tap left-to-right dot product.
 * reference src/dsp/fir_filter.c:116-121: the dot product starts at the 16-byte aligned address at or
 * below the first sample with that many zero taps prepended (VOLK_ALIGNMENT=16 -> up to 3 samples), so
 * the samples just before the window are multiplied by 0.0f and added first: neutral for finite data,
 * NaN when one of them is NaN/Inf. */
float orc_mmse_interp(const float *base, size_t idx, float mu) {
    int imu = cvt_f64_i32(rint((double) (mu * (float) ORC_MMSE_STEPS)));
    if (imu < 0 || imu > ORC_MMSE_STEPS) {
        /* the reference indexes out of bounds here (only reachable once mu is NaN); the oracle
         * defines the result as NaN so the caller takes its NaN branch */
        return NAN;
    }
    size_t lead = idx & 3u;
    float acc = 0.0f;
    for (size_t k = lead; k > 0; k--) {
        acc += base[idx - k] * 0.0f;
    }
    const float *row = orc_mmse_bank[imu];
    for (int j = 0; j < ORC_MMSE_NTAPS; j++) {
        acc += base[idx + (size_t) j] * row[ORC_MMSE_NTAPS - 1 - j];
    }
    return acc;
}

/* ------------------------------------------------------------------ clock recovery */

struct orc_clock {
    float omega, omega_mid, omega_lim, gain_omega, mu, gain_mu, last;
    size_t cap;
    float *work; /* 16-byte aligned; kept tail followed by this call's input */
    size_t kept;
    size_t hcap; /* most samples carried between calls: what a symbol can span (< 1.01 omega + 8), never less than
                  * ORC_CLOCK_HCAP.  The reference provisions 8 and writes past its buffer beyond that; the oracle states
                  * what the reference's arithmetic yields with the buffer long enough (round 4: any samples/symbol) */
    float *out;
};

int orc_clock_create(float omega, float gain_omega, float mu, float gain_mu, float omega_relative_limit,
                     size_t max_input_len, orc_clock **out) {
    orc_clock *c = calloc(1, sizeof(*c));
    if (c == NULL) {
        return -ENOMEM;
    }
    c->omega = omega;
    c->omega_mid = omega;
    c->omega_lim = omega * omega_relative_limit; /* clock_recovery_mm.c:43 */
    c->gain_omega = gain_omega;
    c->mu = mu;
    c->gain_mu = gain_mu;
    c->last = 0.0f;
    c->cap = max_input_len;
    c->hcap = ORC_CLOCK_HCAP;
    if (omega * 1.01f + 24.0f > (float) c->hcap && omega < 1.0e6f) {
        c->hcap = (size_t) (omega * 1.01f) + 24;
    }
    c->out = malloc(sizeof(float) * (max_input_len + 1));
    void *w = NULL;
    if (posix_memalign(&w, 64, sizeof(float) * (max_input_len + c->hcap + 8)) != 0) {
        w = NULL;
    }
    c->work = w;
    if (c->out == NULL || c->work == NULL) {
        orc_clock_destroy(c);
        return -ENOMEM;
    }
    memset(c->work, 0, sizeof(float) * (max_input_len + c->hcap + 8));
    *out = c;
    return 0;
}

/* reference src/dsp/clock_recovery_mm.c:78-139 */
void orc_clock_process(orc_clock *c, const float *input, size_t n, float **output, size_t *output_len) {
    if (n > c->cap) {
        fprintf(stderr, "<3>requested buffer %zu is more than max: %zu\n", n, c->cap);
        *output = NULL;
        *output_len = 0;
        return;
    }
    memcpy(c->work + c->kept, input, n * sizeof(float));
    size_t len = c->kept + n;
    if (len < ORC_MMSE_NTAPS) { /* :94-99 */
        c->kept = len;
        *output = NULL;
        *output_len = 0;
        return;
    }
    size_t limit = len - (ORC_MMSE_NTAPS - 1);
    int ii = 0, prev = 0;
    size_t oo = 0;
    /* `ii` is an int compared against size_t in the reference: a negative ii ends the loop */
    while ((size_t) (int64_t) ii < limit && oo < c->cap) {
        float o = orc_mmse_interp(c->work, (size_t) ii, c->mu);
        if (isnan(o)) { /* :107-113 */
            c->out[oo++] = 0.0f;
            prev = ii;
            ii += cvt_f32_i32(floorf(c->omega));
            continue;
        }
        float s_last = (c->last < 0) ? -1.0f : 1.0f; /* slice(), :70-72 */
        float s_o = (o < 0) ? -1.0f : 1.0f;
        float mm = s_last * o - s_o * c->last; /* :115 */
        c->last = o;
        prev = ii;
        c->omega = c->omega + c->gain_omega * mm;
        float dev = c->omega - c->omega_mid;
        float clipped = 0.5f * (fabsf(dev + c->omega_lim) - fabsf(dev - c->omega_lim)); /* :74-76 */
        c->omega = c->omega_mid + clipped;
        c->mu = c->mu + c->omega + c->gain_mu * mm;
        float whole = floorf(c->mu);
        ii += cvt_f32_i32(whole);
        c->mu = c->mu - whole;
        c->out[oo++] = o;
    }
    /* :127-135.  When the last step jumped past the end of the data the reference restarts the next call
     * from the position of the last produced symbol (this re-emits a symbol when sps >= 8). */
    size_t from = ((size_t) (int64_t) ii > len) ? (size_t) prev : (size_t) ii;
    size_t keep = len - from;
    if (keep > c->hcap) { /* bounded where the reference would overrun its buffer; not reachable: a symbol spans < hcap */
        from = len - c->hcap;
        keep = c->hcap;
    }
    memmove(c->work, c->work + from, keep * sizeof(float));
    c->kept = keep;
    *output = c->out;
    *output_len = oo;
}

void orc_clock_destroy(orc_clock *c) {
    if (c == NULL) {
        return;
    }
    free(c->work);
    free(c->out);
    free(c);
}

/* ------------------------------------------------------------------ fsk_demod operator */

struct orc_fsk {
    orc_fir *lpf1, *lpf2;
    orc_quad *quad;
    orc_dc *dc;
    orc_clock *clock;
    int8_t *out;
    size_t out_cap;
    const float *soft;
    size_t soft_len;
    orc_fsk_info info;
};

/* reference src/dsp/fsk_demod.c:28-78 */
int orc_fsk_create(uint64_t fs, uint32_t baud, int64_t deviation, uint8_t decimation, uint32_t tw, bool use_dc,
                   uint32_t maxlen, orc_fsk **out) {
    orc_fsk *d = calloc(1, sizeof(*d));
    if (d == NULL) {
        return -ENOMEM;
    }
    double carson = (double) llabs(deviation) + (double) baud / 2; /* :36 */
    int code = orc_lpf_create(1, fs, (uint64_t) carson, (uint32_t) (0.1f * carson), maxlen, 2, &d->lpf1);
    if (code == 0) {
        d->info.quad_gain = (float) ((double) fs / (2 * M_PI * (double) deviation)); /* :42 */
        code = orc_quad_create(d->info.quad_gain, maxlen, &d->quad);
    }
    if (code == 0) {
        code = orc_lpf_create(decimation, fs, baud / 2, tw, maxlen, 1, &d->lpf2); /* :47 */
    }
    float sps = (float) ((double) fs / baud / decimation); /* :53 */
    d->info.sps = sps;
    if (code == 0 && use_dc) {
        d->info.dc_length = (uint32_t) (int) ceilf(sps * 32); /* :56 */
        code = orc_dc_create((int) d->info.dc_length, &d->dc);
    }
    if (code == 0) {
        d->info.gain_omega = (sps * (float) M_PI) / 100; /* :63 */
        d->info.gain_mu = 0.5f / 8.0f;
        code = orc_clock_create(sps, d->info.gain_omega, 0.5f, d->info.gain_mu, 0.01f, maxlen, &d->clock);
    }
    if (code == 0) {
        d->info.omega_lim = d->clock->omega_lim;
        d->info.taps1_len = (uint32_t) d->lpf1->ntaps;
        d->info.taps2_len = (uint32_t) d->lpf2->ntaps;
        d->out_cap = maxlen;
        d->out = malloc(sizeof(int8_t) * ((size_t) maxlen + 1));
        if (d->out == NULL) {
            code = -ENOMEM;
        }
    }
    if (code != 0) {
        orc_fsk_destroy(d);
        return code;
    }
    *out = d;
    return 0;
}

/* VOLK generic volk_32f_s32f_convert_8i as called at fsk_demod.c:106: scale, clamp, rint-to-even */
static inline int8_t soft_to_i8(float v) {
    float r = v * 127.0f;
    if (r > 127.0f) {
        return 127;
    }
    if (r < -128.0f) {
        return -128;
    }
    return (int8_t) rintf(r); /* NaN never reaches here: the clock stage replaces NaN symbols by 0 */
}

/* reference src/dsp/fsk_demod.c:80-110 */
void orc_fsk_process(orc_fsk *d, const float *iq, size_t n, int8_t **output, size_t *output_len) {
    float *a = NULL, *b = NULL, *c = NULL, *e = NULL;
    size_t na = 0, nb = 0, nc = 0, ne = 0;
    orc_fir_process(d->lpf1, iq, n, &a, &na);
    orc_quad_process(d->quad, a, na, &b, &nb);
    orc_fir_process(d->lpf2, b, nb, &c, &nc);
    if (d->dc != NULL && c != NULL) {
        orc_dc_process(d->dc, c, nc);
    }
    orc_clock_process(d->clock, c, nc, &e, &ne);
    for (size_t i = 0; i < ne; i++) {
        d->out[i] = soft_to_i8(e[i]);
    }
    d->soft = e;
    d->soft_len = ne;
    *output = d->out;
    *output_len = ne;
}

const float *orc_fsk_last_soft(const orc_fsk *d, size_t *len) {
    *len = d->soft_len;
    return d->soft;
}

void orc_fsk_get_info(const orc_fsk *d, orc_fsk_info *info, const float **taps1, const float **taps2) {
    *info = d->info;
    /* hand out the taps in design order (un-reversed) */
    static __thread float *scratch1 = NULL, *scratch2 = NULL;
    scratch1 = realloc(scratch1, sizeof(float) * d->lpf1->ntaps);
    scratch2 = realloc(scratch2, sizeof(float) * d->lpf2->ntaps);
    for (size_t j = 0; j < d->lpf1->ntaps; j++) {
        scratch1[j] = d->lpf1->rev[d->lpf1->ntaps - 1 - j];
    }
    for (size_t j = 0; j < d->lpf2->ntaps; j++) {
        scratch2[j] = d->lpf2->rev[d->lpf2->ntaps - 1 - j];
    }
    if (taps1 != NULL) {
        *taps1 = scratch1;
    }
    if (taps2 != NULL) {
        *taps2 = scratch2;
    }
}

void orc_fsk_destroy(orc_fsk *d) {
    if (d == NULL) {
        return;
    }
    orc_fir_destroy(d->lpf1);
    orc_fir_destroy(d->lpf2);
    orc_quad_destroy(d->quad);
    orc_dc_destroy(d->dc);
    orc_clock_destroy(d->clock);
    free(d->out);
    free(d);
}

/* ------------------------------------------------------------------ NCO (next row f-1) */

struct orc_nco {
    float phase, amplitude;
    uint64_t fs;
    uint32_t cap;
    float *osc; /* generated oscillator, interleaved */
    float *out; /* product with the input */
};

int orc_nco_create(float amplitude, uint64_t fs, uint32_t max_len, orc_nco **out) {
    orc_nco *s = calloc(1, sizeof(*s));
    if (s == NULL) {
        return -ENOMEM;
    }
    s->amplitude = amplitude;
    s->fs = fs;
    s->cap = max_len;
    s->osc = malloc(sizeof(float) * 2 * (size_t) max_len + 8);
    s->out = malloc(sizeof(float) * 2 * (size_t) max_len + 8);
    if (s->osc == NULL || s->out == NULL) {
        orc_nco_destroy(s);
        return -ENOMEM;
    }
    *out = s;
    return 0;
}

/* reference src/dsp/sig_source.c:43-58: fp32 phase accumulator, cos/sin evaluated in double on it,
 * wrap by one turn when the phase leaves [-2pi, 2pi] */
void orc_nco_process(orc_nco *s, int64_t freq, size_t n, float **iq_out, size_t *out_len) {
    const float two_pi = (float) (2 * M_PI);
    float step = two_pi * (float) freq / s->fs;
    for (size_t i = 0; i < n; i++) {
        s->osc[2 * i] = (float) (cos(s->phase) * s->amplitude);
        s->osc[2 * i + 1] = (float) (sin(s->phase) * s->amplitude);
        s->phase += step;
        if (s->phase < -two_pi) {
            s->phase += two_pi;
        }
        if (s->phase > two_pi) {
            s->phase -= two_pi;
        }
    }
    *iq_out = s->osc;
    *out_len = n;
}

/* reference src/dsp/sig_source.c:60-75 (VOLK generic volk_32fc_x2_multiply_32fc = C complex multiply) */
void orc_nco_multiply(orc_nco *s, int64_t freq, const float *iq, size_t n, float **iq_out, size_t *out_len) {
    if (n > s->cap) {
        fprintf(stderr, "<3>requested buffer %zu is more than max: %u\n", n, s->cap);
        *iq_out = NULL;
        *out_len = 0;
        return;
    }
    float *osc = NULL;
    size_t m = 0;
    orc_nco_process(s, freq, n, &osc, &m);
    for (size_t i = 0; i < n; i++) {
        float a = iq[2 * i], b = iq[2 * i + 1], c = osc[2 * i], d = osc[2 * i + 1];
        s->out[2 * i] = a * c - b * d;
        s->out[2 * i + 1] = a * d + b * c;
    }
    *iq_out = s->out;
    *out_len = n;
}

void orc_nco_destroy(orc_nco *s) {
    if (s == NULL) {
        return;
    }
    free(s->osc);
    free(s->out);
    free(s);
}

/* ------------------------------------------------------------------ Doppler batching (next row f-1) */

struct orc_doppler {
    orc_nco *nco;
    uint64_t interval;       /* update_interval_samples = sampling_freq (doppler.c:84) */
    uint64_t in_interval;    /* current_samples, starts "expired" (doppler.c:85) */
    double cur, next, slope; /* current / next second's shift and the per-sample slope between them */
    const double *shifts;
    size_t n_shifts, second; /* next per-second value to hand out */
    uint32_t cap;
    float *out;
};

int orc_doppler_create(uint64_t fs, const double *shifts, size_t n_shifts, uint32_t max_len, orc_doppler **out) {
    orc_doppler *d = calloc(1, sizeof(*d));
    if (d == NULL) {
        return -ENOMEM;
    }
    d->interval = fs;
    d->in_interval = fs;
    d->cap = max_len;
    double *copy = malloc(sizeof(double) * (n_shifts ? n_shifts : 1));
    d->out = malloc(sizeof(float) * 2 * (size_t) max_len + 8);
    if (copy == NULL || d->out == NULL || orc_nco_create(1.0f, fs, max_len, &d->nco) != 0) {
        free(copy);
        orc_doppler_destroy(d);
        return -ENOMEM;
    }
    memcpy(copy, shifts, sizeof(double) * n_shifts);
    d->shifts = copy;
    d->n_shifts = n_shifts;
    *out = d;
    return 0;
}

static double doppler_shift_at(orc_doppler *d, size_t k) {
    if (d->n_shifts == 0) {
        return 0.0;
    }
    return d->shifts[k < d->n_shifts ? k : d->n_shifts - 1];
}

/* one batch of reference src/dsp/doppler.c:131-178: its length and its (truncated) frequency */
static size_t doppler_next_batch(orc_doppler *d, size_t remaining, int64_t *freq) {
    size_t batch;
    if (d->interval < remaining + d->in_interval) {
        if (d->in_interval >= d->interval) {
            batch = d->interval < remaining ? (size_t) d->interval : remaining;
        } else {
            batch = (size_t) (d->interval - d->in_interval);
        }
    } else {
        batch = remaining;
    }
    if (d->in_interval >= d->interval) {
        d->in_interval = 0;
        if (d->next == 0) {
            d->cur = doppler_shift_at(d, d->second++);
        } else {
            d->cur = d->next;
        }
        d->next = doppler_shift_at(d, d->second++);
        d->slope = (d->next - d->cur) / d->interval;
    } else {
        d->cur += d->slope * (double) batch;
    }
    d->in_interval += batch;
    *freq = (int64_t) d->cur;
    return batch;
}

size_t orc_doppler_plan(orc_doppler *d, size_t n, uint32_t *lens, int64_t *freqs, size_t cap) {
    size_t done = 0, count = 0;
    while (done < n && count < cap) {
        int64_t f;
        size_t b = doppler_next_batch(d, n - done, &f);
        lens[count] = (uint32_t) b;
        freqs[count] = f;
        count++;
        done += b;
    }
    return count;
}

void orc_doppler_process(orc_doppler *d, const float *iq, size_t n, float **iq_out, size_t *out_len) {
    if (iq == NULL || n == 0) { /* doppler.c:117-121 */
        *iq_out = NULL;
        *out_len = 0;
        return;
    }
    if (n > d->cap) {
        fprintf(stderr, "<3>requested buffer %zu is more than max: %u\n", n, d->cap);
        *iq_out = NULL;
        *out_len = 0;
        return;
    }
    size_t done = 0;
    while (done < n) {
        int64_t f;
        size_t b = doppler_next_batch(d, n - done, &f);
        float *part = NULL;
        size_t m = 0;
        orc_nco_multiply(d->nco, f, iq + 2 * done, b, &part, &m);
        memcpy(d->out + 2 * done, part, sizeof(float) * 2 * m);
        done += b;
    }
    *iq_out = d->out;
    *out_len = done;
}

void orc_doppler_destroy(orc_doppler *d) {
    if (d == NULL) {
        return;
    }
    orc_nco_destroy(d->nco);
    free((void *) d->shifts);
    free(d->out);
    free(d);
}

/* ------------------------------------------------------------------ CPU baseline timing */

struct bench_job {
    const float *iq;
    size_t total, chunk;
    uint64_t fs;
    uint32_t baud, tw;
    int64_t dev;
    uint8_t decim;
    bool dc;
    double min_seconds;
    uint64_t samples;
    double seconds;
    int64_t checksum;
    int failed;
};

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void *bench_thread(void *arg) {
    struct bench_job *j = arg;
    orc_fsk *d = NULL;
    if (orc_fsk_create(j->fs, j->baud, j->dev, j->decim, j->tw, j->dc, (uint32_t) j->chunk, &d) != 0) {
        j->failed = 1;
        return NULL;
    }
    double t0 = now_s(), t1 = t0;
    uint64_t done = 0;
    int64_t sum = 0;
    do {
        for (size_t off = 0; off < j->total; off += j->chunk) {
            size_t n = j->total - off < j->chunk ? j->total - off : j->chunk;
            int8_t *o = NULL;
            size_t on = 0;
            orc_fsk_process(d, j->iq + 2 * off, n, &o, &on);
            for (size_t i = 0; i < on; i++) {
                sum += o[i];
            }
            done += n;
        }
        t1 = now_s();
    } while (t1 - t0 < j->min_seconds);
    j->samples = done;
    j->seconds = t1 - t0;
    j->checksum = sum;
    orc_fsk_destroy(d);
    return NULL;
}

double orc_bench_fsk(const float *iq, size_t total, size_t chunk, uint64_t fs, uint32_t baud, int64_t dev, uint8_t decim,
                     uint32_t tw, bool dc, int threads, double min_seconds, double *seconds_out, uint64_t *samples_out) {
    if (threads < 1) {
        threads = 1;
    }
    struct bench_job *jobs = calloc((size_t) threads, sizeof(*jobs));
    pthread_t *tids = calloc((size_t) threads, sizeof(*tids));
    double t0 = now_s();
    for (int t = 0; t < threads; t++) {
        jobs[t] = (struct bench_job) {.iq = iq, .total = total, .chunk = chunk, .fs = fs, .baud = baud, .tw = tw,
                                      .dev = dev, .decim = decim, .dc = dc, .min_seconds = min_seconds};
        pthread_create(&tids[t], NULL, bench_thread, &jobs[t]);
    }
    uint64_t all = 0;
    for (int t = 0; t < threads; t++) {
        pthread_join(tids[t], NULL);
        all += jobs[t].samples;
    }
    double wall = now_s() - t0;
    if (seconds_out != NULL) {
        *seconds_out = wall;
    }
    if (samples_out != NULL) {
        *samples_out = all;
    }
    free(jobs);
    free(tids);
    return (double) all / wall / 1e6;
}
