// sdrm_design.cpp -- see sdrm_design.h.  Compiled with -ffp-contract=off: the fp32 normalisation of the
// taps must round exactly like the reference's (src/dsp/lpf_taps.c:89-98).
#include "sdrm_design.h"

#include <errno.h>
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "sdrm_core.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace sdrm {

int design_lowpass(float gain, uint64_t fs, uint64_t fc, uint32_t tw, std::vector<float> &taps) {
    // argument checks and messages: reference src/dsp/lpf_taps.c:14-31
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
    // tap count: lpf_taps.c:33-40 (Hamming: 53 dB => 53/22 * fs/tw, forced odd)
    int count = (int) (53.0 * (double) fs / (22.0 * (double) tw));
    count |= 1;
    const int last = count - 1;
    const int mid = last / 2;
    taps.assign((size_t) count, 0.0f);
    const double w0 = 2 * M_PI * (double) fc / (double) fs;
    for (int i = 0; i < count; i++) {
        // window value is stored as fp32 first (lpf_taps.c:50), then promoted again in the product (:83-86)
        const float window = (float) (0.54 - 0.46 * cos((2 * M_PI * i) / last));
        const int n = i - mid;
        if (n == 0) {
            taps[(size_t) i] = (float) (w0 / M_PI * window);
        } else {
            taps[(size_t) i] = (float) (sin((double) n * w0) / (n * M_PI) * window);
        }
    }
    // fp32 DC gain from the centre tap outwards (lpf_taps.c:89-92), one fp32 reciprocal-scale (:94-98)
    float dc = taps[(size_t) mid];
    for (int n = 1; n <= mid; n++) {
        dc += 2 * taps[(size_t) (mid + n)];
    }
    gain /= dc;
    for (float &t : taps) {
        t *= gain;
    }
    return 0;
}

int design_channel(const sdrm_fsk_config &cfg, ChannelDesign &out) {
    out.cfg = cfg;
    if (cfg.baud_rate == 0 || cfg.decimation == 0 || cfg.deviation == 0) {
        // the reference divides by these (fsk_demod.c:42,53); it would produce inf/NaN parameters
        fprintf(stderr, "<3>baud rate, decimation and deviation should be non-zero\n");
        return -1;
    }
    // fsk_demod.c:36-37: Carson bandwidth low-pass in front of the discriminator
    const double carson = (double) llabs(cfg.deviation) + (double) cfg.baud_rate / 2;
    int code = design_lowpass(1.0f, cfg.sampling_freq, (uint64_t) carson, (uint32_t) (0.1f * carson), out.taps1);
    if (code != 0) {
        return code;
    }
    out.quad_gain = (float) ((double) cfg.sampling_freq / (2 * M_PI * (double) cfg.deviation));  // :42
    code = design_lowpass(1.0f, cfg.sampling_freq, cfg.baud_rate / 2, cfg.transition_width, out.taps2);  // :47
    if (code != 0) {
        return code;
    }
    out.sps = (float) ((double) cfg.sampling_freq / cfg.baud_rate / cfg.decimation);  // :53
    out.dc_length = cfg.use_dc_block ? (uint32_t) (int) ceilf(out.sps * 32) : 0;     // :55-56
    out.gain_omega = (out.sps * (float) M_PI) / 100;                                  // :63
    out.gain_mu = 0.5f / 8.0f;
    out.omega_lim = out.sps * 0.01f;  // clock_recovery_mm.c:43
    // What the fast, LDS-resident stages are sized for (DESIGN.md "Supported range"):
    //  - the clock stage carries < 1.01*sps + 8 samples between calls; SDRM_CLOCK_HCAP are provisioned in its rings
    //  - the DC blocker keeps three delay lines of L + 64 floats per channel in LDS: one channel per workgroup still fits
    //    at L = 7712 (159 KB); the three-instruction quotient is proven for every length up to there (tools/dc_div_sweep)
    // A channel beyond either (the reference accepts any samples per symbol, fsk_demod.c:53-63) is served by the generic forms
    // of those two stages -- state in global memory, the IEEE division proper -- at a fraction of the speed, which such a
    // channel (>= 245 samples per symbol: a few hundred symbols per call) does not notice.
    // Fewer than ~1.01 samples per symbol (the reference accepts them: a decimation beyond the symbol length): the timing
    // loop is not tame at any amplitude, the clock stage runs such a channel from global memory (sdrm_k3_rescue).
    if (!(out.sps > 0.0f) || !(out.sps <= (float) SDRM_GEN_MAX_SPS)) {
        fprintf(stderr, "<3>samples per symbol %.3f outside the supported range (0, %d]\n", (double) out.sps, SDRM_GEN_MAX_SPS);
        return -ENOTSUP;
    }
    if (cfg.use_dc_block && out.dc_length < 2) {
        fprintf(stderr, "<3>dc blocker length %u outside the supported range\n", out.dc_length);
        return -ENOTSUP;
    }
    // (+ 24: the margin the oracle's own bound on the carried samples takes, oracle/sdrm_oracle.c orc_clock_create)
    out.generic = out.sps * 1.01f + 24.0f > (float) (SDRM_CLOCK_HCAP - 1) || (cfg.use_dc_block && out.dc_length > SDRM_DC_MAX_LEN);
    if ((size_t) cfg.decimation > out.taps2.size()) {
        fprintf(stderr, "<3>decimation %u exceeds the filter length %zu\n", cfg.decimation, out.taps2.size());
        return -ENOTSUP;
    }
    return 0;
}

}  // namespace sdrm
