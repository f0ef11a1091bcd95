// sdrm_design.h -- host-side derivation of everything fsk_demod_create() derives (reference
// src/dsp/fsk_demod.c:28-78): filter taps, gains, loop constants, plus the tiling constants of the HIP path.
// Plain C++ (no HIP); done once per channel on the host in double/float exactly like the reference, then
// uploaded ("do on host, upload").
#ifndef SDRM_DESIGN_H
#define SDRM_DESIGN_H

#include <stdint.h>

#include <vector>

#include "../../include/sdrmodem_hip.h"

namespace sdrm {

// Hamming-windowed sinc low-pass, unit DC gain; reference src/dsp/lpf_taps.c:14-103.
// Returns 0 or -1 (argument checks of lpf_taps.c:14-31, with the same messages).
int design_lowpass(float gain, uint64_t sampling_freq, uint64_t cutoff_freq, uint32_t transition_width,
                   std::vector<float> &taps);

struct ChannelDesign {
    sdrm_fsk_config cfg;
    std::vector<float> taps1;  // LPF1 taps, design order (complex stage, decimation 1)
    std::vector<float> taps2;  // LPF2 taps, design order (real stage, decimation cfg.decimation)
    float quad_gain;
    float sps;
    uint32_t dc_length;  // 0 when the DC blocker is off
    float gain_omega, gain_mu, omega_lim;
    // the symbols are longer (or the DC boxcar is) than the LDS-resident DC and clock stages are sized for: the channel's
    // DC blocker and clock recovery run in their generic forms, state in global memory (sdrm_kernels.h, "generic channels")
    bool generic;
};

// 0, -1 (bad parameters, as the reference), -ENOTSUP (outside what the device path sizes for)
int design_channel(const sdrm_fsk_config &cfg, ChannelDesign &out);

}  // namespace sdrm

#endif
