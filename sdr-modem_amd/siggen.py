"""Synthetic GMSK test/bench signal: per channel c, seed 0x5D2 + c; random bits at `baud`;
rectangular NRZ upsampled to fs; Gaussian pulse shaping BT = 0.5 over 4 symbols (unit DC gain); FM with peak
deviation baud/4 (modulation index 0.5); unit amplitude; complex AWGN sigma 0.05 per component; optional static
carrier offset.  Returned as complex64 (interleaved re,im fp32 = the reference's .cf32 layout)."""
import numpy as np

SEED0 = 0x5D2


def gaussian_taps(sps, bt=0.5, span=4):
    n = int(round(span * sps)) | 1
    t = (np.arange(n) - (n - 1) / 2.0) / sps
    alpha = np.sqrt(np.log(2.0) / 2.0) / bt
    h = np.exp(-(np.pi * t / alpha) ** 2)
    return h / h.sum()


def gmsk_channel(channel, n_samples, fs=48000, baud=9600, noise=0.05, carrier_offset_hz=0.0, amplitude=1.0):
    rng = np.random.default_rng(SEED0 + int(channel))
    sps = fs / float(baud)
    nsym = int(np.ceil((n_samples + 64) / sps)) + 8
    bits = rng.integers(0, 2, nsym) * 2.0 - 1.0
    # NRZ at fs (non-integer sps handled by index mapping)
    idx = np.floor(np.arange(n_samples + 64) / sps).astype(np.int64)
    nrz = bits[idx]
    shaped = np.convolve(nrz, gaussian_taps(sps), mode="same")
    dev = baud / 4.0
    phase = 2.0 * np.pi * np.cumsum(shaped * dev + carrier_offset_hz) / fs
    sig = amplitude * np.exp(1j * phase[:n_samples])
    sig = sig + noise * (rng.standard_normal(n_samples) + 1j * rng.standard_normal(n_samples))
    return sig.astype(np.complex64)


def gmsk_batch(n_channels, n_samples, fs=48000, baud=9600, noise=0.05, first_channel=0):
    out = np.empty((n_channels, n_samples), dtype=np.complex64)
    for c in range(n_channels):
        out[c] = gmsk_channel(first_channel + c, n_samples, fs, baud, noise)
    return out
