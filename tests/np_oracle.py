"""Second, INDEPENDENT checker of the float soft bits: a numpy-fp32 restatement of the fsk_demod path written from
SURVEY.md Appendix B (sections B1-B8) -- not from oracle/sdrm_oracle.c, and sharing no code with it.  Test infrastructure.

It processes the WHOLE stream at once (global-index formulation, FIR vectorised across outputs and sequential over
taps), where the C oracle and the reference work chunk by chunk with carried state: two different program structures
that must produce the same bits for samples-per-symbol < 8 (SURVEY finding 3: chunk invariance).  Every arithmetic
step is an fp32 numpy operation = one IEEE rounding, no FMA; "double" is stated where Appendix B states it.

Only the two constant tables are shared data: tests/golden/tables.json (the 129 x 8 MMSE bank and the 257-entry arctan
table, extracted from the reference by tests/golden/make_golden.py).

reference: src/dsp/fsk_demod.c:28-110 and the stage files cited per function.
"""
import json
import math
import os

import numpy as np

F = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_TABLES = json.load(open(os.path.join(_HERE, "golden", "tables.json")))
ATAN = np.array(_TABLES["atan"], dtype=np.float32)          # fast_atan2f.c:23-67
MMSE = np.array(_TABLES["mmse"], dtype=np.float32)          # mmse_fir_interpolator.c:23-154, rows mu = i/128


def design_taps(fs, fc, tw):
    """B1 (lpf_taps.c:33-103): Hamming-windowed sinc, gain 1; window and taps stored fp32, normalised in fp32."""
    if fs == 0 or fc == 0 or fc > fs / 2 or tw == 0:
        return None
    ntaps = int(53.0 * float(fs) / (22.0 * float(tw)))
    if ntaps % 2 == 0:
        ntaps += 1
    m = (ntaps - 1) // 2
    w = np.array([F(0.54 - 0.46 * math.cos((2.0 * math.pi * n) / (ntaps - 1))) for n in range(ntaps)], dtype=np.float32)
    omega = 2.0 * math.pi * float(fc) / float(fs)
    h = np.zeros(ntaps, dtype=np.float32)
    for n in range(-m, m + 1):
        if n == 0:
            h[n + m] = F(omega / math.pi * float(w[n + m]))
        else:
            h[n + m] = F(math.sin(n * omega) / (n * math.pi) * float(w[n + m]))
    s = h[m]
    for n in range(1, m + 1):
        s = F(s + F(F(2.0) * h[n + m]))
    g = F(F(1.0) / s)
    return (h * g).astype(np.float32)


def fir_stream(x, taps, decim):
    """B2 (fir_filter.c:93-144): output k = sum_{j<T} x[k d - (T-1) + j] r[j], r = taps reversed, x = 0 before the stream;
    accumulated from +0, j ascending, one multiply and one add per tap.  x: float32 [n] or [n, 2] (re, im)."""
    r = taps[::-1]
    t = len(r)
    n = len(x)
    pad = np.zeros((t - 1,) + x.shape[1:], dtype=np.float32)
    xp = np.concatenate([pad, x])
    k = (n + decim - 1) // decim          # outputs whose last sample k d has arrived
    acc = np.zeros((k,) + x.shape[1:], dtype=np.float32)
    for j in range(t):
        seg = xp[j: j + (k - 1) * decim + 1: decim] if k > 0 else xp[:0]
        acc = acc + seg * r[j]
    return acc


def fir_stream_lanes(x, taps, decim, lanes, naccs):
    """The same FIR with the summation order of a SIMD dot product (what libvolk's tuned kernels do where the reference's CI
    forces the generic ones): `naccs` accumulator vectors of `lanes` floats take the products round-robin, the vectors are then
    added to the first in turn, the lanes of the result summed left to right, the taps that do not fill a block added last one
    by one.  For complex input the products (re t, im t) interleave, so a lane pair holds one tap.  Not what the reference pins
    (its tests run VOLK_GENERIC=1) -- used to show where the golden files' +-1 LSB against the generic order comes from."""
    r = taps[::-1].astype(np.float32)
    t, n = len(r), len(x)
    cplx = x.ndim == 2
    pad = np.zeros((t - 1,) + x.shape[1:], dtype=np.float32)
    xp = np.concatenate([pad, x])
    k = (n + decim - 1) // decim

    def seg(j):
        return xp[j: j + (k - 1) * decim + 1: decim]
    per_block = lanes * naccs // (2 if cplx else 1)
    nblocks = t // per_block
    accs = [[np.zeros(k, np.float32) for _ in range(lanes)] for _ in range(naccs)]
    for b in range(nblocks):
        for a in range(naccs):
            for ln in range(lanes):
                if cplx:
                    j = b * per_block + a * (lanes // 2) + ln // 2
                    p = (seg(j)[:, ln % 2] * r[j]).astype(np.float32)
                else:
                    j = b * per_block + a * lanes + ln
                    p = (seg(j) * r[j]).astype(np.float32)
                accs[a][ln] = (accs[a][ln] + p).astype(np.float32)
    v = accs[0]
    for a in range(1, naccs):
        v = [(v[ln] + accs[a][ln]).astype(np.float32) for ln in range(lanes)]
    if cplx:
        re, im = v[0], v[1]
        for ln in range(2, lanes, 2):
            re = (re + v[ln]).astype(np.float32)
            im = (im + v[ln + 1]).astype(np.float32)
        for j in range(nblocks * per_block, t):
            sj = seg(j)
            re = (re + (sj[:, 0] * r[j]).astype(np.float32)).astype(np.float32)
            im = (im + (sj[:, 1] * r[j]).astype(np.float32)).astype(np.float32)
        return np.stack([re, im], axis=1)
    d = v[0]
    for ln in range(1, lanes):
        d = (d + v[ln]).astype(np.float32)
    for j in range(nblocks * per_block, t):
        d = (d + (seg(j) * r[j]).astype(np.float32)).astype(np.float32)
    return d


def fast_atan2(y, x):
    """B4 (fast_atan2f.c:87-157), vectorised.  Finite inputs."""
    ya, xa = np.abs(y), np.abs(x)
    small, big = np.minimum(ya, xa), np.maximum(ya, xa)
    with np.errstate(divide="ignore", invalid="ignore"):
        z = np.where(ya < xa, ya / xa, xa / ya).astype(np.float32)
    alpha = (z * F(255.0)).astype(np.float32)
    idx = (np.nan_to_num(alpha).astype(np.int32)) & 0xff
    frac = (alpha - idx.astype(np.float32)).astype(np.float32)
    lo, hi = ATAN[idx], ATAN[idx + 1]
    interp = (lo + ((hi - lo).astype(np.float32) * frac).astype(np.float32)).astype(np.float32)
    b = np.where(z.astype(np.float64) < 0.003921569, z, interp).astype(np.float32)
    pi_f, half_pi_f = F(3.14159265358979323846), F(1.57079632679489661923)
    wide = xa > ya
    res_wide = np.where(x >= 0, np.where(y >= 0, b, -b), np.where(y >= 0, pi_f - b, b - pi_f))
    res_tall = np.where(y >= 0, np.where(x >= 0, half_pi_f - b, half_pi_f + b), np.where(x >= 0, -half_pi_f + b, -half_pi_f - b))
    out = np.where(wide, res_wide, res_tall).astype(np.float32)
    return np.where((ya > 0) | (xa > 0), out, F(0.0)).astype(np.float32)


def quad_demod(y, gain):
    """B3 (quadrature_demod.c:57-73): gain * atan2(x[n] conj(x[n-1])), x[-1] = 0; C99 complex product, each op rounded."""
    a, b = y[:, 0], y[:, 1]
    c = np.concatenate([[F(0.0)], a[:-1]]).astype(np.float32)
    d = np.concatenate([[F(0.0)], b[:-1]]).astype(np.float32)
    re = ((a * c).astype(np.float32) + (b * d).astype(np.float32)).astype(np.float32)
    im = ((b * c).astype(np.float32) - (a * d).astype(np.float32)).astype(np.float32)
    return (F(gain) * fast_atan2(im, re)).astype(np.float32)


def boxcar(u, length):
    """one stage of B5 (dc_blocker.c:56-64): acc = fl(fl(u_n - u_{n-L}) + acc), out = acc / (float) L"""
    delayed = np.concatenate([np.zeros(length, dtype=np.float32), u[:len(u) - length] if len(u) > length else u[:0]])[:len(u)]
    t = (u - delayed).astype(np.float32)
    acc = np.add.accumulate(np.concatenate([[F(0.0)], t]).astype(np.float32), dtype=np.float32)[1:]  # sequential, from +0
    return (acc / F(length)).astype(np.float32)


def dc_block(x, length):
    """B5 (dc_blocker.c:105-119): out_n = x[n - 2(L-1)] - v3_n"""
    v = x
    for _ in range(4):
        v = boxcar(v, length)
    lag = 2 * (length - 1)
    xd = np.concatenate([np.zeros(lag, dtype=np.float32), x])[:len(x)]
    return (xd - v).astype(np.float32)


def clock_recover(w, sps):
    """B6 + B7 (mmse_fir_interpolator.c:188-191, clock_recovery_mm.c:78-139) over the whole stream (sps < 8: the chunk
    edges of the reference fall out).  The reference's first call sees 0 carried samples; a symbol needs w[ii .. ii+7]."""
    omega_mid = F(sps)
    omega = omega_mid
    g_omega = F(F(omega_mid * F(math.pi)) / F(100.0))
    g_mu = F(0.0625)
    lim = F(omega_mid * F(0.01))
    mu = F(0.5)
    last = F(0.0)
    half = F(0.5)
    out = []
    rows = MMSE[:, ::-1].copy()  # applied reversed: w[ii + j] meets tab[imu][7 - j]
    ii, n = 0, len(w)
    while ii < n - 7:
        imu = int(np.rint(np.float64(F(mu * F(128.0)))))
        row = rows[imu]
        o = F(0.0)
        for j in range(8):
            o = F(o + F(w[ii + j] * row[j]))
        s_last = F(-1.0) if last < 0 else F(1.0)
        s_o = F(-1.0) if o < 0 else F(1.0)
        mm = F(F(s_last * o) - F(s_o * last))
        last = o
        omega = F(omega + F(g_omega * mm))
        dev = F(omega - omega_mid)
        clip = F(half * F(abs(F(dev + lim)) - abs(F(dev - lim))))
        omega = F(omega_mid + clip)
        mu = F(F(mu + omega) + F(g_mu * mm))
        fl = np.floor(mu)
        ii += int(fl)
        mu = F(mu - F(fl))
        out.append(o)
    return np.array(out, dtype=np.float32)


def soft_to_int8(o):
    """B8 (fsk_demod.c:106): clamp(o * 127) to [-128, 127], rint half-to-even"""
    r = (o * F(127.0)).astype(np.float32)
    return np.rint(np.clip(r, F(-128.0), F(127.0))).astype(np.int8)


def demod_stream(cfg, iq, simd_order=None):
    """cfg = (fs, baud, deviation, decimation, transition_width, use_dc); iq: interleaved float32 or complex64.
    Returns (int8 soft bits, float32 soft bits) of the whole stream.  fsk_demod.c:28-110.
    simd_order = (lanes, accumulators): both FIRs sum in that SIMD order (fir_stream_lanes) instead of the generic one."""
    fs, baud, dev, decim, tw, dc = cfg
    iq = np.ascontiguousarray(iq)
    x = (iq.view(np.float32) if iq.dtype == np.complex64 else iq.astype(np.float32)).reshape(-1, 2)
    carson = abs(dev) + baud / 2.0
    taps1 = design_taps(fs, int(carson), int(float(F(0.1)) * carson))   # fsk_demod.c:37, `0.1F * carson_cutoff`
    taps2 = design_taps(fs, baud // 2, tw)                               # :47
    gain = F(float(fs) / (2.0 * math.pi * float(dev)))                    # :42
    sps = F(float(fs) / baud / decim)                                    # :53
    if simd_order is None:
        y = fir_stream(x, taps1, 1)
        q = quad_demod(y, gain)
        z = fir_stream(q, taps2, decim)
    else:
        y = fir_stream_lanes(x, taps1, 1, *simd_order)
        q = quad_demod(y, gain)
        z = fir_stream_lanes(q, taps2, decim, *simd_order)
    if dc:
        z = dc_block(z, int(math.ceil(float(F(sps * F(32.0))))))         # :55-56
    soft = clock_recover(z, sps)
    return soft_to_int8(soft), soft
