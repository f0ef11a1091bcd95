// sdrm_core.h -- per-sample / per-symbol arithmetic of the demodulation path, written once and compiled
// both for gfx950 (inside the HIP kernels) and for the host (tests/emu: a thread-by-thread emulation of
// the kernels used by the CPU-only test suite to check indexing and state hand-off without a GPU).
//
// EXACT MODE CONTRACT: every expression below is evaluated in fp32 with one rounding per written
// operation, left to right; translation units including this header must be compiled with
// -ffp-contract=off (no FMA), IEEE division, denormals preserved.  That is what makes the soft bits
// bit-identical to the reference's generic-kernel CPU path (SURVEY.md finding 2).
#ifndef SDRM_CORE_H
#define SDRM_CORE_H

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SDRM_HD __host__ __device__ __forceinline__
#else
#define SDRM_HD static inline
#endif

#define SDRM_MMSE_TAPS 8
#define SDRM_MMSE_STEPS 128
#define SDRM_CLOCK_HCAP 256  // max samples the clock stage carries between calls: < 1.01 samples/symbol + 8 (round 3: 64 -> 256,
                             // which admits 240 kHz / 1200 baud without decimation: 200 samples per symbol)
#define SDRM_DC_MAX_LEN 7712  // longest boxcar of the DC blocker (32 x 241 samples per symbol)
#define SDRM_GEN_MAX_SPS 16384  // generic channels: samples per symbol (a DC boxcar of up to 2^19 samples, 10 MB of state per channel)
#define SDRM_INT_MIN (-2147483647 - 1)

struct sdrm_f2 {
    float x, y;
};

// bit pattern of a float and back
SDRM_HD uint32_t sdrm_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(uint32_t, f);
#else
    uint32_t u;
    memcpy(&u, &f, sizeof(u));
    return u;
#endif
}
SDRM_HD float sdrm_from_bits(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, u);
#else
    float f;
    memcpy(&f, &u, sizeof(f));
    return f;
#endif
}

// bit-field insert: mask ? a : b, bit by bit (one v_bfi_b32 on the device; the compiler does not form it by itself here)
SDRM_HD uint32_t sdrm_bfi(uint32_t mask, uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b));
    return r;
#else
    return (a & mask) | (b & ~mask);
#endif
}

// x + 1.5 * 2^23 carries rint(x) (round half to even, the current rounding mode) in its low mantissa bits for
// |x| < 2^22: one add instead of round + convert
#define SDRM_RINT_MAGIC 12582912.0f
#define SDRM_RINT_MAGIC_BITS 0x4B400000u

// float -> int32 with the x86-64 cvttss2si result for out-of-range / NaN (INT_MIN); the reference relies on
// that behaviour at src/math/fast_atan2f.c:112 and src/dsp/clock_recovery_mm.c:110,122.
SDRM_HD int sdrm_cvt_i32(float v) {
    // |v| < 2^31 also rejects NaN; -2^31 itself converts to INT_MIN either way
    return (fabsf(v) < 2147483648.0f) ? (int) v : SDRM_INT_MIN;
}

// reference src/math/fast_atan2f.c:87-157 in the reference's own shape (an if-tree); tab = 257-entry arctan table
// (sdrm_tables.h).  The kernels run sdrm_fast_atan2f_flat below; this form is the readable statement of what that one
// computes, and the CPU suite checks the two against each other and against the oracle.
SDRM_HD float sdrm_fast_atan2f(float y, float x, const float *tab) {
    float ya = fabsf(y), xa = fabsf(x);
    if (!(ya > 0.0f || xa > 0.0f)) {
        return 0.0f;
    }
    float z = (ya < xa) ? ya / xa : xa / ya;
    float base;
    if ((double) z < 0.003921569) {
        base = z;
    } else {
        float a = z * 255.0f;
        int idx = sdrm_cvt_i32(a) & 0xff;
        a = a - (float) idx;
        float t0 = tab[idx];
        float t1 = tab[idx + 1];
        base = t0 + (t1 - t0) * a;
    }
    const float pi_f = 3.14159265358979323846f;
    const float half_pi_f = 1.57079632679489661923f;
    if (xa > ya) {
        if (x >= 0.0f) {
            return (y >= 0.0f) ? base : -base;
        }
        return (y >= 0.0f) ? pi_f - base : base - pi_f;
    }
    if (y >= 0.0f) {
        return (x >= 0.0f) ? half_pi_f - base : half_pi_f + base;
    }
    return (x >= 0.0f) ? -half_pi_f + base : -half_pi_f - base;
}

// (double) z < TAN_MAP_RES (0.003921569, fast_atan2f.c:18,107) for a float z: the constant is not a float, so the test is
// z <= the float below it, i.e. z < the float above it
#define SDRM_TAN_MAP_RES_UP_BITS 0x3B808082u

// The same value as sdrm_fast_atan2f without a branch: a wave's lanes land in all eight octants, so every branch of the
// form above is taken by somebody and the wave pays for all of them (two divisions, ~140 instructions per sample; this
// form: one division, ~45).  Identities used, all exact:
//  * the two quotients are one division with selected operands (ya < xa and xa > ya are the same test);
//  * the table path is evaluated for every z (z < RES gives index 0) and selected afterwards; z is in [0, 1] or NaN, so
//    the plain conversion equals sdrm_cvt_i32 (& 0xff of INT_MIN is 0, and the device converts NaN to 0);
//  * every octant's result is offset + (+-base) with offset in {-0, +-pi, +-pi/2}: "pi - base" and "pi + (-base)" are
//    the same fp32 operation; the offset of the first octant is MINUS zero, which leaves both base = +0 and -base = -0
//    as they are (+0 would turn -0 into +0);
//  * the sign is flipped when (x >= 0) != (y >= 0) in the |x| > |y| half and when they are equal in the other.
// STRIDE: floats between two table entries (1: the plain 257-entry table; 2: the front-end's table of {entry, difference}
// pairs, of which this form reads the entries only)
template <int STRIDE = 1>
SDRM_HD float sdrm_fast_atan2f_flat(float y, float x, const float *tab) {
    const float ya = fabsf(y), xa = fabsf(x);
    const bool wide = xa > ya;
    const float z = (wide ? ya : xa) / (wide ? xa : ya);
    const float a = z * 255.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    // z is a quotient smaller / larger, i.e. in [0, 1], or NaN: a is in [0, 255] or NaN, the conversion gives 0..255 (NaN:
    // 0) and `& 0xff` has nothing to do; a - (float) idx = a - floor(a) is v_fract_f32 (both exact; NaN stays NaN)
    int idx;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(idx) : "v"(a));  // NaN -> 0, no undefined behaviour for the compiler to use
    const float frac = __builtin_amdgcn_fractf(a);
#else
    const int idx = sdrm_cvt_i32(a) & 0xff;
    const float frac = a - (float) idx;
#endif
    const float t0 = tab[idx * STRIDE];
    const float t1 = tab[(idx + 1) * STRIDE];
    const float interp = t0 + (t1 - t0) * frac;
    const float base = (z < sdrm_from_bits(SDRM_TAN_MAP_RES_UP_BITS)) ? z : interp;
    const bool xp = x >= 0.0f, yp = y >= 0.0f;
    const float pi_f = 3.14159265358979323846f;
    const float half_pi_f = 1.57079632679489661923f;
    const float off_wide = xp ? -0.0f : (yp ? pi_f : -pi_f);
    const float off_tall = yp ? half_pi_f : -half_pi_f;
    const bool flip = (xp != yp) != !wide;
    const float signed_base = sdrm_from_bits(sdrm_bits(base) ^ (flip ? 0x80000000u : 0u));
    const float angle = (wide ? off_wide : off_tall) + signed_base;
    return (ya > 0.0f || xa > 0.0f) ? angle : 0.0f;
}

// reference src/dsp/quadrature_demod.c:65-67 through the branch-free arctangent (what the front-end kernel runs)
template <int STRIDE = 1>
SDRM_HD float sdrm_quad_sample_flat(sdrm_f2 cur, sdrm_f2 prev, float gain, const float *tab) {
    const float re = cur.x * prev.x + cur.y * prev.y;
    const float im = cur.y * prev.x - cur.x * prev.y;
    return gain * sdrm_fast_atan2f_flat<STRIDE>(im, re, tab);
}

// one boxcar stage of the DC blocker, pointwise parts (reference src/dsp/dc_blocker.c:61-63).
// The running sum itself (y = t + y_prev) is the in-order chain done by the caller.
SDRM_HD float sdrm_boxcar_term(float u, float u_delayed) { return u - u_delayed; }
SDRM_HD float sdrm_boxcar_out(float running, float len_f) { return running / len_f; }

// The same quotient running / len_f (IEEE division, round to nearest even) in three instructions instead of the ~11 of the
// division expansion: with y = RN(1 / len) computed once per channel, q0 = RN(a y) is within an ulp of a / len, the
// residual r = a - q0 len comes out of one FMA exactly, and RN(q0 + r y) is the correctly rounded quotient (Markstein's
// correction step; a / len is never a rounding tie when the quotient is normal).  It does NOT hold when q0 is a
// denormal (ties exist there), an infinity or NaN: `*unsafe` is then set and the caller divides properly.  a = +0 gives
// q0 = +0, which is right as it stands.  tests/test_kernel_logic_cpu.py sweeps every fp32 significand for every supported
// length (32..3968) against the hardware division; tools/dc_div_sweep.cpp is the full sweep.
SDRM_HD float sdrm_boxcar_out_fast(float running, float len_f, float inv_len, bool *unsafe) {
    const float q0 = running * inv_len;
    const float r = fmaf(-q0, len_f, running);
    const uint32_t e = sdrm_bits(q0) & 0x7f800000u;
    *unsafe = (e == 0x7f800000u) | ((e == 0u) & (sdrm_bits(q0) != 0u));  // Inf / NaN / denormal / -0 (the FMAs would turn it into +0)
    return fmaf(r, inv_len, q0);
}

// reference src/dsp/fsk_demod.c:106 (VOLK generic volk_32f_s32f_convert_8i, scale 127)
SDRM_HD int8_t sdrm_soft_to_i8(float v) {
    float r = v * 127.0f;
    // clamp-then-round == the generic kernel's compare-then-round for every non-NaN r (NaN symbols never get here:
    // the clock stage emits 0 for them)
    r = fminf(fmaxf(r, -128.0f), 127.0f);
    return (int8_t) (int) rintf(r);
}

// the same for a value known to be finite: one median instead of max+min on the device
SDRM_HD int8_t sdrm_soft_to_i8_finite(float v) {
    float r = v * 127.0f;
#if defined(__HIP_DEVICE_COMPILE__)
    r = __builtin_amdgcn_fmed3f(r, -128.0f, 127.0f);
#else
    r = fminf(fmaxf(r, -128.0f), 127.0f);
#endif
    // rint by the magic add; the low byte of the sum is the two's-complement int8 (the magic's low byte is zero)
    return (int8_t) (uint8_t) (sdrm_bits(r + SDRM_RINT_MAGIC) & 0xffu);
}

// Mueller & Mueller loop state of one channel (reference struct clock_mm_t, clock_recovery_mm.c:9-26)
struct sdrm_mm_state {
    float mu, omega, last;
    int ii;    // position in the call's working buffer (history + input), as in the reference
    int inc;   // advance of the last produced symbol (its position was ii - inc)
};

struct sdrm_mm_consts {
    float omega_mid, omega_lim, gain_omega, gain_mu;
};

// ---- NCO (next scope row f-1; reference src/dsp/sig_source.c:43-58) ----
// The reference computes the oscillator sample as (float) cos((double) phase), (float) sin((double) phase) with the
// host's libm (sig_source.c:46).  Two double-precision implementations of cos agree to an ulp or two of a double, not
// bit for bit, and where the value lies next to a rounding boundary of fp32 (a "midpoint": probability 2^-29 per ulp
// of disagreement) the float results differ in the last place -- the device's math library did on ~1e-5 of the samples.
// So: evaluate fast; when the double lies within 16 ulp of an fp32 midpoint, evaluate again in double-double (~100 bits)
// and round THAT to fp32.  The result is the correctly rounded float of the exact cosine; a libm returns the same float
// unless its own double error (glibc: < 0.55 ulp) carries it across the midpoint, i.e. on ~1e-9 of the samples.
struct sdrm_dd {
    double hi, lo;
};
SDRM_HD sdrm_dd sdrm_dd_make(double hi, double lo) {
    sdrm_dd r;
    r.hi = hi;
    r.lo = lo;
    return r;
}
SDRM_HD sdrm_dd sdrm_two_sum(double a, double b) {  // a + b exactly (Knuth)
    const double s = a + b, bb = s - a;
    return sdrm_dd_make(s, (a - (s - bb)) + (b - bb));
}
SDRM_HD sdrm_dd sdrm_quick_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    return sdrm_dd_make(s, b - (s - a));
}
SDRM_HD sdrm_dd sdrm_two_prod(double a, double b) {  // a * b exactly
    const double p = a * b;
    return sdrm_dd_make(p, fma(a, b, -p));
}
SDRM_HD sdrm_dd sdrm_dd_add(sdrm_dd a, sdrm_dd b) {
    sdrm_dd s = sdrm_two_sum(a.hi, b.hi);
    const sdrm_dd t = sdrm_two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = sdrm_quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return sdrm_quick_two_sum(s.hi, s.lo);
}
SDRM_HD sdrm_dd sdrm_dd_mul(sdrm_dd a, sdrm_dd b) {
    sdrm_dd p = sdrm_two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return sdrm_quick_two_sum(p.hi, p.lo);
}

// sin(x), cos(x) to ~2^-100 for |x| < 2^20 (an fp32 phase is far below that): x - k pi/2 with pi/2 in three 33-bit
// pieces (each product with k is exact) and a tail, then the Taylor series in double-double on |r| <= pi/4.
SDRM_HD void sdrm_sincos_dd(double x, sdrm_dd *sn, sdrm_dd *cs) {
    static const double S[14][2] = {
        {-0x1.5555555555555p-3, -0x1.5555555555555p-57}, {0x1.1111111111111p-7, 0x1.1111111111111p-63},
        {-0x1.a01a01a01a01ap-13, -0x1.a01a01a01a01ap-73}, {0x1.71de3a556c734p-19, -0x1.c154f8ddc6c00p-73},
        {-0x1.ae64567f544e4p-26, 0x1.c062e06d1f209p-80}, {0x1.6124613a86d09p-33, 0x1.f28e0cc748ebep-87},
        {-0x1.ae7f3e733b81fp-41, -0x1.1d8656b0ee8cbp-97}, {0x1.952c77030ad4ap-49, 0x1.ac981465ddc6cp-103},
        {-0x1.2f49b46814157p-57, -0x1.2650f61dbdcb4p-112}, {0x1.71b8ef6dcf572p-66, -0x1.d043ae40c4647p-120},
        {-0x1.761b41316381ap-75, 0x1.3423c7d91404fp-130}, {0x1.3f3ccdd165fa9p-84, -0x1.58ddadf344487p-139},
        {-0x1.d1ab1c2dccea3p-94, -0x1.054d0c78aea14p-149}, {0x1.259f98b4358adp-103, 0x1.eaf8c39dd9bc5p-157}};
    static const double Cc[14][2] = {
        {-0x1.0000000000000p-1, 0.0}, {0x1.5555555555555p-5, 0x1.5555555555555p-59},
        {-0x1.6c16c16c16c17p-10, 0x1.f49f49f49f49fp-65}, {0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-76},
        {-0x1.27e4fb7789f5cp-22, -0x1.cbbc05b4fa99ap-76}, {0x1.1eed8eff8d898p-29, -0x1.2aec959e14c06p-83},
        {-0x1.93974a8c07c9dp-37, -0x1.05d6f8a2efd1fp-92}, {0x1.ae7f3e733b81fp-45, 0x1.1d8656b0ee8cbp-101},
        {-0x1.6827863b97d97p-53, -0x1.eec01221a8b0bp-107}, {0x1.e542ba4020225p-62, 0x1.ea72b4afe3c2fp-120},
        {-0x1.0ce396db7f853p-70, 0x1.aebcdbd20331cp-124}, {0x1.f2cf01972f578p-80, -0x1.9ada5fcc1ab14p-135},
        {-0x1.88e85fc6a4e5ap-89, 0x1.71c37ebd16540p-143}, {0x1.0a18a2635085dp-98, 0x1.b9e2e28e1aa54p-153}};
    const double P1 = 0x1.921fb54400000p+0, P2 = 0x1.0b4611a600000p-34, P3 = 0x1.3198a2e000000p-69, P4 = 0x1.b839a252049c1p-104;
    const double k = rint(x * 0x1.45f306dc9c883p-1);  // x * 2/pi
    sdrm_dd r = sdrm_two_sum(x, -(k * P1));             // k * P1 is exact (33-bit piece, |k| < 2^20)
    r = sdrm_dd_add(r, sdrm_dd_make(-(k * P2), 0.0));
    r = sdrm_dd_add(r, sdrm_dd_make(-(k * P3), 0.0));
    r = sdrm_dd_add(r, sdrm_two_prod(-k, P4));
    const sdrm_dd z = sdrm_dd_mul(r, r);
    sdrm_dd ps = sdrm_dd_make(S[13][0], S[13][1]), pc = sdrm_dd_make(Cc[13][0], Cc[13][1]);
    for (int n = 12; n >= 0; n--) {
        ps = sdrm_dd_add(sdrm_dd_make(S[n][0], S[n][1]), sdrm_dd_mul(z, ps));
        pc = sdrm_dd_add(sdrm_dd_make(Cc[n][0], Cc[n][1]), sdrm_dd_mul(z, pc));
    }
    const sdrm_dd s0 = sdrm_dd_add(r, sdrm_dd_mul(sdrm_dd_mul(r, z), ps));        // r + r^3 (S1 + z (S2 + ...))
    const sdrm_dd c0 = sdrm_dd_add(sdrm_dd_make(1.0, 0.0), sdrm_dd_mul(z, pc));    // 1 + z (C1 + z (C2 + ...))
    const int q = (int) ((long long) k & 3);
    const sdrm_dd ns = sdrm_dd_make(-s0.hi, -s0.lo), nc = sdrm_dd_make(-c0.hi, -c0.lo);
    *sn = q == 0 ? s0 : (q == 1 ? c0 : (q == 2 ? ns : nc));
    *cs = q == 0 ? c0 : (q == 1 ? ns : (q == 2 ? nc : s0));
}

// the float nearest to hi + lo (ties to even), for values in fp32's normal range
SDRM_HD float sdrm_dd_to_f32(sdrm_dd v) {
    const float f = (float) v.hi;
    const double r = (v.hi - (double) f) + v.lo;  // v - f: the first difference is exact
    const uint32_t fb = sdrm_bits(f);
    const bool up_is_away = (f >= 0.0f);  // for f > 0 the next float up is the one of larger magnitude
    const float f_up = sdrm_from_bits(up_is_away ? fb + 1u : fb - 1u), f_dn = sdrm_from_bits(up_is_away ? fb - 1u : fb + 1u);
    const double half_up = ((double) f_up - (double) f) * 0.5, half_dn = ((double) f - (double) f_dn) * 0.5;
    const bool even = (fb & 1u) == 0u;
    if (r > half_up || (r == half_up && !even)) {
        return f_up;
    }
    if (-r > half_dn || (-r == half_dn && !even)) {
        return f_dn;
    }
    return f;
}

// does rounding this double to fp32 depend on the double's last few bits?  (29 bits are dropped; the boundary is their half)
SDRM_HD bool sdrm_f32_rounding_is_fragile(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t b = __builtin_bit_cast(uint64_t, a);
#else
    uint64_t b;
    memcpy(&b, &a, sizeof(b));
#endif
    const int64_t d = (int64_t) (b & 0x1fffffffull) - 0x10000000ll;
    return d >= -16 && d <= 16;
}

// the oscillator sample for an fp32 phase: cos/sin evaluated in double on it, rounded to fp32 (amplitude 1)
SDRM_HD sdrm_f2 sdrm_nco_sample(float phase) {
    sdrm_f2 s;
    const double x = (double) phase, c = cos(x), sn = sin(x);
    s.x = (float) c;
    s.y = (float) sn;
    const bool fc = sdrm_f32_rounding_is_fragile(c), fs = sdrm_f32_rounding_is_fragile(sn);
    if ((fc | fs) && fabs(x) < 1048576.0 && fabs(c) > 1e-30 && fabs(sn) > 1e-30) {
        sdrm_dd es, ec;
        sdrm_sincos_dd(x, &es, &ec);
        if (fc) {
            s.x = sdrm_dd_to_f32(ec);
        }
        if (fs) {
            s.y = sdrm_dd_to_f32(es);
        }
    }
    return s;
}

// fp32 phase accumulator with a one-turn wrap when it leaves [-2pi, 2pi] (sig_source.c:47-53)
SDRM_HD float sdrm_nco_advance(float phase, float step) {
    const float two_pi = 6.28318530717958647692f;  // (float)(2 * M_PI)
    phase = phase + step;
    if (phase < -two_pi) {
        phase = phase + two_pi;
    }
    if (phase > two_pi) {
        phase = phase - two_pi;
    }
    return phase;
}

// The same step without branches or compares, for |phase| <= 2 pi and |step| <= 2 pi (what the accumulator's own wrap
// maintains).  The sum q = phase + step can only leave the interval on the side the step points to, so with
// s = -1 for step < 0 and +1 otherwise (a zero step never wraps) the two tests above are the single test s q > 2 pi.
//   ind = clamp01(fma(q, s 2^24, -2 pi 2^24)): scaling by a power of two is exact, so this is 2^24 (s q - 2 pi) rounded
//         once.  s q > 2 pi means at least one ulp of 2 pi beyond it (2^-21), times 2^24 is >= 8: ind = 1; otherwise the
//         value is <= 0: ind = 0.  Never anything in between.
//   r   = fma(ind, -s 2 pi, q): with ind = 1 the one rounding of q - s 2 pi, the reference's subtraction (addition for
//         a negative step: x + 2 pi and x - (-2 pi) are the same operation); with ind = 0 it is q + (-s 0) = q, sign of
//         zero included (q = -0 needs phase = step = -0, which takes s = +1 and adds -0).
// tests/test_kernel_logic_cpu.py compares the two forms around every boundary.
#define SDRM_NCO_WRAP_BIG 16777216.0f
#define SDRM_NCO_WRAP_C (-6.28318530717958647692f * 16777216.0f)
SDRM_HD float sdrm_nco_wrap_bigs(float step) { return step < 0.0f ? -SDRM_NCO_WRAP_BIG : SDRM_NCO_WRAP_BIG; }
SDRM_HD float sdrm_nco_wrap_negw(float step) { return step < 0.0f ? 6.28318530717958647692f : -6.28318530717958647692f; }
SDRM_HD float sdrm_nco_advance_nomask(float phase, float step, float bigs, float negw) {
    const float q = phase + step;
    float ind = fmaf(q, bigs, SDRM_NCO_WRAP_C);
    ind = ind > 1.0f ? 1.0f : (ind > 0.0f ? ind : 0.0f);  // v_fma_f32 ... clamp
    return fmaf(ind, negw, q);
}

// input sample times oscillator sample: C complex multiply as VOLK generic volk_32fc_x2_multiply_32fc (sig_source.c:71)
SDRM_HD sdrm_f2 sdrm_nco_mix(sdrm_f2 in, sdrm_f2 osc) {
    sdrm_f2 o;
    o.x = in.x * osc.x - in.y * osc.y;
    o.y = in.x * osc.y + in.y * osc.x;
    return o;
}

#endif  // SDRM_CORE_H
