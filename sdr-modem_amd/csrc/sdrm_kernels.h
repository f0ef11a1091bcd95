// sdrm_kernels.h -- device data layout and the per-thread bodies of the three pipeline kernels.
//
// The bodies are plain functions of (thread id, shared-memory pointers, global pointers) so that the HIP
// kernels in sdrm_kernels.hip are thin launch glue, and the SAME bodies can be driven thread-by-thread on
// the host by tests/emu (CPU-only suite: catches tiling / history / ring indexing mistakes without a GPU).
// The host drive is test infrastructure; the product library only ever runs these on the GPU.
//
// Pipeline per call (one "chunk" per channel, channels batched):
//   K1 front  : LPF1 (complex FIR, T1 taps) -> quadrature demod -> LPF2 (real FIR, T2 taps, decimate d)
//               grid (tiles, channels), 256 threads, IQ tile + halo staged in LDS        [parallel in time]
//   K1h       : roll the raw-IQ history (T1+T2-1 samples per channel) for the next call
//   K2 dc     : 4 cascaded boxcars + delay; a workgroup serves 16 channels: ONE chain wave runs every running sum
//               (lane = stage x channel, one add per sample), five helper roles of two waves do the pointwise parts [sequential]
//   K3 clock  : MMSE interpolator + Mueller&Mueller loop + int8; one lane per channel       [sequential]
#ifndef SDRM_KERNELS_H
#define SDRM_KERNELS_H

#include <stdio.h>

#include "sdrm_core.h"

#ifndef SDRM_K1_THREADS
#define SDRM_K1_THREADS 256
#endif
#ifndef SDRM_K1_R
#define SDRM_K1_R 15   // LPF1 outputs per thread (odd: lane stride 15*8 B is LDS-bank-conflict free)
#endif
#define SDRM_K1_RZ SDRM_K1_R  // LPF2 outputs per thread
#ifndef SDRM_K1_U
#define SDRM_K1_U 6    // taps per unrolled step
#endif
#ifndef SDRM_K1_WGS
#define SDRM_K1_WGS 4  // workgroups per CU the register budget is set for
#endif
#define SDRM_K1_NY (SDRM_K1_THREADS * SDRM_K1_R)
#define SDRM_K1_QPAD 16
// bytes of the K1 tile area in LDS: raw IQ tile + halo, later reused by the demodulated samples with the tile's LPF2
// outputs staged behind them
#define SDRM_K1_XS_BYTES(t1_max)                                                     \
    (((size_t) (SDRM_K1_NY + (t1_max)) * 8 > (size_t) (2 * SDRM_K1_NY + SDRM_K1_QPAD) * 4) \
         ? (size_t) (SDRM_K1_NY + (t1_max)) * 8                                      \
         : (size_t) (2 * SDRM_K1_NY + SDRM_K1_QPAD) * 4)

// dynamic LDS of a front-end workgroup for filters of up to t1_max / t2_max taps (the kernel's own layout: raw tile + halo,
// boundary samples, the arctangent tables, both filters' taps); the planner refuses what does not fit a CU's 160 KiB
static inline size_t sdrm_k1_lds_bytes_for(uint32_t t1_max, uint32_t t2_max) {
    return SDRM_K1_XS_BYTES(t1_max) + (size_t) SDRM_K1_THREADS * 8 + (260 + 512) * sizeof(float) +
           (size_t) (((t1_max + 3) & ~3u) + ((t2_max + 3) & ~3u) + 8) * sizeof(float);
}

// Channels per clock-recovery workgroup (one per consumer lane).  The rings of a workgroup fill most of a CU's LDS
// (which also keeps the other stages' workgroups off that CU), so channels x ring length is fixed: fewer channels =
// longer rings = longer staging steps, i.e. fewer hand-overs (barrier, limits, loop entry with its exposed first loads)
// per symbol, and LDS instructions that move a quarter of the data.  A lone wave pays per instruction, not per lane:
// with 16 of the consumer's 64 lanes in use a symbol costs ~257 cycles instead of ~312 (64 lanes), at the price of four
// times as many CUs held.  sdrm_k3_shape_for() picks per batch size.
#define SDRM_K3_WAVE 64     // lanes of the staging wave (lane = time)
#define SDRM_K3_PRE 3       // mirror slots below slot 0 (a symbol reads up to 3 samples before its window)
#define SDRM_K3_POST 8      // mirror slots above slot RING-1 (a window is 8 samples)
// A channel's ring holds PAIRS: element e = {x[e], x[e+1]} (8 bytes, 8-byte aligned), so that the 8 samples of a window
// starting anywhere are elements s, s+2, s+4, s+6: two ds_read2_b64 instead of four ds_read2_b32 (an LDS instruction
// costs a lone wave ~12 cycles of issue whatever its width, tools/ubench_chain.hip; a 16-byte read that is only 4-byte
// aligned is served, but in 64 cycles: tools/ubench_lds_unaligned.hip).
// PLAIN: the ring holds plain samples (4 bytes each) instead of pair elements: half the LDS, and a window that starts
// anywhere is four 4-byte aligned ds_read2_b32 instead of two ds_read2_b64 (two more LDS instructions per symbol).
template <int LANES, int RING = 16384 / LANES, bool PLAIN = false>
struct sdrm_k3_geom {
    static constexpr int lanes = LANES;
    static constexpr int ring = RING;                    // per-channel sample ring in LDS (power of two): 4 staging blocks
    static constexpr int block = ring / 4;               // samples staged per channel per step
    static constexpr int segs = block / SDRM_K3_WAVE;    // 64-sample row segments per channel and step
    static constexpr bool plain = PLAIN;
    static constexpr int rows = SDRM_K3_PRE + ring + SDRM_K3_POST;
    // floats between two channels' rings: pair elements 22 mod 64 for every power-of-two ring (conflict-free b64 reads),
    // plain samples an odd number
    static constexpr int cpitch = PLAIN ? rows : 2 * rows;
    static_assert(segs >= 1, "a staging block is at least one 64-sample row segment");
};
#ifndef SDRM_K3_BANKPITCH
#define SDRM_K3_BANKPITCH 12  // floats between two rows of the MMSE bank copy in LDS (48 B: rows start on 16 different bank offsets instead of 8)
#endif
// Workgroup shape of the clock stage: channels per workgroup x ring length.  Two families:
//   FULL shapes (lanes x ring = 16384 samples: 16 x 1024, 32 x 512, 64 x 256) fill a CU's LDS (141 KB), which keeps every
//   other stage's workgroups off that CU: the symbol loop has its SIMD to itself.  Right while the clock stage is the
//   longest stage (few channels).
//   SLIM shapes (16 x 256: 41 KB, 16 x 512 / 32 x 256: 74 KB) leave room for front-end workgroups (35 KB each) on the same
//   CU.  With thousands of channels the front-end is the longest stage and a FULL clock stage both takes whole CUs away
//   from it and cannot place its own workgroups while the front-end's grid streams through the chip (no CU ever has
//   141 KB free): measured in round 2 as 8.45 ms for a 3.76 ms kernel.  A slim workgroup starts wherever one front-end
//   workgroup has left, and costs the front-end one of its four slots on that CU instead of all of them.
struct sdrm_k3_shape {
    int lanes, ring, plain;
};
SDRM_HD bool sdrm_k3_shape_ok(int lanes, int ring, int plain) {
    if (plain) {
        return (lanes == 64 || lanes == 32) && ring == 256;
    }
    return (lanes == 16 && (ring == 1024 || ring == 512 || ring == 256)) || (lanes == 32 && (ring == 512 || ring == 256)) ||
           (lanes == 64 && ring == 256);
}
// channels per workgroup for a batch of n channels (round 1, ms per step, 131072-sample chunks; lanes 64 / 32 / 16 / 8:
// 256 channels 3.96 / 3.18 / 3.09 / 3.07, 1024: 4.67 / 4.37 / 3.88 / 4.46)
// carried_max: the most samples any channel of the batch can carry between calls (< 1.01 samples/symbol + 8).  They sit
// in the ring in front of the call's first block while the staging wave already writes the second: the ring must hold
// them plus two blocks, i.e. ring / 2 >= carried_max -- long symbols (more than ~120 samples) need the 1024-sample ring.
static inline sdrm_k3_shape sdrm_k3_shape_for(int n_channels, int forced_lanes, int forced_ring, int forced_plain, int carried_max) {
    if (forced_lanes == 16 || forced_lanes == 32 || forced_lanes == 64) {
        sdrm_k3_shape f = {forced_lanes, forced_ring ? forced_ring : 16384 / forced_lanes, forced_plain};
        if (sdrm_k3_shape_ok(f.lanes, f.ring, f.plain) && f.ring / 2 >= carried_max) {
            return f;
        }
    }
    if (carried_max > 128) {
        sdrm_k3_shape l = {16, 1024, 0};
        return l;
    }
    // Measured per batch size (round 2, tools/k3_ab.py, ms per call of 131072 samples per channel; 16 / 32 / 64 channels
    // per workgroup, full shapes): 1280 channels 3.28 / 3.82 / 4.48, 1536: 4.02 / 3.81 / 4.62, 2048: 5.50 / 4.87 / 4.84,
    // 3072: 8.18 / 7.21 / 7.20, 4096: 10.8 / 9.56 / 9.17.  Few channels per workgroup = more waves on the latency chain
    // but more CUs whose LDS the front-end cannot use; the crossovers sit where the front-end becomes the longer stage.
    // Round 3: beyond ~2500 channels the 64-channel workgroup takes the PLAIN ring (75 KB instead of 141 KB: two front-end
    // workgroups fit beside it, and its workgroups are placed as soon as the previous call's leave instead of when the
    // front-end's grid drains): 3072 channels 7.07 -> 6.95 ms per call, 4096: 9.22 -> 9.02 (profiles/r03_k3_shapes.txt);
    // at 2048 the 32-channel pair shape is still ahead (4.96 against 5.33).
    sdrm_k3_shape s = {16, 1024, 0};
    if (n_channels > 1280) {
        s.lanes = n_channels <= 2560 ? 32 : 64;
        s.ring = 16384 / s.lanes;
        s.plain = n_channels > 2560;
    }
    return s;
}
// "lanes", "lanesxring" or "lanesxringp" (SDRM_K3_LANES=64x256p: plain ring): tests and measurements force one shape
static inline void sdrm_k3_parse_shape(const char *text, int *lanes, int *ring, int *plain) {
    *lanes = 0;
    *ring = 0;
    *plain = 0;
    if (text != nullptr) {
        char tail = 0;
        sscanf(text, "%dx%d%c", lanes, ring, &tail);
        *plain = tail == 'p';
    }
}

// immutable per-channel parameters (device array, one per channel)
struct sdrm_chan_params {
    uint32_t T1, T2, decim, dc_len;
    uint32_t taps1_off, taps2_off;  // float offsets into the tap pool; taps are stored REVERSED (fir_filter.c:25-28)
    uint32_t hist_len;              // raw-IQ history carried between calls: T1 + T2 - 1 samples
    uint32_t tile_m;                // LPF2 outputs produced by one K1 tile
    uint32_t max_len;               // max_input_buffer_length
    uint32_t dc_state_off;          // float offset of this channel's DC state in the dc state pool
    float quad_gain;
    float dc_len_f, dc_inv_len;     // (float) L and RN(1 / L): sdrm_boxcar_out_fast
    float omega_mid, omega_lim, gain_omega, gain_mu;
    uint32_t generic;               // the channel's DC blocker and clock recovery run in their generic forms (below: "generic channels")
    float amp_safe;                 // clock-stage input amplitude below which the timing loop provably advances >= 1 sample per symbol ("tame")
    uint32_t can_wild;              // the stages in front of the clock stage CAN produce a sample beyond amp_safe (host-side bounds only)
    uint32_t pad_[1];
};

// Per-call flags of a channel (DeviceBatch::nonfinite[c]), raised by the front-end (no DC blocker) or the DC stage on the stream
// the clock stage is about to read, and the same bits in sdrm_clock_state::poison for what a call leaves behind:
//   NONFINITE  NaN/Inf present: the clock stage's wave takes the NaN-aware form of the symbol (same ring, same order)
//   WILD       a sample at or above the channel's amp_safe: the timing error can exceed the symbol length, i.e. the loop
//              may stand still or walk BACKWARDS through its buffer (reference src/dsp/clock_recovery_mm.c:121-122 with
//              floorf(mu) <= 0), further than an LDS ring remembers.  The channel leaves the ring-based loop for this call
//              and is run from global memory, statement by statement (sdrm_k3_rescue).
#define SDRM_FLAG_NONFINITE 1u
#define SDRM_FLAG_WILD 2u
// max over the 129 rows of the MMSE bank of sum |tap| (sdrm_tables.h; checked by tests/test_kernel_logic_cpu.py): an
// interpolated symbol is at most this times the largest sample of its window
#define SDRM_MMSE_ABS_SUM 1.5975f

// per-call, per-channel control block, written by the host before every launch
struct sdrm_chunk_ctl {
    uint32_t n_in;    // complex input samples consumed by this call
    uint32_t i0;      // in-chunk input index of the first LPF2 output (decimation phase)
    uint32_t nz;      // LPF2 outputs of this call
    uint32_t tiles;   // K1 tiles for this channel
    uint32_t parity;  // which of the two raw-history buffers is current
    uint32_t zbase;   // LPF2 outputs produced by earlier calls (mod 2^32): DC ring phase
    uint32_t nco_off, nco_cnt;  // this call's NCO segments of the channel in the segment table (cnt 0: no NCO)
    uint32_t absent;  // the channel takes no part in this call (SDRM_LEN_ABSENT): no output, stream state untouched --
                      // unlike an EMPTY call, which the reference's clock stage answers from its carried samples
    uint32_t pre;     // the channel's input of this call has been through its constant-frequency oscillator (the file source's
                      // rx_offset, reference src/sdr/file_source.c:120-128) and lies in nco_out: what follows reads it there
};

// one batch of the Doppler pre-correction: `len` samples mixed with an oscillator advancing `step` radians per sample
// (reference src/dsp/doppler.c:180 -> sig_source.c:44: step = fl(fl(2pi_f * (float)freq_hz) / fs), computed on the host)
struct sdrm_nco_seg {
    uint32_t len;
    float step;
};

// mutable clock-recovery state (device array, one per channel)
struct sdrm_clock_state {
    float mu, omega, last;
    uint32_t kept;    // samples carried in hist[] (< SDRM_CLOCK_HCAP)
    uint32_t poison;  // the carried samples may contain NaN/Inf (previous call saw some): no fast path
    uint32_t pad[3];
    float hist[SDRM_CLOCK_HCAP];
};

// ------------------------------------------------------------------------------------------------ K1

struct sdrm_k1_tile {
    int m;        // LPF2 outputs of this tile
    int o_lo;     // chunk-relative index of the first one
    int nq;       // quadrature-demod samples needed: (m-1)*d + T2
    int ny;       // LPF1 outputs needed: nq + 1 (one predecessor)
    int nx;       // raw samples needed: ny + T1 - 1
    int x_first;  // in-chunk index of the first raw sample (negative => history)
};

SDRM_HD sdrm_k1_tile sdrm_k1_tile_setup(const sdrm_chan_params &p, const sdrm_chunk_ctl &c, int tile) {
    sdrm_k1_tile t;
    t.o_lo = tile * (int) p.tile_m;
    int left = (int) c.nz - t.o_lo;
    t.m = left < (int) p.tile_m ? left : (int) p.tile_m;
    t.nq = (t.m - 1) * (int) p.decim + (int) p.T2;
    t.ny = t.nq + 1;
    t.nx = t.ny + (int) p.T1 - 1;
    t.x_first = (int) c.i0 + t.o_lo * (int) p.decim - ((int) p.T2 - 1) - 1 - ((int) p.T1 - 1);
    return t;
}

// logical input stream of a call: history for negative indices, the caller's buffer otherwise
SDRM_HD sdrm_f2 sdrm_ext_sample(const sdrm_f2 *in, const sdrm_f2 *hist, int hist_len, int i) {
    const sdrm_f2 *p = (i < 0) ? hist + (hist_len + i) : in + i;  // one 8-byte load from the selected address
    return *p;
}

// One tap of a dot product: product and sum rounded separately, as the reference's VOLK generic kernels do (what every
// parity claim rests on; built with -ffp-contract=off).  A fused multiply-add here is NOT the reference's bits: round 2-5's
// opt-in SDRM_FLAG_FAST_FMA build failed the reference's own +-2 LSB tolerance on one of its four fixtures (19 LSB on
// lucky7 without DC blocker) and was removed in round 6.
SDRM_HD float sdrm_mac(float acc, float x, float t) {
    return acc + x * t;
}

// K sequential taps on N adjacent outputs of a unit-stride FIR, register blocked: for every output the
// taps are visited in increasing j, one fp32 multiply and one fp32 add each -- the reference's order
// (fir_filter.c:100-105 / :130-135 with VOLK generic dot products).
template <int N, int K>
SDRM_HD void sdrm_fir_block_c(const sdrm_f2 *xs, const float *taps, int ntaps, sdrm_f2 (&acc)[N]) {
    int j0 = 0;
    for (; j0 + K <= ntaps; j0 += K) {
        sdrm_f2 w[N + K - 1];
#pragma unroll
        for (int k = 0; k < N + K - 1; k++) {
            w[k] = xs[j0 + k];
        }
#pragma unroll
        for (int u = 0; u < K; u++) {
            const float tp = taps[j0 + u];
#pragma unroll
            for (int r = 0; r < N; r++) {
                acc[r].x = sdrm_mac(acc[r].x, w[r + u].x, tp);
                acc[r].y = sdrm_mac(acc[r].y, w[r + u].y, tp);
            }
        }
    }
    for (; j0 < ntaps; j0++) {
        const float tp = taps[j0];
#pragma unroll
        for (int r = 0; r < N; r++) {
            sdrm_f2 v = xs[j0 + r];
            acc[r].x = sdrm_mac(acc[r].x, v.x, tp);
            acc[r].y = sdrm_mac(acc[r].y, v.y, tp);
        }
    }
}

template <int N, int K>
SDRM_HD void sdrm_fir_block_r(const float *xs, const float *taps, int ntaps, float (&acc)[N]) {
    int j0 = 0;
    for (; j0 + K <= ntaps; j0 += K) {
        float w[N + K - 1];
#pragma unroll
        for (int k = 0; k < N + K - 1; k++) {
            w[k] = xs[j0 + k];
        }
#pragma unroll
        for (int u = 0; u < K; u++) {
            const float tp = taps[j0 + u];
#pragma unroll
            for (int r = 0; r < N; r++) {
                acc[r] = acc[r] + w[r + u] * tp;
            }
        }
    }
    for (; j0 < ntaps; j0++) {
        const float tp = taps[j0];
#pragma unroll
        for (int r = 0; r < N; r++) {
            acc[r] = acc[r] + xs[j0 + r] * tp;
        }
    }
}

// Two-wide helper for the real FIR: on the device a 2-vector whose multiply and add become one v_pk_mul_f32 / v_pk_add_f32
// (each component rounded separately, no contraction); on the host the same two operations written out.
#if defined(__HIP_DEVICE_COMPILE__)
typedef float sdrm_v2 __attribute__((ext_vector_type(2)));
SDRM_HD sdrm_v2 sdrm_v2_make(float a, float b) {
    sdrm_v2 v = {a, b};
    return v;
}
SDRM_HD sdrm_v2 sdrm_v2_mac(sdrm_v2 acc, sdrm_v2 x, float t) {
    return acc + x * t;
}
#else
struct sdrm_v2 {
    float x, y;
};
SDRM_HD sdrm_v2 sdrm_v2_make(float a, float b) {
    sdrm_v2 v = {a, b};
    return v;
}
SDRM_HD sdrm_v2 sdrm_v2_mac(sdrm_v2 acc, sdrm_v2 x, float t) {
    sdrm_v2 r;
    r.x = sdrm_mac(acc.x, x.x, t);
    r.y = sdrm_mac(acc.y, x.y, t);
    return r;
}
#endif

// one step of K taps of sdrm_fir_block_rp (below): window of N + K - 1 samples from xs + j0, as even- and odd-aligned pairs
template <int N, int K>
SDRM_HD void sdrm_fir_step_rp(const float *xs, const float *xo, const float *taps, int j0, sdrm_v2 (&pa)[N / 2], float &tail) {
    constexpr int P = N / 2;
    constexpr int W = N + K - 1;  // window floats per step
    sdrm_v2 we[(W + 1) / 2], wo[(W + 1) / 2];
#pragma unroll
    for (int k = 0; 2 * k + 1 < W; k++) {
        we[k] = sdrm_v2_make(xs[j0 + 2 * k], xs[j0 + 2 * k + 1]);
    }
    if (W & 1) {
        we[W / 2] = sdrm_v2_make(xs[j0 + W - 1], 0.0f);  // the window's last sample when it has no partner (only .x is used)
    }
#pragma unroll
    for (int k = 0; 2 * k + 2 < W; k++) {
        wo[k] = sdrm_v2_make(xo[j0 + 2 * k], xo[j0 + 2 * k + 1]);
    }
#pragma unroll
    for (int u = 0; u < K; u++) {
        const float tp = taps[j0 + u];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int i = 2 * p + u;
            pa[p] = sdrm_v2_mac(pa[p], (i & 1) ? wo[i / 2] : we[i / 2], tp);
        }
        if (N & 1) {
            const int i = N - 1 + u;
            const float x = (i & 1) ? we[i / 2].y : we[i / 2].x;
            tail = sdrm_mac(tail, x, tp);
        }
    }
}

// the `rest` < K taps left after the last whole step: one step of exactly that many (REST counts down to the match)
template <int N, int REST>
SDRM_HD void sdrm_fir_rest_rp(const float *xs, const float *xo, const float *taps, int j0, int rest, sdrm_v2 (&pa)[N / 2], float &tail) {
    if constexpr (REST > 0) {
        if (rest == REST) {
            sdrm_fir_step_rp<N, REST>(xs, xo, taps, j0, pa, tail);
        } else {
            sdrm_fir_rest_rp<N, REST - 1>(xs, xo, taps, j0, rest, pa, tail);
        }
    }
}

// The same real FIR block with neighbouring outputs paired: (acc[2p], acc[2p+1]) += (x[2p+u], x[2p+u+1]) * tap, one
// packed multiply and one packed add per pair and tap (half the VALU instructions of sdrm_fir_block_r; per output the
// operations and their order are unchanged).  The window is kept twice, as even-aligned pairs (x[2k], x[2k+1]) and as
// odd-aligned pairs (x[2k+1], x[2k+2]), so that every pair operand is a register pair whatever the tap's parity.
// The taps left over after the last whole step of K take one shorter step of the same form (round 2 ran them unpacked:
// 3 of the 57 taps of the 9600-baud filter at two instructions per output instead of one).
template <int N, int K>
SDRM_HD void sdrm_fir_block_rp(const float *xs, const float *taps, int ntaps, float (&acc)[N]) {
    constexpr int P = N / 2;
    sdrm_v2 pa[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        pa[p] = sdrm_v2_make(acc[2 * p], acc[2 * p + 1]);
    }
    float tail = (N & 1) ? acc[N - 1] : 0.0f;
    // the odd-aligned pairs are READ from LDS as pairs (a load costs no vector instruction) instead of being assembled from
    // the even-aligned ones with register moves (13 per 6 taps: 2 % of the front-end's vector instructions); the compiler
    // must not see that xo is xs + 1, or it shares the loads again
    int one = 1;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(one));  // the offset, not the pointer: the pointer must stay recognisable as an LDS address
#endif
    const float *xo = xs + one;
    int j0 = 0;
    for (; j0 + K <= ntaps; j0 += K) {
        sdrm_fir_step_rp<N, K>(xs, xo, taps, j0, pa, tail);
    }
    sdrm_fir_rest_rp<N, K - 1>(xs, xo, taps, j0, ntaps - j0, pa, tail);
#pragma unroll
    for (int p = 0; p < P; p++) {
        acc[2 * p] = pa[p].x;
        acc[2 * p + 1] = pa[p].y;
    }
    if (N & 1) {
        acc[N - 1] = tail;
    }
}

// Decimating real FIR (reference src/dsp/fir_filter.c:93-114 with decimation d > 1) on R outputs whose windows start at
// xs[r * stride]: output r accumulates xs[r * stride + j] * taps[j], j ascending, one separately rounded multiply and add
// each -- the reference's order.  Outputs are paired into packed operands like sdrm_fir_block_rp's (two outputs meet the
// same tap on samples `stride` apart); every operand is its own LDS read.  The kernel hands a thread the outputs tid,
// tid + 256, ...: consecutive lanes then read addresses d floats apart (conflict-free for odd d) and all four waves share
// the work -- round 2 gave each of the first few threads fifteen consecutive outputs and left three waves idle.
template <int R, int K>
SDRM_HD void sdrm_fir_block_rd(const float *xs, int stride, int valid, const float *taps, int ntaps, float (&acc)[R]) {
    static_assert(R % 2 == 0, "outputs are processed in pairs");
    constexpr int P = R / 2;
    sdrm_v2 pa[P];
    const float *p[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        p[r] = xs + (r < valid ? r : 0) * stride;  // outputs past the last one re-read the first window (their sums are dropped)
    }
#pragma unroll
    for (int q = 0; q < P; q++) {
        pa[q] = sdrm_v2_make(acc[2 * q], acc[2 * q + 1]);
    }
    int j0 = 0;
    for (; j0 + K <= ntaps; j0 += K) {
#pragma unroll
        for (int u = 0; u < K; u++) {
            const float tp = taps[j0 + u];
#pragma unroll
            for (int q = 0; q < P; q++) {
                pa[q] = sdrm_v2_mac(pa[q], sdrm_v2_make(p[2 * q][j0 + u], p[2 * q + 1][j0 + u]), tp);
            }
        }
    }
    for (; j0 < ntaps; j0++) {
        const float tp = taps[j0];
#pragma unroll
        for (int q = 0; q < P; q++) {
            pa[q] = sdrm_v2_mac(pa[q], sdrm_v2_make(p[2 * q][j0], p[2 * q + 1][j0]), tp);
        }
    }
#pragma unroll
    for (int q = 0; q < P; q++) {
        acc[2 * q] = pa[q].x;
        acc[2 * q + 1] = pa[q].y;
    }
}

// values a K1 thread keeps in registers between phases
struct sdrm_k1_regs {
    sdrm_f2 y[SDRM_K1_R];
};

// phase 0: stage the raw tile (+halo) and the arctan table into LDS
SDRM_HD void sdrm_k1_phase_load(int tid, const sdrm_k1_tile &t, const sdrm_f2 *in, const sdrm_f2 *hist, int hist_len,
                                const float *atan_tab, sdrm_f2 *xs, float *tab) {
    // the first R + 1 samples of every thread (a whole tile with a halo of up to 256) are loaded before any is stored, so
    // that the loads are in flight together; longer halos take the plain loop
    constexpr int DEPTH = SDRM_K1_R + 1;
    sdrm_f2 v[DEPTH];
    if (t.x_first >= 0 && t.nx >= (DEPTH - 1) * SDRM_K1_THREADS) {
        // every tile but a call's first and last: the whole tile lies in the caller's buffer and only the last of the
        // DEPTH rows is partial -- one uniform base and constant offsets, no per-sample address arithmetic or selection
        // (the general form below spends ~9 vector instructions per load on them: 2.4 % of the kernel's)
        const sdrm_f2 *src = in + t.x_first + tid;
        const bool last = tid + (DEPTH - 1) * SDRM_K1_THREADS < t.nx;
#pragma unroll
        for (int i = 0; i < DEPTH - 1; i++) {
            v[i] = src[i * SDRM_K1_THREADS];
        }
        if (last) {
            v[DEPTH - 1] = src[(DEPTH - 1) * SDRM_K1_THREADS];
        }
#pragma unroll
        for (int i = 0; i < DEPTH - 1; i++) {
            xs[tid + i * SDRM_K1_THREADS] = v[i];
        }
        if (last) {
            xs[tid + (DEPTH - 1) * SDRM_K1_THREADS] = v[DEPTH - 1];
        }
    } else {
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            const int k = tid + i * SDRM_K1_THREADS;
            v[i].x = 0.0f;
            v[i].y = 0.0f;
            if (k < t.nx) {
                v[i] = sdrm_ext_sample(in, hist, hist_len, t.x_first + k);
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            const int k = tid + i * SDRM_K1_THREADS;
            if (k < t.nx) {
                xs[k] = v[i];
            }
        }
    }
    for (int k = tid + DEPTH * SDRM_K1_THREADS; k < t.nx; k += SDRM_K1_THREADS) {
        xs[k] = sdrm_ext_sample(in, hist, hist_len, t.x_first + k);
    }
    for (int k = tid; k < 257; k += SDRM_K1_THREADS) {
        tab[k] = atan_tab[k];
    }
}

// phase 1: LPF1 on R adjacent positions (reference src/dsp/fir_filter.c:123-144 via lpf.c:38-40)
SDRM_HD void sdrm_k1_phase_lpf1(int tid, const sdrm_k1_tile &t, const sdrm_chan_params &p, const float *taps1_rev,
                                const sdrm_f2 *xs, sdrm_f2 *bnd, sdrm_k1_regs &regs) {
#pragma unroll
    for (int r = 0; r < SDRM_K1_R; r++) {
        regs.y[r].x = 0.0f;
        regs.y[r].y = 0.0f;
    }
    if (tid * SDRM_K1_R < t.ny) {
        sdrm_fir_block_c<SDRM_K1_R, SDRM_K1_U>(xs + tid * SDRM_K1_R, taps1_rev, (int) p.T1, regs.y);
    }
    bnd[tid] = regs.y[SDRM_K1_R - 1];
}

// ---- quadrature demod, the short form (device only).
// sdrm_quad_sample_flat costs 41 vector instructions per sample, 11 of them one IEEE division.  The front-end is bound by
// vector-instruction issue, so this phase is rewritten for the common case and keeps the flat form as the fall-back:
//  * the division.  The compiler's expansion of a / b is v_div_scale x 2, v_rcp, SEVEN fused operations (one Newton step
//    on the reciprocal, the quotient, two residual corrections), v_div_fmas, v_div_fixup.  Its first and last three
//    instructions only act when an operand or the quotient is near the ends of the exponent range (v_div_scale leaves both
//    operands as they are unless the denominator is denormal, 1/b or a/b is denormal, the exponents are 96 or more apart or
//    the numerator is below 2^-103; v_div_fmas is then a plain FMA and v_div_fixup returns its first operand).  With
//    2^-60 <= a <= b <= 2^60 -- checked for the thread's samples at once on the bit patterns, 0.5 instructions per
//    sample and bound -- the seven fused operations on the unscaled operands ARE that division, and they pack: two
//    samples per v_pk_fma_f32 / v_pk_mul_f32 (component-wise, each rounded as the scalar instruction rounds).
//    a == 0 (a sample on an axis: real-valued test input, silence), tiny, huge, infinite and NaN operands fail the check;
//    the whole wave then takes the flat form for this tile.  The host emulation always takes the flat form: both compute
//    the correctly rounded quotient.
//  * x[n] conj(x[n-1]) is three packed instructions (a c, b c | b d, a d | sum with the second imaginary term negated)
//    instead of six: re = fl(fl(a c) + fl(b d)), im = fl(fl(b c) - fl(a d)) as in the reference (quadrature_demod.c:65).
//  * the table entry and the difference to its successor come from a table of pairs {tab[i], fl(tab[i+1] - tab[i])}
//    (the reference's subtraction, fast_atan2f.c:118, done once per workgroup instead of once per sample).
//  * octant fix-up: every octant's result is offset + (+-base) (see sdrm_fast_atan2f_flat); here the offset is
//    copysign(wide ? (x >= 0 ? 0 : pi) : pi/2, y >= 0 ? + : -).  That differs from the flat form's table in one entry, +0
//    instead of -0 for x >= 0, y >= 0, |x| > |y|, where the other term is +base with base >= +0: (-0) + (+0) and
//    (+0) + (+0) are both +0, and for base > 0 the zero's sign is irrelevant.  The `>= 0` tests stay float compares: the
//    reference treats -0 as non-negative (fast_atan2f.c:131-155), and x[n] conj(x[n-1]) does produce -0 (real input).
//  * the flat form's guard `ya > 0 || xa > 0` is true whenever the range check passes (b >= 2^-60).
#if defined(__HIP_DEVICE_COMPILE__)
#define SDRM_QUAD_LO_BITS 0x21800000u  // 2^-60
#define SDRM_QUAD_HI_BITS 0x5d800000u  // 2^60
SDRM_HD sdrm_v2 sdrm_v2_fma(sdrm_v2 a, sdrm_v2 b, sdrm_v2 c) { return __builtin_elementwise_fma(a, b, c); }

// x[n] conj(x[n-1]) = (a + ib)(c - id): (a c, b c), (b d, a d), then (a c + b d, b c - a d) -- three packed instructions
SDRM_HD sdrm_v2 sdrm_cmul_conj_pk(sdrm_v2 cur, sdrm_v2 pv) {
    sdrm_v2 p1, p2, ri;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p1) : "v"(cur), "v"(pv));
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(p2) : "v"(cur), "v"(pv));
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(ri) : "v"(p1), "v"(p2));
    return ri;
}

// q[r] for the thread's R samples; false when some operand is outside the range the short division covers
typedef const __attribute__((address_space(3))) float *sdrm_lds_cf;
template <int R>
SDRM_HD bool sdrm_quad_block_fast(const sdrm_f2 (&y)[R], sdrm_f2 prev, float gain, sdrm_lds_cf tab2, float (&q)[R]) {
    float re[R + 1], im[R + 1];
    sdrm_v2 pv = sdrm_v2_make(prev.x, prev.y);
#pragma unroll
    for (int r = 0; r < R; r++) {
        const sdrm_v2 cur = sdrm_v2_make(y[r].x, y[r].y);
        const sdrm_v2 ri = sdrm_cmul_conj_pk(cur, pv);
        re[r] = ri.x;
        im[r] = ri.y;
        pv = cur;
    }
    re[R] = re[R - 1];  // pad to whole pairs (R is odd)
    im[R] = im[R - 1];
    uint32_t lo = 0xffffffffu, hi = 0u;
#pragma unroll
    for (int r = 0; r < R + 1; r += 2) {
        float a[2], b[2];
        bool wide[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float xa = fabsf(re[r + k]), ya = fabsf(im[r + k]);
            wide[k] = xa > ya;
            a[k] = wide[k] ? ya : xa;
            b[k] = wide[k] ? xa : ya;
            lo = min(lo, sdrm_bits(a[k]));
            hi = max(hi, sdrm_bits(b[k]));
        }
        const sdrm_v2 A = sdrm_v2_make(a[0], a[1]), B = sdrm_v2_make(b[0], b[1]);
        const sdrm_v2 one = sdrm_v2_make(1.0f, 1.0f);
        const sdrm_v2 y0 = sdrm_v2_make(__builtin_amdgcn_rcpf(b[0]), __builtin_amdgcn_rcpf(b[1]));
        const sdrm_v2 e = sdrm_v2_fma(-B, y0, one);
        const sdrm_v2 y1 = sdrm_v2_fma(e, y0, y0);
        const sdrm_v2 q0 = A * y1;
        const sdrm_v2 r0 = sdrm_v2_fma(-B, q0, A);
        const sdrm_v2 q1 = sdrm_v2_fma(r0, y1, q0);
        const sdrm_v2 r1 = sdrm_v2_fma(-B, q1, A);
        const sdrm_v2 z = sdrm_v2_fma(r1, y1, q1);
        const sdrm_v2 al = z * sdrm_v2_make(255.0f, 255.0f);
        float base[2], off[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float zk = k ? z.y : z.x, alk = k ? al.y : al.x;
            int idx;
            asm("v_cvt_i32_f32 %0, %1" : "=v"(idx) : "v"(alk));
            const float frac = __builtin_amdgcn_fractf(alk);
            const float t0 = tab2[2 * idx], d = tab2[2 * idx + 1];
            const float interp = t0 + d * frac;
            base[k] = (zk < sdrm_from_bits(SDRM_TAN_MAP_RES_UP_BITS)) ? zk : interp;
            const bool xp = re[r + k] >= 0.0f, yp = im[r + k] >= 0.0f;
            const float pi_f = 3.14159265358979323846f, half_pi_f = 1.57079632679489661923f;
            const float mag = wide[k] ? (xp ? 0.0f : pi_f) : half_pi_f;
            off[k] = yp ? mag : -mag;
            const bool flip = (xp != yp) != !wide[k];
            base[k] = flip ? -base[k] : base[k];
        }
        const sdrm_v2 ang = sdrm_v2_make(off[0], off[1]) + sdrm_v2_make(base[0], base[1]);
        const sdrm_v2 out = sdrm_v2_make(gain, gain) * ang;
        q[r] = out.x;
        if (r + 1 < R) {
            q[r + 1] = out.y;
        }
    }
    return lo >= SDRM_QUAD_LO_BITS && hi <= SDRM_QUAD_HI_BITS;
}
#endif

// phase 2: quadrature demod (reference src/dsp/quadrature_demod.c:57-73) into LDS.  qs[-1] must be a valid slot: sample
// k of the tile goes to qs[k], and the tile's first thread also writes its k = -1 (the predecessor-less sample nobody
// reads); every thread stores all of its R samples -- positions past the tile's last needed sample (nq) hold finite
// values nobody reads either, and NY - 2 < NY + QPAD -- so the stores are one base address plus constants and no
// per-sample predicate (which also made the compiler sink half of the arctangent, with its constants re-made per sample,
// into fifteen predicated blocks: 51 -> 40 vector instructions per sample).
// tab2 (device only; nullptr on the host): the table of {entry, difference} pairs of the short form above.
SDRM_HD void sdrm_k1_phase_quad(int tid, const sdrm_k1_tile &t, const sdrm_chan_params &p, const float *tab, const float *tab2,
                                const sdrm_f2 *bnd, const sdrm_k1_regs &regs, float *qs) {
    (void) t;
    (void) tab2;
    // the tile's first thread has no predecessor in the tile: its sample k = -1 is the one nobody reads, and its value
    // does not matter -- (1, 0) rather than zero keeps it inside the short form's range (a zero product would send the
    // wave, a quarter of all waves, to the general form)
    sdrm_f2 prev;
    prev.x = 1.0f;
    prev.y = 0.0f;
    if (tid > 0) {
        prev = bnd[tid - 1];
    }
#if defined(__HIP_DEVICE_COMPILE__)
    int zero = 0;
    asm volatile("" : "+s"(zero));  // keeps the table's LDS address one scalar (else: base + constant, an add per sample)
    tab += zero;
#endif
    float q[SDRM_K1_R];
    float *dst = qs + tid * SDRM_K1_R - 1;
#if defined(__HIP_DEVICE_COMPILE__)
    // LDS-typed, opaque copy of the pair table's address: one scalar, reached with the index shifted in (v_lshl_add)
    sdrm_lds_cf tab2_l = (sdrm_lds_cf) tab2;
    asm volatile("" : "+s"(tab2_l));
    if (tab2 != nullptr && __all(sdrm_quad_block_fast<SDRM_K1_R>(regs.y, prev, p.quad_gain, tab2_l, q))) {
#pragma unroll
        for (int r = 0; r < SDRM_K1_R; r++) {
            dst[r] = q[r];
        }
        return;
    }
#endif
    // all of the thread's samples (threads past the tile's end hold zeros): fifteen independent chains the compiler can
    // interleave, so that the table reads and the reciprocals wait for each other's work
#pragma unroll
    for (int r = 0; r < SDRM_K1_R; r++) {
        sdrm_f2 cur = regs.y[r];
#if defined(__HIP_DEVICE_COMPILE__)
        // the LPF1 accumulators stay (re, im) register pairs: left to itself the compiler vectorises this phase ACROSS
        // samples and pays for the layout it then wants with ~70 register moves inside the LPF1 tap loop
        sdrm_v2 pair = sdrm_v2_make(cur.x, cur.y);
        asm volatile("" : "+v"(pair));
        cur.x = pair.x;
        cur.y = pair.y;
#endif
        q[r] = sdrm_quad_sample_flat(cur, prev, p.quad_gain, tab);
        prev = cur;
    }
#pragma unroll
    for (int r = 0; r < SDRM_K1_R; r++) {
        dst[r] = q[r];
    }
}

// phase 3: LPF2 with decimation (reference src/dsp/fir_filter.c:93-114), results to the tile's staging area
template <int R>
SDRM_HD bool sdrm_k1_lpf2_decimated(int tid, const sdrm_k1_tile &t, int d, int ntaps, const float *taps2_rev, const float *qs, float *zs,
                                    float tame) {
    bool odd = false;
    // thread tid takes the outputs tid + k * THREADS (k = 0, 1, ...), R of them per pass
    for (int o0 = tid; o0 < t.m; o0 += R * SDRM_K1_THREADS) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            acc[r] = 0.0f;
        }
        const int valid = (t.m - o0 + SDRM_K1_THREADS - 1) / SDRM_K1_THREADS;  // outputs o0 + r * THREADS below m
        sdrm_fir_block_rd<R, SDRM_K1_U>(qs + o0 * d, SDRM_K1_THREADS * d, valid, taps2_rev, ntaps, acc);
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int o = o0 + r * SDRM_K1_THREADS;
            if (o < t.m) {
                zs[o] = acc[r];
                odd |= !(fabsf(acc[r]) < tame);
            }
        }
    }
    return odd;
}

// The level a stage compares its outputs with: the channel's amp_safe on the stream the clock stage reads (the front-end's
// without a DC blocker, the DC stage's with one), +Inf elsewhere.  A channel whose loop is not tame at ANY amplitude
// (amp_safe <= 0: fewer than ~1.01 samples per symbol) is wild by configuration -- the clock stage knows, nothing to flag.
SDRM_HD float sdrm_tame_level(const sdrm_chan_params &p, bool dc_stage) {
    return ((p.dc_len != 0) == dc_stage && p.amp_safe > 0.0f) ? p.amp_safe : INFINITY;
}
// which of SDRM_FLAG_NONFINITE / SDRM_FLAG_WILD a sample of the clock stage's input raises (Inf raises both; NaN only the first)
SDRM_HD uint32_t sdrm_flag_bits(float v, float tame) {
    return (!(fabsf(v) < INFINITY) ? SDRM_FLAG_NONFINITE : 0u) | ((fabsf(v) >= tame) ? SDRM_FLAG_WILD : 0u);
}
SDRM_HD void sdrm_flag_raise(uint32_t *flag, uint32_t bits) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(flag, bits);
#else
    *flag |= bits;
#endif
}

SDRM_HD void sdrm_k1_phase_lpf2(int tid, const sdrm_k1_tile &t, const sdrm_chan_params &p, const float *taps2_rev,
                                const float *qs, float *zs, uint32_t *nonfinite_flag) {
    bool odd = false;
    // the one compare per output that used to look for NaN/Inf also looks for samples the timing loop is not provably tame
    // with (SDRM_FLAG_WILD) -- when this IS the clock stage's input; behind a DC blocker that stage looks at its own output
    const float tame = sdrm_tame_level(p, false);
    if (p.decim == 1) {
        const int base = tid * SDRM_K1_RZ;
        if (base >= t.m) {
            return;
        }
        float acc[SDRM_K1_RZ];
#pragma unroll
        for (int r = 0; r < SDRM_K1_RZ; r++) {
            acc[r] = 0.0f;
        }
        sdrm_fir_block_rp<SDRM_K1_RZ, SDRM_K1_U>(qs + base, taps2_rev, (int) p.T2, acc);
        // results go to the tile's staging area (lane stride 15 floats: conflict-free) and leave in sdrm_k1_phase_store
#pragma unroll
        for (int r = 0; r < SDRM_K1_RZ; r++) {
            if (base + r < t.m) {
                zs[base + r] = acc[r];
                odd |= !(fabsf(acc[r]) < tame);
            }
        }
        if (odd) {  // rare: say which (the thread's own outputs, from the staging area)
            uint32_t bits = 0;
            for (int r = 0; r < SDRM_K1_RZ && base + r < t.m; r++) {
                bits |= sdrm_flag_bits(zs[base + r], tame);
            }
            sdrm_flag_raise(nonfinite_flag, bits);
        }
        return;
    } else if (t.m <= 2 * SDRM_K1_THREADS) {
        odd = sdrm_k1_lpf2_decimated<2>(tid, t, (int) p.decim, (int) p.T2, taps2_rev, qs, zs, tame);
    } else {
        odd = sdrm_k1_lpf2_decimated<4>(tid, t, (int) p.decim, (int) p.T2, taps2_rev, qs, zs, tame);
    }
    if (odd) {
        uint32_t bits = 0;
        for (int o = tid; o < t.m; o += SDRM_K1_THREADS) {
            bits |= sdrm_flag_bits(zs[o], tame);
        }
        sdrm_flag_raise(nonfinite_flag, bits);
    }
}

// the tile's LPF2 outputs, staged by sdrm_k1_phase_lpf2, written with consecutive lanes on consecutive samples
// THROUGH (device code, in-call hand-off): device-scope stores, written through this XCD's L2 -- a stage that is already
// running on another XCD reads them with the same scope as soon as they are acknowledged; no cache write-back needed
template <bool THROUGH>
SDRM_HD void sdrm_store_out(float *dst, float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (THROUGH) {
        __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#endif
    *dst = v;
}
template <bool THROUGH = false>
SDRM_HD void sdrm_k1_phase_store(int tid, const sdrm_k1_tile &t, const float *zs, float *z_out) {
    float *dst = z_out + t.o_lo + tid;
    const float *src = zs + tid;
    constexpr int FULL = SDRM_K1_RZ - 1;  // rows of 256 outputs a full tile certainly has (it has NY - (T2 - 1) - 1 outputs)
    if (t.m >= FULL * SDRM_K1_THREADS) {
        // constant offsets from one base: no per-sample address arithmetic
        float v[FULL];
#pragma unroll
        for (int i = 0; i < FULL; i++) {
            v[i] = src[i * SDRM_K1_THREADS];
        }
#pragma unroll
        for (int i = 0; i < FULL; i++) {
            sdrm_store_out<THROUGH>(dst + i * SDRM_K1_THREADS, v[i]);
        }
        for (int i = tid + FULL * SDRM_K1_THREADS; i < t.m; i += SDRM_K1_THREADS) {
            sdrm_store_out<THROUGH>(z_out + t.o_lo + i, zs[i]);
        }
        return;
    }
    for (int i = tid; i < t.m; i += SDRM_K1_THREADS) {
        sdrm_store_out<THROUGH>(z_out + t.o_lo + i, zs[i]);
    }
}

// K1h: next call's history = the last hist_len samples of (history ++ input)
SDRM_HD void sdrm_hist_roll(int tid, int nthreads, const sdrm_chan_params &p, const sdrm_chunk_ctl &c, const sdrm_f2 *in,
                            const sdrm_f2 *hist_cur, sdrm_f2 *hist_next) {
    const int H = (int) p.hist_len;
    for (int j = tid; j < H; j += nthreads) {
        hist_next[j] = sdrm_ext_sample(in, hist_cur, H, (int) c.n_in - H + j);
    }
}

// ------------------------------------------------------------------------------------------------ K2

// DC blocker (reference src/dsp/dc_blocker.c:56-64, 105-119): four cascaded length-L boxcars, each
//   t = u[n] - u[n-L];  acc = acc + t;  v[n] = acc / L          (acc: a strictly sequential fp32 recursion)
// and out[n] = x[n - 2(L-1)] - v3[n].  Everything except `acc = acc + t` is pointwise.
//
// One workgroup serves up to 16 channels ("slots") with 1 + 5 x (16 / P) waves (eleven with P = 8):
//   chain wave   lane = (stage, slot): reads the 64 terms of its block from its LDS row (16 x ds_read_b128), adds them
//                one by one to its running sum and leaves a checkpoint every P terms {sum before the block, after P, 2P, ..};
//   feeder       terms of stage 0 from the front-end's output: t = x[n] - x[n-L];
//   3 x stage    for stage s -> s+1: every lane takes P consecutive samples of one slot, rebuilds the running sums from
//                its checkpoint with the SAME additions in the same order, divides, appends the quotients to the stage's
//                delay ring in LDS, reads the delayed ones and writes the next stage's terms;
//   output       the same for stage 3, then x[n - 2(L-1)] - v3[n] to global memory.
// (a helper role is 16 / P waves: a wave covers P slots with 64 / P lanes each)
// The stages run two blocks apart (stage s sums block it - 2s in iteration `it`, the helpers convert it in iteration
// it + 1), one barrier per iteration; a row keeps three blocks (being written / being summed / being read).
// A lone wave pays per instruction, not per lane: the chain wave's 64 additions serve 4 stages x 16 channels, where the
// earlier in-order DPP chain spent 63 wave instructions on 64 samples of ONE stage of ONE channel.
#define SDRM_K2_SLOTS 16
#define SDRM_K2_BLK 64
#ifndef SDRM_K2_P
#define SDRM_K2_P 8       // consecutive samples per helper lane (16: one wave per helper role, 8: two; measured 1.37 / 1.14 ms)
#endif
#define SDRM_K2_LPS (SDRM_K2_BLK / SDRM_K2_P)    // helper lanes per slot and block = checkpoints per row and block
#define SDRM_K2_WPR (SDRM_K2_SLOTS / SDRM_K2_P)  // waves per helper role (a wave covers P slots)
#define SDRM_K2_NBUF 3
#define SDRM_K2_TSPITCH (SDRM_K2_NBUF * SDRM_K2_BLK + 4)  // floats per row: 49 sixteen-byte units (odd: conflict-free ds_read_b128 across rows)
#define SDRM_K2_MIRROR 16  // the first 16 ring slots are repeated behind the ring: 16 delayed samples never wrap
#define SDRM_K2_WAVES (1 + 5 * SDRM_K2_WPR)  // chain, feeder, three stage helpers, output
#define SDRM_K2_ROWS (4 * SDRM_K2_SLOTS)

// per-slot constants (in LDS on the device)
struct sdrm_k2_slot {
    int chan;            // -1: no channel in this slot (or no DC blocker / absent from the call)
    uint32_t L;          // boxcar length
    uint32_t A;          // L rounded up to a block: ring slot of call-relative sample n is (A + n) mod rcap
    uint32_t rcap;       // A + 64
    uint32_t nz;         // samples of this call
    uint32_t HX;         // 2 (L - 1): samples of x carried between calls
    float Lf, invL;
    int alias;           // 1: a REPLICA of another slot of the group (sdrm_k2_fill_aliases): computes the same, saves nothing
};

// A slot of the group without a channel in this call (a batcher slot without a buffer this round, or waiting for a client)
// becomes a replica of the group's longest live slot: it computes exactly what that slot computes -- same channel, same state
// read at the start, same stores of the same values -- so that the group's waves keep their straight-line code, which needs
// every lane to have a channel (a 512-slot batcher with 3 live clients ran its DC stage in 2.6 ms instead of 1.16:
// profiles/r05_node_schedule.txt).  What a replica must NOT do is save state that its original reads while saving (the carried
// samples of x): the savers skip replicas.  Slots at or beyond n_use (the group's ring space) stay empty.
SDRM_HD void sdrm_k2_fill_aliases(sdrm_k2_slot *slots, int n_use) {
    int best = -1;
    for (int i = 0; i < n_use; i++) {
        if (slots[i].chan >= 0 && (best < 0 || slots[i].nz > slots[best].nz)) {
            best = i;
        }
    }
    for (int i = 0; i < n_use && best >= 0; i++) {
        if (slots[i].chan < 0) {
            slots[i] = slots[best];
            slots[i].alias = 1;
        }
    }
}

// DC state of a channel in global memory: hx[hx_cap] (the last 2(L-1) samples of x, oldest first, at the FRONT of the
// array), three tails of l_cap floats (the last L quotients of stages 0..2, oldest first), four running sums
SDRM_HD float *sdrm_k2_state_tail(float *st, int ring, uint32_t hx_cap, uint32_t l_cap) { return st + hx_cap + (size_t) ring * l_cap; }
SDRM_HD float *sdrm_k2_state_acc(float *st, uint32_t hx_cap, uint32_t l_cap) { return st + hx_cap + 3 * (size_t) l_cap; }
SDRM_HD size_t sdrm_k2_state_floats(uint32_t hx_cap, uint32_t l_cap) { return (size_t) hx_cap + 3 * (size_t) l_cap + 8; }

// floats between two rings in LDS (odd number of 16-byte units)
SDRM_HD uint32_t sdrm_k2_ring_pitch(uint32_t rcap_max) { return rcap_max + SDRM_K2_MIRROR + 4; }

SDRM_HD void sdrm_k2_slot_setup(sdrm_k2_slot &s, int chan, const sdrm_chan_params &p, uint32_t nz) {
    s.alias = 0;
    s.chan = chan;
    s.L = p.dc_len;
    s.A = (p.dc_len + SDRM_K2_BLK - 1) / SDRM_K2_BLK * SDRM_K2_BLK;
    s.rcap = s.A + SDRM_K2_BLK;
    s.nz = nz;
    s.HX = 2 * (p.dc_len - 1);
    s.Lf = p.dc_len_f;
    s.invL = p.dc_inv_len;
}

// x[n] of this call: the front-end's output for n >= 0, the carried samples before
SDRM_HD float sdrm_k2_x(const float *z, const float *hx, uint32_t HX, int n) { return n >= 0 ? z[n] : hx[(int) HX + n]; }

// ring <- tail at the start of a call: the sample L - j before the call sits in slot A - L + j
SDRM_HD void sdrm_k2_ring_load(float *ring, const sdrm_k2_slot &s, const float *tail, int tid, int nthreads) {
    for (uint32_t j = tid; j < s.L; j += nthreads) {
        const uint32_t slot = s.A - s.L + j;
        ring[slot] = tail[j];
        if (slot < SDRM_K2_MIRROR) {
            ring[s.rcap + slot] = tail[j];
        }
    }
}

// tail <- ring at the end: the last L samples, i.e. call-relative nz - L .. nz - 1
SDRM_HD void sdrm_k2_ring_save(const float *ring, const sdrm_k2_slot &s, float *tail, int tid, int nthreads) {
    for (uint32_t j = tid; j < s.L; j += nthreads) {
        tail[j] = ring[(s.A + s.nz - s.L + j) % s.rcap];
    }
}

// ring slot of the first sample of block k
SDRM_HD uint32_t sdrm_k2_block_slot(const sdrm_k2_slot &s, int k) { return (uint32_t) (((uint64_t) s.A + (uint64_t) k * SDRM_K2_BLK) % s.rcap); }

// ---- chain wave: row = (stage, slot).  64 additions in order; checkpoints {before, after P, 2P, ..}; returns the sum after 64
SDRM_HD float sdrm_k2_chain_block(const float *row_buf, float *check, float acc) {
    for (int g = 0; g < SDRM_K2_LPS; g++) {
        check[g] = acc;
        for (int i = 0; i < SDRM_K2_P; i++) {
            acc = acc + row_buf[SDRM_K2_P * g + i];
        }
    }
    return acc;
}

// ---- feeder: the P terms of stage 0 a lane owns in block k (q = lane % LPS): t = x[n] - x[n - L], 0 beyond the call's end
SDRM_HD void sdrm_k2_feed(const sdrm_k2_slot &s, int k, int q, const float *z, const float *hx, float *row_buf) {
    const int n0 = k * SDRM_K2_BLK + q * SDRM_K2_P;
    for (int i = 0; i < SDRM_K2_P; i++) {
        const int n = n0 + i;
        float t = 0.0f;
        if ((uint32_t) n < s.nz) {
            t = sdrm_boxcar_term(z[n], sdrm_k2_x(z, hx, s.HX, n - (int) s.L));
        }
        row_buf[q * SDRM_K2_P + i] = t;
    }
}

// the P quotients a lane owns: running sums rebuilt from the checkpoint, then sums / L
SDRM_HD void sdrm_k2_quotients(const sdrm_k2_slot &s, const float *row_buf, float check, int q, float (&v)[SDRM_K2_P]) {
    float acc = check;
    float sums[SDRM_K2_P];
    bool any_unsafe = false;
    for (int i = 0; i < SDRM_K2_P; i++) {
        acc = acc + row_buf[q * SDRM_K2_P + i];
        sums[i] = acc;
        bool unsafe;
        v[i] = sdrm_boxcar_out_fast(acc, s.Lf, s.invL, &unsafe);
        any_unsafe |= unsafe;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    any_unsafe = __any(any_unsafe);
#endif
    if (any_unsafe) {  // denormal / infinite / NaN quotients: the division proper (the same values wherever the short form is valid)
        for (int i = 0; i < SDRM_K2_P; i++) {
            v[i] = sdrm_boxcar_out(sums[i], s.Lf);
        }
    }
}

// ---- stage s -> s+1: quotients into the delay ring, delayed quotients out of it, terms of the next stage
SDRM_HD void sdrm_k2_transition(const sdrm_k2_slot &s, int k, int q, const float *in_buf, float check, float *ring, float *out_buf) {
    float v[SDRM_K2_P];
    sdrm_k2_quotients(s, in_buf, check, q, v);
    // Lanes past the channel's end leave the ring alone (another channel of the group may have many more blocks, and
    // their garbage would run round the ring into the samples the next call needs); a lane that holds the end writes
    // fewer than P slots beyond it, which the ring's 64 spare slots absorb.
    if ((uint32_t) (k * SDRM_K2_BLK + q * SDRM_K2_P) >= s.nz) {
        for (int i = 0; i < SDRM_K2_P; i++) {
            out_buf[q * SDRM_K2_P + i] = 0.0f;
        }
        return;
    }
    const uint32_t base = sdrm_k2_block_slot(s, k) + (uint32_t) q * SDRM_K2_P;  // < rcap, a multiple of P
    for (int i = 0; i < SDRM_K2_P; i++) {
        ring[base + i] = v[i];
    }
    if (base < SDRM_K2_MIRROR) {
        for (int i = 0; i < SDRM_K2_P; i++) {
            ring[s.rcap + base + i] = v[i];
        }
    }
    const uint32_t from = (base + s.rcap - s.L) % s.rcap;  // from + P - 1 < rcap + MIRROR
    const int n0 = k * SDRM_K2_BLK + q * SDRM_K2_P;
    for (int i = 0; i < SDRM_K2_P; i++) {
        const float ud = ring[from + i];
        out_buf[q * SDRM_K2_P + i] = ((uint32_t) (n0 + i) < s.nz) ? sdrm_boxcar_term(v[i], ud) : 0.0f;
    }
}

// ---- output: x[n - 2(L-1)] - v3[n]; returns the flags its results raise (SDRM_FLAG_NONFINITE / SDRM_FLAG_WILD)
SDRM_HD uint32_t sdrm_k2_output(const sdrm_k2_slot &s, int k, int q, const float *in_buf, float check, const float *z, const float *hx,
                                float *out, float tame) {
    float v[SDRM_K2_P];
    sdrm_k2_quotients(s, in_buf, check, q, v);
    const int n0 = k * SDRM_K2_BLK + q * SDRM_K2_P;
    uint32_t bits = 0;
    for (int i = 0; i < SDRM_K2_P; i++) {
        const int n = n0 + i;
        if ((uint32_t) n < s.nz) {
            const float o = sdrm_k2_x(z, hx, s.HX, n - (int) s.HX) - v[i];
            out[n] = o;
            bits |= sdrm_flag_bits(o, tame);
        }
    }
    return bits;
}

// new hx = the last HX samples of (hx ++ z[0 .. nz)): element j for j in [j0, j1) (the caller orders reads before writes)
SDRM_HD float sdrm_k2_hx_source(const sdrm_k2_slot &s, const float *z, const float *hx, uint32_t j) {
    return sdrm_k2_x(z, hx, s.HX, (int) s.nz - (int) s.HX + (int) j);
}

// ------------------------------------------------------------------------------------------------ K3

// per-lane context of the clock-recovery kernel; col = this channel's sample ring in LDS (ring[channel][slot]).
// Ring slot of chunk-relative sample n (n < 0: carried history) is n & (RING-1).
struct sdrm_k3_lane {
    sdrm_mm_state st;
    sdrm_mm_consts k;
    int kept;          // carried samples at the start of the call
    int nz;            // new samples this call
    uint32_t oo;       // symbols produced so far
    uint32_t cap;      // max symbols per call (= max_input_buffer_length, clock_recovery_mm.c:103)
};

// `col` = this channel's ring (ring + channel * CPITCH).  Pair layout: element e = {x[e], x[e+1]} lives at
// col[2 * (slot + PRE)], slot = e & (RING-1); plain layout: sample n lives at col[slot + PRE].  Slots < POST are
// mirrored above the ring and slots >= RING-PRE below it, so the samples [slot-3, slot+7] around any slot are contiguous:
// a symbol's samples are one base address plus constant offsets.
template <typename G>
SDRM_HD void sdrm_k3_elem_put(float *col, int slot, int half, float v) {
    col[2 * (slot + SDRM_K3_PRE) + half] = v;
    if (slot < SDRM_K3_POST) {
        col[2 * (slot + G::ring + SDRM_K3_PRE) + half] = v;
    }
    if (slot >= G::ring - SDRM_K3_PRE) {
        col[2 * (slot - G::ring + SDRM_K3_PRE) + half] = v;
    }
}

// pair layout: sample n is the first half of its own element and the second half of its predecessor's
template <typename G>
SDRM_HD void sdrm_k3_ring_put(float *col, int n, float v) {
    if (G::plain) {
        const int slot = n & (G::ring - 1);
        col[slot + SDRM_K3_PRE] = v;
        if (slot < SDRM_K3_POST) {
            col[slot + G::ring + SDRM_K3_PRE] = v;
        }
        if (slot >= G::ring - SDRM_K3_PRE) {
            col[slot - G::ring + SDRM_K3_PRE] = v;
        }
    } else {
        sdrm_k3_elem_put<G>(col, n & (G::ring - 1), 0, v);
        sdrm_k3_elem_put<G>(col, (n - 1) & (G::ring - 1), 1, v);
    }
}

template <typename G>
SDRM_HD float sdrm_k3_ring_get(const float *col, int n) {
    return G::plain ? col[(n & (G::ring - 1)) + SDRM_K3_PRE] : col[2 * ((n & (G::ring - 1)) + SDRM_K3_PRE)];
}

// The loop condition `ii < working_len - 7` (clock_recovery_mm.c:103, ii compared as size_t: negative => stop) with
// `avail` chunk samples staged, as ONE unsigned compare against a per-block limit: the window starts at chunk-relative
// ii - kept and needs +7 < avail, i.e. 0 <= ii < avail + kept - 7.
SDRM_HD uint32_t sdrm_k3_limit(const sdrm_k3_lane &L, int avail) {
    const int lim = avail + L.kept - 7;
    return lim > 0 ? (uint32_t) lim : 0u;
}

// can this lane produce its next symbol?  (`&& oo < output_len` of the same loop condition)
SDRM_HD bool sdrm_k3_can_step(const sdrm_k3_lane &L, uint32_t limit) {
    return ((uint32_t) L.st.ii < limit) & (L.oo < L.cap);
}

// everything one symbol reads from LDS: fetched for the lane's current (ii, mu) BEFORE the symbol is computed, so
// that the fetch for the next symbol can be issued while the current one is still being quantised and stored
struct sdrm_k3_operands {
    float w[8];     // window samples ii .. ii+7
    float lead[3];  // samples ii-3 .. ii-1 (only the general form looks at them)
    float tap[8];   // MMSE bank row for mu, reversed: tap[j] multiplies w[j]
    bool row_ok;    // false when mu is NaN (the reference indexes out of bounds there)
};

// reference src/dsp/mmse_fir_interpolator.c:189: row = rint(mu * 128) (mu*128 in fp32, half-to-even)
// (RingPtr / BankPtr: plain `const float *` on the host, LDS-address-space pointers in the kernel)
template <bool FINITE, typename G, typename RingPtr, typename BankPtr>
SDRM_HD void sdrm_k3_fetch(const sdrm_k3_lane &L, RingPtr col, BankPtr bank_rev, sdrm_k3_operands &F) {
    const int n = L.st.ii - L.kept;
    if (G::plain) {
        const RingPtr base = col + ((n & (G::ring - 1)) + SDRM_K3_PRE);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            F.w[j] = base[j];
        }
        if (!FINITE) {
            F.lead[0] = base[-3];
            F.lead[1] = base[-2];
            F.lead[2] = base[-1];
        }
    } else {
        const RingPtr base = col + 2 * ((n & (G::ring - 1)) + SDRM_K3_PRE);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {  // elements s, s+2, s+4, s+6
            F.w[j] = base[2 * j];
            F.w[j + 1] = base[2 * j + 1];
        }
        if (!FINITE) {
            F.lead[0] = base[-6];  // x[s-3], x[s-2] = element s-3; x[s-1] = first half of element s-1
            F.lead[1] = base[-5];
            F.lead[2] = base[-2];
        }
    }
    const float scaled = L.st.mu * (float) SDRM_MMSE_STEPS;
    int imu;
    F.row_ok = true;
    if (FINITE) {
        // mu in [0,1] => 0..128.  mu * 128 is exact (power of two), so the explicit fused multiply-add rounds once,
        // exactly where rint(mu * 128) does
        imu = (int) (sdrm_bits(fmaf(L.st.mu, (float) SDRM_MMSE_STEPS, SDRM_RINT_MAGIC)) & 0xffu);
    } else {
        F.row_ok = (scaled >= 0.0f) & (scaled <= (float) SDRM_MMSE_STEPS);  // false for NaN
        imu = F.row_ok ? (int) rintf(scaled) : 0;
    }
    const BankPtr row = bank_rev + imu * SDRM_K3_BANKPITCH;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        F.tap[j] = row[j];
    }
}

// The exact form of the reference's 16-byte aligned dot product (src/dsp/fir_filter.c:116-121): the `nlead` = ii & 3
// samples in front of the window are multiplied by zero taps and summed first.
SDRM_HD float sdrm_k3_lead_exact(const float *lead, int nlead) {
    float acc = 0.0f;
    if (nlead >= 3) acc = acc + lead[0] * 0.0f;
    if (nlead >= 2) acc = acc + lead[1] * 0.0f;
    if (nlead >= 1) acc = acc + lead[2] * 0.0f;
    return acc;
}

// One symbol (reference src/dsp/clock_recovery_mm.c:103-125 with mmse_fir_interpolator.c:188-191 inlined) from
// operands fetched by sdrm_k3_fetch for the lane's current state.
// FINITE: the caller guarantees every sample this lane can touch and the loop state are finite (no NaN/Inf entered
// the stream): then the interpolator output is finite, the NaN branch and the zero-tap lead samples cannot matter,
// mu stays in [0,1) and the float->int conversions are in range -- the same values with fewer instructions.
template <bool FINITE>
SDRM_HD float sdrm_k3_step(sdrm_k3_lane &L, const sdrm_k3_operands &F) {
    const int ii = L.st.ii;
    const float mu = L.st.mu;
    float acc = 0.0f;
    if (!FINITE) {
        // leading zero-tap samples only matter when one of them is NaN/Inf (x*0 = NaN); test all three at once
        float probe = F.lead[0] * 0.0f;
        probe = probe + F.lead[1] * 0.0f;
        probe = probe + F.lead[2] * 0.0f;
        if (probe != probe) {
            acc = sdrm_k3_lead_exact(F.lead, ii & 3);
        }
    }
    acc = acc + F.w[0] * F.tap[0];
    acc = acc + F.w[1] * F.tap[1];
    acc = acc + F.w[2] * F.tap[2];
    acc = acc + F.w[3] * F.tap[3];
    acc = acc + F.w[4] * F.tap[4];
    acc = acc + F.w[5] * F.tap[5];
    acc = acc + F.w[6] * F.tap[6];
    acc = acc + F.w[7] * F.tap[7];
    // the reference indexes out of bounds when mu is NaN; defined as a NaN symbol here
    const float o = (FINITE || F.row_ok) ? acc : NAN;
    // regular symbol (clock_recovery_mm.c:115-123)
    const float last = L.st.last;
    float a, bneg;
    if (FINITE) {
        // Both products take the sign sign(o) ^ sign(last).  The sign bit is the `< 0` test here: a finite dot product
        // accumulated from +0 is never -0 (x + y is -0 only if both are), so neither o nor last can be -0 or NaN.
        const uint32_t bo = sdrm_bits(o), bl = sdrm_bits(last), x = bo ^ bl;
        a = sdrm_from_bits(sdrm_bfi(0x7fffffffu, bo, x));
        bneg = sdrm_from_bits(sdrm_bfi(0x7fffffffu, bl, x));
    } else {
        a = (last < 0.0f) ? -o : o;        // slice(last) * o
        bneg = (o < 0.0f) ? -last : last;  // slice(o) * last
    }
    const float mm = a - bneg;
    float om = L.st.omega + L.k.gain_omega * mm;
    const float dev = om - L.k.omega_mid;
    const float clipped = 0.5f * (fabsf(dev + L.k.omega_lim) - fabsf(dev - L.k.omega_lim));
    om = L.k.omega_mid + clipped;
    const float m2 = L.st.mu + om + L.k.gain_mu * mm;
    const float whole = floorf(m2);
    if (FINITE) {
        L.st.omega = om;
        L.st.mu = m2 - whole;
        L.st.last = o;
        L.st.inc = (int) whole;
        L.st.ii = ii + L.st.inc;
        return o;
    }
    // NaN symbol (:107-113): emit 0, skip floor(omega) samples, leave the loop state alone
    const bool bad = o != o;
    const float skip = floorf(L.st.omega);
    const int step = bad ? sdrm_cvt_i32(skip) : sdrm_cvt_i32(whole);
    L.st.omega = bad ? L.st.omega : om;
    L.st.mu = bad ? mu : m2 - whole;
    L.st.last = bad ? last : o;
    L.st.inc = step;
    L.st.ii = (int) ((uint32_t) ii + (uint32_t) step);
    return bad ? 0.0f : o;
}

// after the last block: decide what to carry (clock_recovery_mm.c:127-135). Returns the chunk-relative index of
// the first carried sample and the new kept count.
SDRM_HD void sdrm_k3_finish(const sdrm_k3_lane &L, int *from_n, int *new_kept) {
    const int64_t len = (int64_t) L.kept + L.nz;
    int64_t from;
    if (len < SDRM_MMSE_TAPS) {
        from = 0;  // :94-99 keep everything, no symbols
    } else {
        const bool past = (L.st.ii < 0) || ((int64_t) L.st.ii > len);  // `ii > working_len` as size_t
        const int prev = (int) ((uint32_t) L.st.ii - (uint32_t) L.st.inc);  // position of the last produced symbol
        from = past ? (int64_t) prev : (int64_t) L.st.ii;
    }
    int64_t keep = len - from;
    if (keep > SDRM_CLOCK_HCAP - 1) {  // bounded where the reference overruns its buffer
        keep = SDRM_CLOCK_HCAP - 1;
        from = len - keep;
    }
    if (keep < 0) {
        keep = 0;
        from = len;
    }
    *from_n = (int) (from - L.kept);
    *new_kept = (int) keep;
}

// ------------------------------------------------------------------------------------------------ generic channels
// A channel whose symbols are longer than the LDS-resident stages are sized for (more than ~244 samples per symbol: the clock
// stage's rings hold at most 255 carried samples; a DC boxcar beyond 7712 samples does not fit a CU's LDS) keeps the fast
// front-end and runs its DC blocker and clock recovery in these forms: all state in global memory, the reference's arithmetic
// statement by statement (IEEE division, the NaN-aware symbol), one workgroup of 64 threads per channel.  Such a channel has
// at most a few hundred symbols per call, so the stages' cost is what the DC blocker's four running sums cost: ~8 ms per
// 131072-sample call, 1.5 % of the time such a signal takes to arrive.  Reference: src/dsp/dc_blocker.c:56-64,105-119,
// src/dsp/clock_recovery_mm.c:78-139 -- which accept any samples per symbol (src/dsp/fsk_demod.c:53-63).
//
// Layout of a channel's state (floats):
//   [0..3] running sums of the four boxcars   [4] ring position of the boxcars' delay lines (uint32 bits)
//   [5] ring position of the output delay (uint32 bits)   [6..7] unused
//   ring[s], s = 0..3: the last L inputs of boxcar s      xring: the last 2 (L - 1) inputs of the blocker
//   work: 3 floats of padding (what an aligned dot product reads in front of position 0: zero taps), then the clock stage's
//         working buffer = carried samples followed by the call's new ones (clock_recovery_mm.c:90)
struct sdrm_gen_layout {
    uint32_t L, XL;    // boxcar length (0: no DC blocker) and output delay 2 (L - 1)
    uint32_t hcap;     // most samples the clock stage carries between calls (< 1.01 samples per symbol + 8)
    uint32_t wcap;     // working buffer: hcap + the most new samples of a call + 8
    uint32_t off_ring, off_x, off_work, total;
};
SDRM_HD sdrm_gen_layout sdrm_gen_layout_for(uint32_t dc_len, float sps, uint32_t max_len, uint32_t decim) {
    sdrm_gen_layout g;
    g.L = dc_len;
    g.XL = dc_len ? 2u * (dc_len - 1u) : 0u;
    g.hcap = (uint32_t) (sps * 1.01f) + 24u;
    g.wcap = g.hcap + max_len / (decim ? decim : 1u) + 16u;
    g.off_ring = 8u;
    g.off_x = g.off_ring + 4u * g.L;
    g.off_work = (g.off_x + g.XL + 3u) & ~3u;
    g.total = g.off_work + 4u + g.wcap;
    return g;
}

// the clock stage's view of a generic channel's working buffer: a ring that never wraps (positions count from the start of
// the carried samples, as the reference's ii does), plain samples behind three floats of padding
struct sdrm_k3_geom_linear {
    static constexpr int lanes = 1;
    static constexpr int ring = 1 << 30;
    static constexpr int block = 1 << 28;
    static constexpr bool plain = true;
};

// after the symbol loop of a generic channel: what to carry (clock_recovery_mm.c:127-135), bounded by the channel's own cap
SDRM_HD void sdrm_k3_finish_linear(const sdrm_k3_lane &L, uint32_t hcap, int *from, int *keep) {
    const int64_t len = (int64_t) L.nz;  // L.kept is 0: positions are absolute in the working buffer
    int64_t f;
    if (len < SDRM_MMSE_TAPS) {
        f = 0;
    } else {
        const bool past = (L.st.ii < 0) || ((int64_t) L.st.ii > len);
        const int prev = (int) ((uint32_t) L.st.ii - (uint32_t) L.st.inc);
        f = past ? (int64_t) prev : (int64_t) L.st.ii;
    }
    int64_t k = len - f;
    if (k > (int64_t) hcap) {
        k = (int64_t) hcap;
        f = len - k;
    }
    if (k < 0) {
        k = 0;
        f = len;
    }
    *from = (int) f;
    *keep = (int) k;
}

// ------------------------------------------------------------------------------------------------ wild channels
// The ring-based symbol loops rest on one property of the timing loop: every symbol advances the position by at least one
// sample.  It holds while |mm| stays small: mu' = mu + omega + gain_mu * mm >= (omega_mid - omega_lim) - |gain_mu| * |mm|,
// |mm| <= |o| + |last| (clock_recovery_mm.c:115), |o| <= SDRM_MMSE_ABS_SUM * max |sample|.  sdrm_amp_safe() is the sample
// amplitude up to which that gives mu' >= 1 with a 1e-4 relative margin over the few-ulp rounding of the operations involved;
// the stage in front of the clock stage compares every sample it writes with it (the compare it already made for NaN/Inf)
// and raises SDRM_FLAG_WILD otherwise.  A wild channel can stand still (symbols without consuming samples, until the output
// buffer is full) or walk backwards by thousands of samples (discriminator gain of ~1900 at a deviation of a few Hz: mm of
// several thousand) -- both defined behaviour in the reference as long as `ii` stays inside [0, working_len).
SDRM_HD float sdrm_amp_safe(float omega_mid, float omega_lim, float gain_mu) {
    const double room = ((double) omega_mid - (double) omega_lim) * (1.0 - 1e-6) - 1.0;  // what gain_mu * mm may take away
    const double per_amp = 2.0 * fabs((double) gain_mu) * (double) SDRM_MMSE_ABS_SUM;
    if (!(room > 0.0) || !(per_amp > 0.0)) {
        return 0.0f;  // never tame (or NaN parameters)
    }
    // ... and never by more than the ring-based stage can carry into the next call: a symbol that jumps past the end of the
    // call's samples leaves everything from its own position on (clock_recovery_mm.c:127-133), i.e. less than one advance
    const double reach = (double) (SDRM_CLOCK_HCAP - 2) - ((double) omega_mid + (double) omega_lim) * (1.0 + 1e-6) - 1.0;
    const double a = (room < reach ? room : reach) / per_amp * (1.0 - 1e-4);
    if (!(a > 0.0)) {
        return 0.0f;
    }
    return a > 3.0e38 ? 3.0e38f : (float) a;
}
// the same bound for the carried `last` symbol: what a tame window can produce, with room for the dot product's rounding
SDRM_HD float sdrm_symbol_safe(float amp_safe) { return amp_safe * (SDRM_MMSE_ABS_SUM * 1.00001f); }

// One call of one WILD channel, from global memory: the reference's loop statement by statement (src/dsp/clock_recovery_mm.c:
// 78-139, the NaN-aware symbol) over the working buffer `carried samples ++ this call's samples`, any advance in either
// direction.  `cs` holds the state the call started from (the ring-based loop has not touched it) and receives the state it
// ends with; float and int8 soft bits and the count are written here.  Slow (a memory round trip per symbol) and rare.
// Returns the symbol count.  `flagged`: the call's SDRM_FLAG_* bits.
template <typename BankPtr>
SDRM_HD uint32_t sdrm_k3_rescue(const sdrm_chan_params &p, sdrm_clock_state *cs, const float *src, int nz, BankPtr bank_rev,
                                float *out_f32, int8_t *out_i8, uint32_t flagged) {
    const int kept = (int) cs->kept;
    sdrm_k3_lane L;
    L.k.omega_mid = p.omega_mid;
    L.k.omega_lim = p.omega_lim;
    L.k.gain_omega = p.gain_omega;
    L.k.gain_mu = p.gain_mu;
    L.kept = 0;  // positions are absolute in the working buffer, as the reference's ii
    L.nz = kept + nz;
    L.oo = 0;
    L.cap = p.max_len;
    L.st.mu = cs->mu;
    L.st.omega = cs->omega;
    L.st.last = cs->last;
    L.st.ii = 0;
    L.st.inc = 0;
    const float *hist = cs->hist;
    // sample q of the working buffer; in front of it: what an aligned dot product reads there, multiplied by zero taps
    auto at = [&](int q) -> float { return q < 0 ? 0.0f : (q < kept ? hist[q] : src[q - kept]); };
    const uint32_t lim = sdrm_k3_limit(L, L.nz);
    while (sdrm_k3_can_step(L, lim)) {
        sdrm_k3_operands F;
        const int ii = L.st.ii;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            F.w[j] = at(ii + j);
        }
        F.lead[0] = at(ii - 3);
        F.lead[1] = at(ii - 2);
        F.lead[2] = at(ii - 1);
        const float scaled = L.st.mu * (float) SDRM_MMSE_STEPS;
        F.row_ok = (scaled >= 0.0f) & (scaled <= (float) SDRM_MMSE_STEPS);  // false for NaN
        const BankPtr row = bank_rev + (F.row_ok ? (int) rintf(scaled) : 0) * SDRM_K3_BANKPITCH;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            F.tap[j] = row[j];
        }
        const float soft = sdrm_k3_step<false>(L, F);
        out_f32[L.oo] = soft;
        out_i8[L.oo] = sdrm_soft_to_i8(soft);
        L.oo++;
    }
    int from, keep;
    sdrm_k3_finish_linear(L, (uint32_t) (SDRM_CLOCK_HCAP - 1), &from, &keep);
    // carried samples, in place: entry j comes from position from + j >= j, so an ascending copy never reads what it wrote
    bool tame_carry = true;
    const float tame = p.amp_safe > 0.0f ? p.amp_safe : 0.0f;
    for (int j = 0; j < keep; j++) {
        const float v = at(from + j);
        cs->hist[j] = v;
        tame_carry &= fabsf(v) < tame;
    }
    cs->kept = (uint32_t) keep;
    cs->mu = L.st.mu;
    cs->omega = L.st.omega;
    cs->last = L.st.last;
    tame_carry &= fabsf(L.st.last) < sdrm_symbol_safe(tame);
    const bool finite_state = fabsf(L.st.mu) < INFINITY && fabsf(L.st.omega) < INFINITY && fabsf(L.st.last) < INFINITY;
    cs->poison = (((flagged & SDRM_FLAG_NONFINITE) != 0 || !finite_state) ? SDRM_FLAG_NONFINITE : 0u) | (tame_carry ? 0u : SDRM_FLAG_WILD);
    return L.oo;
}

#endif  // SDRM_KERNELS_H
