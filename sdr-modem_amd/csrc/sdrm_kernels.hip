// sdrm_kernels.hip -- gfx950 kernels of the GMSK/FSK demodulation pipeline (exact mode).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile): no FMA contraction, IEEE
// division, fp32 denormals preserved -- required for bit parity with the reference's CPU path.
#include <atomic>
#include <type_traits>

#include "sdrm_launch.h"

namespace sdrm {

// diagnostics: when a timeline buffer is attached every workgroup records when it started and ended (first start / last
// end per kernel and call survive), in ticks of the 100 MHz reference clock
__device__ __forceinline__ void tl_mark(const DeviceBatch &b, int kernel, int end) {
    if (b.timeline != nullptr && threadIdx.x == 0) {
        unsigned long long *slot = b.timeline + ((size_t) b.tl_row * 3 + kernel) * 2 + end;
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        if (end) {
            atomicMax(slot, t);
        } else {
            atomicMin(slot, t);
        }
    }
}

// ================================================================================================ K0 (NCO, row f-1)

// Phase accumulator (reference src/dsp/sig_source.c:43-58): the fp32 recursion phase += step with its wrap is sequential
// per channel, three issue slots per sample (k0_advance4) plus the hand-over, so a chunk costs its length times ~20
// cycles whatever the channel count.  One workgroup serves K0_CH = 64 channels with two waves: the generator wave runs the recursion, lane per
// channel, and drops 128-sample blocks into an LDS ring (rows of 128 + 4 floats: conflict-free b128 writes); the store
// wave reads them back time-major and writes 512-byte runs per channel (a lane-per-channel store touches one cache line
// per lane).  One barrier per block hands a ring half over.  (Unlike the clock stage, this chain gains nothing from a
// narrower wave: with 16 lanes a step costs 22.5 cycles instead of 20.2 and the hand-over 7.8 instead of 7.2 per
// sample, tools/ubench_nco.hip; 16 channels x 256-sample blocks measured 1.91 ms per chunk, 64 x 64 1.80.)
#define K0_CH 64
#define K0_BLK 128
#define K0_ROW (K0_BLK + 4)
#define K0_ROWS_PER_STORE (256 / K0_BLK)            // channel rows one 64-lane x 16-byte access covers
#define K0_STORES (K0_CH / K0_ROWS_PER_STORE)       // accesses per block
// hand a ring half over: the wave's own LDS traffic has landed (the store wave's global stores stay in flight)
#define K0_HANDOVER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// true when every lane can run a whole block branch-free: no batch boundary inside it and |step|, |phase| <= 2 pi
__device__ __forceinline__ bool k0_block_is_plain(uint32_t left, float step, float phase) {
    const float two_pi = 6.28318530717958647692f;
    return __all(left >= K0_BLK && fabsf(step) <= two_pi && fabsf(phase) <= two_pi);
}

// Four steps of sdrm_nco_advance_nomask (sdrm_core.h: why it returns the bits of the reference's two-test wrap): three
// dependent VALU instructions per sample and no compare -- a lone wave issues one instruction every four cycles, so the
// chain costs 12.1 cycles per sample where add / compare / subtract / wait / select cost 20.2 (a compare's mask may be
// read two issue slots after it is written; tools/ubench_nco.hip, modes 1 and 13).
// v = the phases of the four samples, phase = the phase after them.  bigs = +-2^24 and negw = -+2 pi by the step's sign.
#define K0_STEP(in, out) \
    "v_add_f32 " out ", " in ", %5\n\t"             \
    "v_fma_f32 %4, " out ", %6, %7 clamp\n\t"       \
    "v_fma_f32 " out ", %4, %8, " out "\n\t"
__device__ __forceinline__ void k0_advance4(float4 &v, float &phase, float step, float bigs, float negw) {
    float y, z, u, next, t;
    asm volatile(K0_STEP("%9", "%0") K0_STEP("%0", "%1") K0_STEP("%1", "%2") K0_STEP("%2", "%3")
                 : "=&v"(y), "=&v"(z), "=&v"(u), "=&v"(next), "=&v"(t)
                 : "v"(step), "v"(bigs), "s"(SDRM_NCO_WRAP_C), "v"(negw), "v"(phase));
    v.x = phase;
    v.y = y;
    v.z = z;
    v.w = u;
    phase = next;
}

__global__ __launch_bounds__(128) void k0_nco_phase(DeviceBatch b) {
    __shared__ __attribute__((aligned(16))) float ring[2][K0_CH * K0_ROW];
    __shared__ uint32_t blocks_of[K0_CH];
    __shared__ uint32_t blocks_max;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c0 = blockIdx.x * K0_CH;
    if (threadIdx.x == 0) {
        blocks_max = 0;
    }
    __syncthreads();
    if (wave == 0 && lane < K0_CH) {
        const int c = c0 + lane;
        uint32_t nb = 0;
        if (c < b.n_channels && b.ctl[c].nco_cnt != 0) {
            nb = (b.ctl[c].n_in + K0_BLK - 1) / K0_BLK;
        }
        blocks_of[lane] = nb;
        atomicMax(&blocks_max, nb);
    }
    __syncthreads();
    const uint32_t n_blocks = blocks_max;
    if (n_blocks == 0) {
        return;
    }
    if (wave == 0) {
        // ---- generator: a single dependent chain; issue ahead of whatever else shares the SIMD
        __builtin_amdgcn_s_setprio(3);
        if (lane < K0_CH) {  // the other lanes stay out of the exec mask for good (the barriers below are per wave)
            const int c = c0 + lane;
            const bool mine = blocks_of[lane] != 0;
            sdrm_chunk_ctl ctl;
            ctl.nco_cnt = 0;
            ctl.nco_off = 0;
            if (mine) {
                ctl = b.ctl[c];
            }
            const sdrm_nco_seg *seg = b.nco_segs + ctl.nco_off;
            float phase = mine ? b.nco_phase_state[c] : 0.0f;
            float last = phase;      // the state to keep: the phase after the channel's last sample
            bool open = mine;        // still inside its batches
            uint32_t k = 0, left = 0;
            float step = 0.0f;
            float *row = &ring[0][0] + lane * K0_ROW;
            for (uint32_t blk = 0; blk < n_blocks; blk++) {
                float *dst = row + (blk & 1) * (K0_CH * K0_ROW);
                if (left == 0 && open) {
                    // next batch (empty ones are skipped); past the last one the lane idles on step 0 until the block loop ends
                    while (k < ctl.nco_cnt && seg[k].len == 0) {
                        k++;
                    }
                    if (k < ctl.nco_cnt) {
                        step = seg[k].step;
                        left = seg[k].len;
                        k++;
                    } else {
                        open = false;
                        last = phase;
                        step = 0.0f;
                    }
                }
                if (!open) {
                    left = 0xffffffffu;
                }
                if (k0_block_is_plain(left, step, phase)) {
                    const float bigs = sdrm_nco_wrap_bigs(step), negw = sdrm_nco_wrap_negw(step);
                    for (int g0 = 0; g0 < K0_BLK / 4; g0 += 16) {
#pragma unroll
                        for (int g = 0; g < 16; g += 2) {
                            // two writes back to back cost less than two apart (ubench_nco.hip, modes 14 and 16)
                            float4 v0, v1;
                            k0_advance4(v0, phase, step, bigs, negw);
                            k0_advance4(v1, phase, step, bigs, negw);
                            *reinterpret_cast<float4 *>(dst + 4 * (g0 + g)) = v0;
                            *reinterpret_cast<float4 *>(dst + 4 * (g0 + g + 1)) = v1;
                        }
                    }
                    left -= K0_BLK;
                } else {
                    // a batch ends inside this block for some lane (about once a second per channel), or a step beyond
                    // one turn: the reference's two-test wrap, batch bookkeeping per sample
                    for (int s = 0; s < K0_BLK; s++) {
                        while (left == 0 && open) {
                            if (k < ctl.nco_cnt) {
                                step = seg[k].step;
                                left = seg[k].len;
                                k++;
                            } else {
                                open = false;
                                last = phase;
                                step = 0.0f;
                                left = 0xffffffffu;
                            }
                        }
                        dst[s] = phase;
                        phase = sdrm_nco_advance(phase, step);
                        left--;
                    }
                }
                K0_HANDOVER();
            }
            if (mine) {
                b.nco_phase_state[c] = open ? phase : last;
            }
        }
    } else {
        // ---- store wave: K0_BLK / 4 lanes per channel row, K0_ROWS_PER_STORE rows per instruction, 16 instructions per pass
        __builtin_amdgcn_s_setprio(2);
        const int piece = lane % (K0_BLK / 4), sub = lane / (K0_BLK / 4);
        uint32_t nb[K0_STORES], n_common = 0xffffffffu;
#pragma unroll
        for (int q = 0; q < K0_STORES; q++) {
            nb[q] = blocks_of[q * K0_ROWS_PER_STORE + sub];
            n_common = nb[q] < n_common ? nb[q] : n_common;
        }
        // blocks every channel of the workgroup still has: stores without a test
        uint32_t all_common = n_blocks;
        for (int l = 0; l < 64; l += K0_BLK / 4) {
            const uint32_t v = (uint32_t) __builtin_amdgcn_readlane((int) n_common, l);
            all_common = v < all_common ? v : all_common;
        }
        float *base = b.nco_phase + (size_t) (c0 + sub) * b.nco_phase_stride + piece * 4;
        const float *src = &ring[0][0] + sub * K0_ROW + piece * 4;
        for (uint32_t blk = 0; blk < n_blocks; blk++) {
            K0_HANDOVER();
            const float *from = src + (blk & 1) * (K0_CH * K0_ROW);
            float *col = base + (size_t) blk * K0_BLK;
#pragma unroll
            for (int q0 = 0; q0 < K0_STORES; q0 += 16) {
                float4 v[16];
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    v[q] = *reinterpret_cast<const float4 *>(from + (q0 + q) * K0_ROWS_PER_STORE * K0_ROW);
                }
                // all sixteen reads in flight before the first store
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
                }
                if (blk < all_common) {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        *reinterpret_cast<float4 *>(col + (size_t) ((q0 + q) * K0_ROWS_PER_STORE) * b.nco_phase_stride) = v[q];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        if (blk < nb[q0 + q]) {
                            *reinterpret_cast<float4 *>(col + (size_t) ((q0 + q) * K0_ROWS_PER_STORE) * b.nco_phase_stride) = v[q];
                        }
                    }
                }
            }
        }
    }
}

// mix: out[n] = in[n] * (cos, sin)(phase[n]), cos/sin in double on the fp32 phase (sig_source.c:46, :71)
__global__ __launch_bounds__(256) void k0_nco_mix(DeviceBatch b, const sdrm_f2 *__restrict__ d_in, size_t in_stride) {
    const int c = blockIdx.y;
    const sdrm_chunk_ctl ctl = b.ctl[c];
    if (ctl.nco_cnt == 0) {
        return;
    }
    // (a second oscillator in series -- the Doppler correction behind the file source's offset, src/dsp_worker.c:65-71 behind
    // src/sdr/file_source.c:120-128 -- mixes in place what the first one left: every sample is rounded to fp32 in between)
    const sdrm_f2 *in = ctl.pre ? b.nco_out + (size_t) c * b.nco_stride : d_in + (size_t) c * in_stride;
    const float *ph = b.nco_phase + (size_t) c * b.nco_phase_stride;
    sdrm_f2 *out = b.nco_out + (size_t) c * b.nco_stride;
    for (uint32_t n = blockIdx.x * blockDim.x + threadIdx.x; n < ctl.n_in; n += gridDim.x * blockDim.x) {
        out[n] = sdrm_nco_mix(in[n], sdrm_nco_sample(ph[n]));
    }
}

void launch_nco_phase(const DeviceBatch &b, hipStream_t s) {
    if (b.nco_segs == nullptr) {
        return;
    }
    hipLaunchKernelGGL(k0_nco_phase, dim3((unsigned) ((b.n_channels + K0_CH - 1) / K0_CH)), dim3(128), 0, s, b);
}

void launch_nco_mix(const DeviceBatch &b, const sdrm_f2 *d_in, size_t in_stride, uint32_t max_len, hipStream_t s) {
    if (b.nco_segs == nullptr || max_len == 0) {
        return;
    }
    unsigned gx = (max_len + 1023) / 1024;
    hipLaunchKernelGGL(k0_nco_mix, dim3(gx ? gx : 1, (unsigned) b.n_channels), dim3(256), 0, s, b, d_in, in_stride);
}

// ================================================================================================ K1

size_t k1_lds_bytes(uint32_t t1_max, uint32_t t2_max) { return sdrm_k1_lds_bytes_for(t1_max, t2_max); }

// grid (max_tiles, channels), 256 threads.  LDS: raw IQ tile + (T1-1) halo | quadrature-demod samples |
// per-thread boundary samples | arctan table.
template <bool HAND>
__global__ __launch_bounds__(SDRM_K1_THREADS, SDRM_K1_WGS) void k1_front(DeviceBatch b, const sdrm_f2 *__restrict__ d_in,
                                                            size_t in_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char k1_lds[];
    // grid (tiles, channels); with the in-call hand-off (channels, tiles): workgroups are dispatched x first, so every channel's
    // tile 0 comes before anybody's tile 1 and the stages behind can start on all channels at once.  (Round 6 tried (channels,
    // tiles) for the ordinary build too -- the workgroups dispatched last are then every channel's short last tile --: alone the
    // kernel ties, 0.434 vs 0.430 ms, in the pipeline it loses 6 %, 0.494 vs 0.465 ms on one box, profiles/r06_ab.txt: consecutive
    // workgroups then read 256 different channels' rows, a megabyte apart, beside the other stages' streams.)
    const int c = HAND ? blockIdx.x : blockIdx.y;
    const unsigned tile_id = HAND ? blockIdx.y + blockIdx.z * gridDim.y : blockIdx.x;  // (z: hand-off calls of more than 65535 tiles)
    const sdrm_chunk_ctl ctl = b.ctl[c];
    const sdrm_chan_params p = b.params[c];
    if (tile_id == 0) {
        // The channel's first workgroup also rolls the raw history for the next call (the last T1 + T2 - 1 samples of
        // history ++ input, into the other of the two history buffers: the tiles of THIS call read the current one).
        // Every channel gets this workgroup, also one without a tile this call (an empty or absent input still rolls).
        const sdrm_f2 *cur = b.raw_hist + ((size_t) c * 2 + ctl.parity) * b.hist_stride;
        sdrm_f2 *next = b.raw_hist + ((size_t) c * 2 + (ctl.parity ^ 1u)) * b.hist_stride;
        const sdrm_f2 *src = (ctl.nco_cnt | ctl.pre) ? b.nco_out + (size_t) c * b.nco_stride : d_in + (size_t) c * in_stride;
        sdrm_hist_roll((int) threadIdx.x, SDRM_K1_THREADS, p, ctl, src, cur, next);
    }
    if (tile_id >= ctl.tiles) {
        return;
    }
    tl_mark(b, 0, 0);
#ifndef SDRM_K1_NOPRIO
    __builtin_amdgcn_s_setprio(1);  // ahead of the clock stage's companion waves (priority 0) wherever they share a SIMD
#endif
    sdrm_f2 *xs = reinterpret_cast<sdrm_f2 *>(k1_lds);
    float *qs = reinterpret_cast<float *>(xs) + 1;  // aliases the raw tile: written only after every LPF1 read (barrier); qs[-1] exists
    float *zs = qs + SDRM_K1_NY + SDRM_K1_QPAD - 1;  // LPF2 outputs of the tile, behind the demodulated samples
    sdrm_f2 *bnd = reinterpret_cast<sdrm_f2 *>(k1_lds + SDRM_K1_XS_BYTES(b.t1_max));
    float *tab = reinterpret_cast<float *>(bnd + SDRM_K1_THREADS);

    const int tid = threadIdx.x;
    const sdrm_k1_tile t = sdrm_k1_tile_setup(p, ctl, (int) tile_id);
    const sdrm_f2 *in = (ctl.nco_cnt | ctl.pre) ? b.nco_out + (size_t) c * b.nco_stride : d_in + (size_t) c * in_stride;
    const sdrm_f2 *hist = b.raw_hist + ((size_t) c * 2 + ctl.parity) * b.hist_stride;

    const bool stamp = b.k3_stamps != nullptr;  // diagnostics: per-phase cycles, summed over workgroups
    unsigned long long t0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    // The taps go to LDS as well.  Their loads are issued BEFORE the tile's (filters of up to 512 taps: up to four values per
    // thread held in registers), so that one wait covers both; filters beyond 512 taps are staged afterwards.
    float *tab2 = tab + 260;  // {tab[i], tab[i+1] - tab[i]}, i < 256 (sdrm_quad_block_fast)
    float *taps1 = tab2 + 512, *taps2 = taps1 + ((p.T1 + 3) & ~3u);  // 16-byte aligned (tab starts aligned, 260 % 4 == 0)
    const bool short_taps = p.T1 <= 2 * SDRM_K1_THREADS && p.T2 <= 2 * SDRM_K1_THREADS;
    float tv0 = 0.0f, tv1 = 0.0f, tv2 = 0.0f, tv3 = 0.0f;
    if (short_taps) {
        if ((uint32_t) tid < p.T1) tv0 = b.tap_pool[p.taps1_off + tid];
        if ((uint32_t) tid + SDRM_K1_THREADS < p.T1) tv1 = b.tap_pool[p.taps1_off + tid + SDRM_K1_THREADS];
        if ((uint32_t) tid < p.T2) tv2 = b.tap_pool[p.taps2_off + tid];
        if ((uint32_t) tid + SDRM_K1_THREADS < p.T2) tv3 = b.tap_pool[p.taps2_off + tid + SDRM_K1_THREADS];
    }
    sdrm_k1_phase_load(tid, t, in, hist, (int) p.hist_len, b.atan_tab, xs, tab);
    for (int k = tid; k < 256; k += SDRM_K1_THREADS) {
        const float t0 = b.atan_tab[k], t1 = b.atan_tab[k + 1];  // 257 entries
        tab2[2 * k] = t0;
        tab2[2 * k + 1] = t1 - t0;  // the reference's subtraction (fast_atan2f.c:118), once per workgroup
    }
    if (short_taps) {
        if ((uint32_t) tid < p.T1) taps1[tid] = tv0;
        if ((uint32_t) tid + SDRM_K1_THREADS < p.T1) taps1[tid + SDRM_K1_THREADS] = tv1;
        if ((uint32_t) tid < p.T2) taps2[tid] = tv2;
        if ((uint32_t) tid + SDRM_K1_THREADS < p.T2) taps2[tid + SDRM_K1_THREADS] = tv3;
    } else {
        // longer filters (more than 512 taps: 200 samples per symbol at 240 kHz): every thread's tap loads are issued before
        // the first is stored, so that they are in flight together like the tile's
        constexpr int TAPS_DEPTH = 6;  // 1536 taps per pass
        for (uint32_t k0 = 0; k0 < p.T1; k0 += TAPS_DEPTH * SDRM_K1_THREADS) {
            float v[TAPS_DEPTH];
#pragma unroll
            for (int i = 0; i < TAPS_DEPTH; i++) {
                const uint32_t k = k0 + tid + i * SDRM_K1_THREADS;
                v[i] = k < p.T1 ? b.tap_pool[p.taps1_off + k] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < TAPS_DEPTH; i++) {
                const uint32_t k = k0 + tid + i * SDRM_K1_THREADS;
                if (k < p.T1) taps1[k] = v[i];
            }
        }
        for (uint32_t k0 = 0; k0 < p.T2; k0 += TAPS_DEPTH * SDRM_K1_THREADS) {
            float v[TAPS_DEPTH];
#pragma unroll
            for (int i = 0; i < TAPS_DEPTH; i++) {
                const uint32_t k = k0 + tid + i * SDRM_K1_THREADS;
                v[i] = k < p.T2 ? b.tap_pool[p.taps2_off + k] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < TAPS_DEPTH; i++) {
                const uint32_t k = k0 + tid + i * SDRM_K1_THREADS;
                if (k < p.T2) taps2[k] = v[i];
            }
        }
    }
    __syncthreads();
    unsigned long long t1 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    sdrm_k1_regs regs;
    sdrm_k1_phase_lpf1(tid, t, p, taps1, xs, bnd, regs);
    __syncthreads();
    unsigned long long t2 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    sdrm_k1_phase_quad(tid, t, p, tab, b.quad_general ? nullptr : tab2, bnd, regs, qs);
    __syncthreads();
    unsigned long long t3 = stamp ? __builtin_amdgcn_s_memtime() : 0;
    sdrm_k1_phase_lpf2(tid, t, p, taps2, qs, zs, b.nonfinite + c);
    __syncthreads();
    if (HAND) {
        // The tile is in memory (and any flag it raised, sdrm_k1_phase_lpf2) before its stamp says so: written THROUGH this
        // XCD's L2 with device-scope stores, every wave waits for the acknowledgement of its own, then one thread stamps.
        // (Not a release fence: that is a write-back of the whole L2, and 8960 of them in half a millisecond stalled every
        // other client of the cache -- the clock stage's soft-bit stores waited ~1800 cycles per staging step for theirs.)
        sdrm_k1_phase_store<true>(tid, t, zs, b.z + (size_t) c * b.z_stride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(b.hand_tiles + (size_t) c * b.hand_tiles_cap + tile_id, b.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        sdrm_k1_phase_store(tid, t, zs, b.z + (size_t) c * b.z_stride);
    }
    tl_mark(b, 0, 1);
    if (stamp && tid == 0) {
        unsigned long long t4 = __builtin_amdgcn_s_memtime();
        unsigned long long *k1s = b.k3_stamps + SDRM_STAMP_K3_WAVES(b.n_channels) * 4;  // after the K3 per-wave records
        atomicAdd(k1s + 0, t1 - t0);
        atomicAdd(k1s + 1, t2 - t1);
        atomicAdd(k1s + 2, t3 - t2);
        atomicAdd(k1s + 3, t4 - t3);
        atomicAdd(k1s + 4, 1ull);
    }
}

// dynamic LDS above the 64 KiB default has to be requested per kernel (up to the CU's 160 KiB); the attribute belongs
// to the kernel's code object on ONE device, so what has been granted is remembered per device (a process that drives
// several GPUs gets it right for each)
struct lds_grant {
    std::atomic<size_t> bytes[16] = {};  // handles are created and used from many threads
};
template <typename K>
static void allow_lds(K kernel, size_t bytes, lds_grant *granted) {
    int dev = 0;
    (void) hipGetDevice(&dev);
    std::atomic<size_t> &have = granted->bytes[dev & 15];
    if (bytes > 64 * 1024 && bytes > have.load(std::memory_order_acquire)) {
        // setting the attribute twice (two threads at once) is harmless; recording it before it is set would not be
        (void) hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
        size_t seen = have.load(std::memory_order_relaxed);
        while (seen < bytes && !have.compare_exchange_weak(seen, bytes, std::memory_order_release)) {
        }
    }
}

// Every stage's launch is described once (kernel, grid, block, dynamic LDS) and then either launched on a stream or put
// into an explicitly built graph (sdrm_call.hip: the one-channel blocking call).  func == nullptr: nothing to launch.
static void launch_described(const KernelLaunch &k, void **args, hipStream_t s) {
    if (k.func != nullptr) {
        (void) hipLaunchKernel(k.func, k.grid, k.block, args, k.lds, s);
    }
}

KernelLaunch describe_front(const DeviceBatch &b) {
    KernelLaunch k;
    static lds_grant granted, granted_hand;
    k.lds = k1_lds_bytes(b.t1_max, b.t2_max);
    if (b.handoff) {
        allow_lds(k1_front<true>, k.lds, &granted_hand);
        k.func = reinterpret_cast<const void *>(k1_front<true>);
    } else {
        allow_lds(k1_front<false>, k.lds, &granted);
        k.func = reinterpret_cast<const void *>(k1_front<false>);
    }
    const unsigned tiles = b.max_tiles ? b.max_tiles : 1u;  // tile 0 of every channel also rolls its history
    k.grid = dim3(tiles, (unsigned) b.n_channels);
    if (b.handoff) {
        k.grid = dim3((unsigned) b.n_channels, tiles < 65535u ? tiles : 65535u, (tiles + 65534u) / 65535u);
    }
    k.block = dim3(SDRM_K1_THREADS);
    return k;
}

void launch_front(const DeviceBatch &b, const sdrm_f2 *d_in, size_t in_stride, hipStream_t s) {
    void *args[] = {(void *) &b, (void *) &d_in, (void *) &in_stride};
    launch_described(describe_front(b), args, s);
}

// ================================================================================================ K2

// Which stage's workgroups get onto the compute units first matters, and the dispatcher reserves nothing:
//  * A clock-recovery workgroup needs most of a CU's LDS (141 KB in the full shapes), a DC workgroup of sixteen channels
//    ~117 KB: the two never share a CU.  The clock stage of call k and the DC stage of call k+1 are released by the same
//    event (the clock stage of call k-1 finishing); with many channels the DC grid would cover every CU before the clock
//    stage's workgroups are placed, which then wait for DC workgroups to finish.
//  * The front-end of call k+2 is released by that event too; once its thousands of small workgroups are streaming through
//    the chip no CU ever has 141 KB free and the clock stage starts only when that grid has drained (seen: every other call
//    0.5-1.2 ms late at 512 channels).
//  * With thousands of channels a DC workgroup cannot start on a CU that holds more than one front-end workgroup, and the
//    front-end's grid of the NEXT call refills every CU as fast as it drains: the DC stage then starts only when that grid
//    is nearly through (9.4 ms for 1.5 ms of work at 4096 channels).
// So a stage that would swamp the chip waits, in front of its kernel, until the stage that must go first HAS ITS WORKGROUPS
// PLACED: every DC / clock-stage workgroup bumps a counter when it starts (b.placed[0] / [1], cumulative over calls), and a
// one-wave kernel on the waiting stream spins on that counter (s_sleep between looks) up to a bound.  Rounds 1-2 used
// fixed sleeps (~50 / 120 / 240 us) tuned on 131072-sample calls; with 4096-sample calls those cost up to 16 %
// (profiles/r03_heuristics_before.txt), and the counter takes what the placement really needs (tens of microseconds).
__global__ void k_hold_until(const uint32_t *counter, uint32_t target, int max_looks) {
    for (int i = 0; i < max_looks; i++) {
        if ((int32_t) (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) {
            break;
        }
        __builtin_amdgcn_s_sleep(32);  // ~1 us between two looks
    }
}
void launch_hold_until(const uint32_t *counter, uint32_t target, int max_us, hipStream_t s) {
    hipLaunchKernelGGL(k_hold_until, dim3(1), dim3(64), 0, s, counter, target, max_us);
}
// The front-end waits for the clock stage's placement between these channel counts (below, the front-end is short enough
// to be through before the next clock stage looks for its CUs; above, the front-end is what the step waits for).  The lower
// bound was 128 until the end of round 3: with the placement counters and the int8 conversion inside the clock stage the
// wait costs 2-3 % from 128 to 768 channels and still gains 2-5 % from 896 to 1024 (131072- and 32768-sample calls,
// profiles/r03_front_hold_bounds.txt).  SDRM_FRONT_HOLD="lo,hi" overrides.
bool front_hold_is_forced() { return getenv("SDRM_FRONT_HOLD") != nullptr; }
bool k3_shape_is_forced() { return getenv("SDRM_K3_LANES") != nullptr; }
bool front_waits_for_clock_start(int n_channels) {
    static const char *e = getenv("SDRM_FRONT_HOLD");
    int lo = 832, hi = 1024;
    if (e != nullptr) {
        sscanf(e, "%d,%d", &lo, &hi);
    }
    return n_channels >= lo && n_channels <= hi;
}
// (Rounds 1-3 also had the front-end wait for the DC stage's placement and the DC stage for the clock stage's -- SDRM_DC_FIRST,
// SDRM_DC_HOLD.  Measured useless since the clock stage of large batches takes the plain ring, profiles/r03_dcfirst.txt and
// r03_dchold.txt; removed in round 5.)

// DC blocker: design in sdrm_kernels.h (K2).  Workgroup = dc_group channels (16 unless long boxcars need the LDS), six
// waves: 0 chain, 1 feeder, 2..4 stage s -> s+1, 5 output.  Iteration `it` (one barrier each): the chain wave sums block
// it - 2s of stage s; the feeder prepares block it + 1 of stage 0; the helper of stage s converts block it - 1 - 2s.
// The helpers below compute what sdrm_k2_feed / _transition / _output (sdrm_kernels.h, also run by the CPU emulation)
// state, with 16-byte LDS accesses, the global loads of the next block in flight while this one is worked on, and the
// ring position kept incrementally instead of by division.
typedef float k2_f4 __attribute__((ext_vector_type(4)));
// One barrier per iteration hands LDS blocks from role to role: the wave's own LDS traffic must have landed, its global
// loads (the NEXT block's operands, into registers) and stores stay in flight -- __syncthreads() would wait for those too
// and expose a full memory latency per iteration.
#define K2_HANDOVER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// ---- in-call hand-off: what a stage behind the front-end looks at before it reads samples of THIS call
// (every look is bounded: a producer that never comes -- a failed launch -- must not hang the device; the call then fails
// loudly, DeviceBatch::counters[1] and SDRM_OUT_LEN_FAILED)
// has the front-end tile that holds output `last` (>= 0) of channel c been written?  (relaxed; the caller fences)
__device__ __forceinline__ bool hand_tile_done(const DeviceBatch &b, int c, uint32_t tile_m, int last) {
    const uint32_t tile = (uint32_t) last / tile_m;
    return __hip_atomic_load(b.hand_tiles + (size_t) c * b.hand_tiles_cap + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == b.epoch;
}
// DC-blocker outputs of this call published for channel c (0 while the stamp is an older call's)
__device__ __forceinline__ uint32_t hand_prog_of(const DeviceBatch &b, int c) {
    const unsigned long long v = __hip_atomic_load(b.hand_prog + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t) (v >> 32) == b.epoch ? (uint32_t) v : 0u;
}
// Device-scope (sc1) vector accesses through a raw buffer resource over the workgroup's rows of the array: written THROUGH / read PAST this XCD's
// L2, so that a producer and a consumer running at the same time on different XCDs meet in memory -- a store is there once it
// has been acknowledged (s_waitcnt vmcnt), a load issued after that sees it; no cache write-back or invalidation involved.
// (Buffer instructions because the cache policy of a 16-byte access cannot be said otherwise; the compiler tracks them like any load.)
typedef unsigned int hand_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int hand_u2 __attribute__((ext_vector_type(2)));
#define HAND_SC1 16  // cache policy bit: device scope
__device__ __forceinline__ __amdgpu_buffer_rsrc_t hand_rsrc(const float *base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ void hand_store4(__amdgpu_buffer_rsrc_t r, size_t index, float x, float y, float z, float w) {
    const hand_u4 v = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int) (uint32_t) (index * sizeof(float)), 0, HAND_SC1);
}
__device__ __forceinline__ void hand_load4(__amdgpu_buffer_rsrc_t r, size_t index, float *v) {
    const hand_u4 t = __builtin_amdgcn_raw_buffer_load_b128(r, (int) (uint32_t) (index * sizeof(float)), 0, HAND_SC1);
    v[0] = __uint_as_float(t.x);
    v[1] = __uint_as_float(t.y);
    v[2] = __uint_as_float(t.z);
    v[3] = __uint_as_float(t.w);
}
__device__ __forceinline__ void hand_load2(__amdgpu_buffer_rsrc_t r, size_t index, float *v) {
    const hand_u2 t = __builtin_amdgcn_raw_buffer_load_b64(r, (int) (uint32_t) (index * sizeof(float)), 0, HAND_SC1);
    v[0] = __uint_as_float(t.x);
    v[1] = __uint_as_float(t.y);
}

struct k2_lane {
    uint32_t tile_m;     // hand-off: front-end outputs per tile of this channel
    sdrm_k2_slot s;
    int q;               // which P-sample piece of a block this lane owns
    const float *z;      // the channel's front-end output of this call
    const float *hx;     // its carried samples of x
    float *out;          // its DC-free output
    bool on;
};
#define K2_P SDRM_K2_P
#define K2_V4 (SDRM_K2_P / 4)  // 16-byte pieces per lane

// P consecutive floats, 16-byte aligned (LDS rows and ring slots of whole lanes, z rows at lane starts)
__device__ __forceinline__ void k2_load_p(const float *src, float (&v)[K2_P]) {
    const k2_f4 *p = reinterpret_cast<const k2_f4 *>(src);
#pragma unroll
    for (int g = 0; g < K2_V4; g++) {
        const k2_f4 t = p[g];
        v[4 * g] = t.x;
        v[4 * g + 1] = t.y;
        v[4 * g + 2] = t.z;
        v[4 * g + 3] = t.w;
    }
}
__device__ __forceinline__ void k2_store_p(float *dst, const float (&v)[K2_P]) {
    k2_f4 *d = reinterpret_cast<k2_f4 *>(dst);
#pragma unroll
    for (int g = 0; g < K2_V4; g++) {
        d[g] = k2_f4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    }
}
// P consecutive floats from an 8-byte aligned address (x[n - 2(L-1)]: the delay is even)
__device__ __forceinline__ void k2_load_p8(const float *src, float (&v)[K2_P]) {
    typedef float k2_f2 __attribute__((ext_vector_type(2)));
    const k2_f2 *p = reinterpret_cast<const k2_f2 *>(src);
#pragma unroll
    for (int g = 0; g < K2_P / 2; g++) {
        const k2_f2 t = p[g];
        v[2 * g] = t.x;
        v[2 * g + 1] = t.y;
    }
}
// the same from an address that is only 4-byte aligned (delayed samples: the delay is any number)
__device__ __forceinline__ void k2_load_p_any(const float *src, bool aligned, float (&v)[K2_P]) {
    if (aligned) {
        k2_load_p(src, v);
    } else {
#pragma unroll
        for (int i = 0; i < K2_P; i++) v[i] = src[i];
    }
}

// P consecutive samples x[m0 .. m0+P-1] of the channel's input stream (m0 may be negative: carried samples)
__device__ __forceinline__ void k2_load_x(const k2_lane &L, int m0, float (&v)[K2_P]) {
    if (m0 >= 0) {
        k2_load_p_any(L.z + m0, (m0 & 3) == 0, v);
    } else if (m0 + K2_P <= 0) {
        const float *p = L.hx + ((int) L.s.HX + m0);
#pragma unroll
        for (int i = 0; i < K2_P; i++) v[i] = p[i];
    } else {
#pragma unroll
        for (int i = 0; i < K2_P; i++) v[i] = sdrm_k2_x(L.z, L.hx, L.s.HX, m0 + i);
    }
}

// the P quotients of a lane: running sums rebuilt from the checkpoint (the chain wave's additions, in its order), / L.
// sdrm_boxcar_out_fast's `unsafe` test for all of them at once: a running sum that is not finite stays so (Inf + x, NaN + x),
// so the lane's LAST sum tells; a zero or denormal quotient shows in the smallest |sum| (|sum| < 2^-100 is far above
// what gives a denormal quotient for any length up to 4096, and costs nothing but the division it then takes).
__device__ __forceinline__ void k2_quotients(const k2_lane &L, const float (&t)[K2_P], float cp, float (&v)[K2_P]) {
    float sums[K2_P];
    float acc = cp;
#pragma unroll
    for (int i = 0; i < K2_P; i++) {
        acc = acc + t[i];
        sums[i] = acc;
        const float q0 = acc * L.s.invL;
        const float r = fmaf(-q0, L.s.Lf, acc);
        v[i] = fmaf(r, L.s.invL, q0);
    }
    float lo = fabsf(sums[0]);
#pragma unroll
    for (int i = 1; i < K2_P; i++) {
        lo = fminf(lo, fabsf(sums[i]));  // v_min3_f32 with |.| modifiers: two sums per instruction
    }
    const bool unsafe = !(fabsf(acc) < INFINITY) | (lo < 7.888609e-31f);
    if (__any(unsafe)) {
#pragma unroll
        for (int i = 0; i < K2_P; i++) v[i] = sdrm_boxcar_out(sums[i], L.s.Lf);
    }
}

// HAND: the in-call hand-off (the front-end is still running; DeviceBatch::handoff) -- a build of its own, so that the
// ordinary one carries none of its code
template <bool HAND>
__global__ __launch_bounds__(64 * SDRM_K2_WAVES) void k2_dc(DeviceBatch b) {
    extern __shared__ __attribute__((aligned(16))) float k2_lds[];
    float *ts = k2_lds;                                                  // [64 rows][TSPITCH]
    float *check = ts + SDRM_K2_ROWS * SDRM_K2_TSPITCH;                  // [64 rows][NBUF][LPS]
    float *xdt = check + SDRM_K2_ROWS * SDRM_K2_NBUF * SDRM_K2_LPS;      // [16 slots][2][64]: x[n - 2(L-1)] for the output role
    float *rings = xdt + SDRM_K2_SLOTS * 2 * SDRM_K2_BLK;                // [3][group][ring pitch]
    const uint32_t rpitch = b.dc_rpitch;
    sdrm_k2_slot *slots = reinterpret_cast<sdrm_k2_slot *>(rings + 3 * (size_t) b.dc_group * rpitch);
    __shared__ int nb_sh;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int c0 = blockIdx.x * b.dc_group;
    tl_mark(b, 1, 0);
    if (b.placed != nullptr && tid == 0) {
        atomicAdd(b.placed + 0, 1u);  // this workgroup has its CU (k_hold_until)
    }

    // ---- set-up: slot constants, zeroed term rows, delay rings from the channels' tails
    if (tid == 0) {
        nb_sh = 0;
    }
    for (int i = tid; i < SDRM_K2_ROWS * SDRM_K2_TSPITCH; i += blockDim.x) {
        ts[i] = 0.0f;
    }
    __syncthreads();
    if (tid < SDRM_K2_SLOTS) {
        sdrm_k2_slot s;
        s.chan = -1;
        s.L = 1; s.A = 64; s.rcap = 128; s.nz = 0; s.HX = 0; s.Lf = 1.0f; s.invL = 1.0f; s.alias = 0;
        const int c = c0 + tid;
        if (tid < b.dc_group && c < b.n_channels) {
            const sdrm_chan_params p = b.params[c];
            const sdrm_chunk_ctl ctl = b.ctl[c];
            if (p.dc_len != 0 && ctl.absent == 0 && p.generic == 0) {  // generic channels: k2_dc_generic
                sdrm_k2_slot_setup(s, c, p, ctl.nz);
                atomicMax(&nb_sh, (int) ((ctl.nz + SDRM_K2_BLK - 1) / SDRM_K2_BLK));
            }
        }
        slots[tid] = s;
    }
    __syncthreads();
    const int nb = nb_sh;
    if (nb == 0) {
        return;  // nothing to do for any channel of the group: states stay as they are
    }
    if (tid == 0) {
        sdrm_k2_fill_aliases(slots, (int) b.dc_group);  // empty slots beside live ones: replicas, so that every lane has a channel
    }
    __syncthreads();
    // roles: wave 0 = chain; then WPR waves each of feeder, stage 0, 1, 2, output; a helper wave covers P slots
    const int role = wave == 0 ? 0 : 1 + (wave - 1) / SDRM_K2_WPR;      // 0 chain, 1 feeder, 2..4 stage (role - 2), 5 output
    const int sub = wave == 0 ? 0 : (wave - 1) % SDRM_K2_WPR;
    k2_lane L;
    const int slot_h = sub * SDRM_K2_P + lane / SDRM_K2_LPS;
    L.s = slots[slot_h];
    L.q = lane % SDRM_K2_LPS;
    L.on = L.s.chan >= 0;
    L.tile_m = L.on ? b.params[L.s.chan].tile_m : 1u;
    L.hx = b.dc_state + (L.on ? b.params[L.s.chan].dc_state_off : 0);
    L.z = b.z + (size_t) (L.on ? L.s.chan : 0) * b.z_stride;
    L.out = b.dcout + (size_t) (L.on ? L.s.chan : 0) * b.z_stride;
    if (role >= 2 && role <= 4) {
        // the stage helper of a slot owns that slot's ring of its stage: the wave fills its P rings one at a time
        for (int i = 0; i < SDRM_K2_P; i++) {
            const int sl = sub * SDRM_K2_P + i;
            const sdrm_k2_slot s = slots[sl];
            if (sl < b.dc_group && s.chan >= 0) {
                float *st = b.dc_state + b.params[s.chan].dc_state_off;
                sdrm_k2_ring_load(rings + ((size_t) (role - 2) * b.dc_group + sl) * rpitch, s,
                                  sdrm_k2_state_tail(st, role - 2, b.dc_hx_cap, b.dc_l_cap), lane, 64);
            }
        }
    }
    const unsigned long long t_begin = (b.k3_stamps != nullptr && blockIdx.x == 0) ? __builtin_amdgcn_s_memtime() : 0;

    float acc = 0.0f;                // chain wave: running sum of row (stage, slot) = lane
    bool odd = false;
    float amax = 0.0f;               // output role: largest |output| of this lane (NaN never enters: v_max returns the other operand)
    uint32_t bslot = L.s.A;          // stage helpers: ring slot of the block they convert next (block 0: A)
    const int n_lane = L.q * K2_P;   // first sample of this lane inside a block
    const int nz = (int) L.s.nz;
    // delayed samples may be fetched before this block's quotients are stored when they cannot be among them
    const bool early = __all(!L.on || L.s.L >= (uint32_t) SDRM_K2_BLK);
    if (role == 0) {
        const sdrm_k2_slot s = slots[lane & (SDRM_K2_SLOTS - 1)];
        if (s.chan >= 0) {
            acc = sdrm_k2_state_acc(b.dc_state + b.params[s.chan].dc_state_off, b.dc_hx_cap, b.dc_l_cap)[lane >> 4];
        }
        // one dependent chain: issue ahead of whatever shares the SIMD (any level measured alike, profiles/r04_dc_prio.txt)
        __builtin_amdgcn_s_setprio(3);
    } else {
        __builtin_amdgcn_s_setprio(1);  // the helpers: ahead of the clock stage's companion waves (priority 0)
    }
    // In-call hand-off: the feeder (the only role that reads the front-end's output inside the loop) makes sure, before it
    // touches outputs [.., upto) of this call, that the tiles holding them are in memory for every slot of its wave.  One
    // acquire per newly finished tile row; z_ready remembers how far that reaches.
    int z_ready = HAND ? 0 : 0x7fffffff;  // wave-uniform: outputs [0, z_ready) of every slot of this wave are known to be there
    int my_ready = L.on ? 0 : 0x7fffffff;      // this lane's slot alone
    auto await_z = [&](int upto) {
        if (upto <= z_ready) {
            return;
        }
        const int need = upto < nz ? upto : nz;  // this lane's slot: nothing beyond its own call
        auto look = [&]() {  // walk over the tiles that are there, in order (a tile may be shorter than a block)
            while (my_ready < need && hand_tile_done(b, L.s.chan, L.tile_m, my_ready)) {
                my_ready = (int) (((uint32_t) my_ready / L.tile_m + 1u) * L.tile_m);
            }
            return my_ready >= need;
        };
        bool there = look();
        for (int looks = 0; !__all(there) && looks < SDRM_HAND_MAX_LOOKS; looks++) {
            __builtin_amdgcn_s_sleep(16);
            there = look();
        }
        if (!__all(there)) {
            atomicOr(b.counters + 1, 1u);  // gave up: what follows is computed from whatever is there, the call is void
            my_ready = 0x7fffffff;         // ... and nobody waits again (one bound per kernel, not one per block)
        }
        int reach = my_ready >= nz ? 0x7fffffff : my_ready;
        for (int sl = 0; sl < SDRM_K2_P; sl++) {
            const int r = __builtin_amdgcn_readlane(reach, sl * SDRM_K2_LPS);
            reach = r < reach ? r : reach;
        }
        z_ready = __builtin_amdgcn_readfirstlane(reach);
        if (z_ready < upto) {
            z_ready = upto;  // (a wait that ran into its bound: do not come back for every block)
        }
    };
    // this call's front-end output as the feeder reads it: plainly, or -- while the front-end is still running (hand-off) -- with
    // device scope, past this XCD's L2 (hand_load*)
    // (the resource starts at the row of the workgroup's first channel: the 32-bit byte offsets then span dc_group rows, not the
    // whole array -- 1024 channels x 1 M samples is 4 GiB; the host refuses the hand-off beyond 2^32 bytes per group)
    const __amdgpu_buffer_rsrc_t z_rsrc = hand_rsrc(b.z + (size_t) c0 * b.z_stride);
    const size_t z_row = (size_t) (L.on ? L.s.chan - c0 : 0) * b.z_stride;
    auto x4 = [&](int m, float *v) {    // x[m .. m+3]: any alignment, maybe carried samples
        if (m >= 0 && (m & 3) == 0) {
            if (HAND) {
                hand_load4(z_rsrc, z_row + (size_t) m, v);
            } else {
                const k2_f4 t = *reinterpret_cast<const k2_f4 *>(L.z + m);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            }
        } else {
            for (int e = 0; e < 4; e++) {
                v[e] = (HAND && m + e >= 0) ? __hip_atomic_load(L.z + (m + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                 : sdrm_k2_x(L.z, L.hx, L.s.HX, m + e);
            }
        }
    };
    if (role == 1) {
        await_z(SDRM_K2_BLK);
        if (L.on && !HAND) {
            sdrm_k2_feed(L.s, 0, L.q, L.z, L.hx, ts + slot_h * SDRM_K2_TSPITCH);  // block 0 of stage 0
        } else if (L.on) {
            // the same terms (sdrm_k2_feed: t = x[n] - x[n - L] for the P samples of this lane), the samples read with device scope
            for (int g = 0; g < K2_P / 4; g++) {
                const int n = L.q * K2_P + 4 * g;
                float va[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (n < nz) {
                    x4(n, va);
                    x4(n - (int) L.s.L, vb);
                }
                float *dst = ts + slot_h * SDRM_K2_TSPITCH + n;
                for (int e = 0; e < 4; e++) {
                    dst[e] = (n + e < nz) ? sdrm_boxcar_term(va[e], vb[e]) : 0.0f;
                }
            }
        }
    }
    __syncthreads();
    const int n_it = nb + 7;
    const bool stamp = b.k3_stamps != nullptr && blockIdx.x == 0;  // diagnostics: cycles each role works per iteration
    unsigned long long busy = 0;
#ifdef SDRM_K2_DEBUG_STAGE
    unsigned long long dbg[4] = {0, 0, 0, 0};
#endif
    // Interior iterations take straight-line code: every lane of the wave has a channel, its block lies wholly inside
    // the call, nothing comes from the carried samples, delayed quotients sit at 16-byte boundaries.  Everything else
    // (the first and last blocks, ragged batches, odd delays, partial groups) takes the predicated code below.
    int nz_min = nz, l_max = (int) L.s.L, hx_max = (int) L.s.HX;
    bool all_on = L.on;
    {
        const sdrm_k2_slot *sp = slots + (role == 0 ? 0 : sub * SDRM_K2_P);
        for (int i = 0; i < (role == 0 ? SDRM_K2_SLOTS : SDRM_K2_P); i++) {
            nz_min = min(nz_min, (int) sp[i].nz);
            l_max = max(l_max, (int) sp[i].L);
            hx_max = max(hx_max, (int) sp[i].HX);
            all_on = all_on && sp[i].chan >= 0;
        }
    }
    const int nb_full = all_on ? nz_min / SDRM_K2_BLK : 0;  // blocks 0 .. nb_full-1 are whole for every slot of the wave
    const bool l_mod4 = __all((L.s.L & 3u) == 0);           // x[n-L] and the delayed quotients at 16-byte boundaries
    const bool fast_wave = all_on && early && l_mod4;
    // Every role runs its own copy of the iteration loop (the same number of barriers in each): the roles then share no
    // control flow, and the compiler's bookkeeping of outstanding loads and stores of one role (the feeder's operands in
    // flight across the barrier, the output role's stores) cannot make another role wait for them.
    const int role_rt = role;
    // In-call hand-off, output role: block k of this lane's slot goes to memory with device-scope (write-through) stores, and
    // the slot's count of finished outputs follows ONE ITERATION LATER -- by then the stores (and any flag the block raised:
    // the clock stage must see it before the samples) have long been acknowledged, so the wait in front of the count costs
    // nothing and no cache write-back is needed.  Every lane of the wave comes here once per block, `valid` of its P outputs count.
    const float tame5 = (role == 5 && L.on) ? sdrm_tame_level(b.params[L.s.chan], true) : INFINITY;
    // `whole`: the straight-line iterations, where every lane stores its P outputs with exactly K2_V4 store instructions: the
    // count then lags K2_LAG blocks, so that only stores issued K2_LAG - 1 iterations ago are waited for (a write through to
    // memory takes longer than one iteration: waiting for the previous block's stalled the whole workgroup at its barrier,
    // 0.54 -> 1.02 us per block).  Stores of a wave are acknowledged in order.
#define K2_LAG 4
    int done_said = 0;
    const __amdgpu_buffer_rsrc_t out_rsrc = hand_rsrc(b.dcout + (size_t) c0 * b.z_stride);
    const size_t out_row = (size_t) (L.on ? L.s.chan - c0 : 0) * b.z_stride;
    auto publish = [&](int k, int n0, const float (&o)[K2_P], int valid, bool whole, bool suspicious) {
        int done;  // blocks 0 .. done - 1 of this wave's slots are in memory
        if (whole) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K2_V4 * (K2_LAG - 1)) : "memory");
            done = k - (K2_LAG - 1);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            done = k;
        }
        const bool further = done > done_said;  // (the straight-line iterations lag: never take a count back)
        done_said = further ? done : done_said;
        if (L.on && L.q == 0 && further) {
            const uint32_t have = (uint32_t) (done * SDRM_K2_BLK < nz ? done * SDRM_K2_BLK : nz);
            __hip_atomic_store(b.hand_prog + L.s.chan, ((unsigned long long) b.epoch << 32) | have, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (suspicious) {  // rare: NaN/Inf among this lane's outputs so far, or one beyond the tame amplitude -- say which, now
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < K2_P; i++) {
                bits |= i < valid ? sdrm_flag_bits(o[i], tame5) : 0u;
            }
            if (bits) {
                atomicOr(b.nonfinite + L.s.chan, bits);
            }
        }
        if (valid == K2_P) {
#pragma unroll
            for (int g = 0; g < K2_V4; g++) {
                hand_store4(out_rsrc, out_row + (size_t) (n0 + 4 * g), o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < K2_P; i++) {
                if (i < valid) {
                    __hip_atomic_store(L.out + n0 + i, o[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    };
    auto body = [&](auto role_c, auto fast_c, int it, unsigned long long w0) {
        constexpr int R = decltype(role_c)::value;  // 0 chain, 2 any of the three stages, 5 output (the feeder has its own loops)
        constexpr bool FAST = decltype(fast_c)::value;  // interior iteration: straight-line code
        const int role = R == 2 ? role_rt : R;
        if (R == 2) {
            __builtin_assume(role >= 2 && role <= 4);
        }
        (void) w0;
        if (FAST && role == 0) {
            // ---- chain, all four stages inside their calls: no lane stays out
            const int ks = it - 2 * (lane >> 4);
            const int buf = ks % SDRM_K2_NBUF;
            const k2_f4 *row = reinterpret_cast<const k2_f4 *>(ts + lane * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK);
            k2_f4 t[16];
#pragma unroll
            for (int g = 0; g < 16; g++) {
                t[g] = row[g];
            }
            float cp[SDRM_K2_LPS];
#pragma unroll
            for (int g = 0; g < 16; g++) {
                if ((4 * g) % K2_P == 0) cp[(4 * g) / K2_P] = acc;
                acc = acc + t[g].x;
                acc = acc + t[g].y;
                acc = acc + t[g].z;
                acc = acc + t[g].w;
            }
            k2_f4 *cdst = reinterpret_cast<k2_f4 *>(check + (lane * SDRM_K2_NBUF + buf) * SDRM_K2_LPS);
#pragma unroll
            for (int g = 0; g < SDRM_K2_LPS / 4; g++) {
                cdst[g] = k2_f4{cp[4 * g], cp[4 * g + 1], cp[4 * g + 2], cp[4 * g + 3]};
            }
        } else if (FAST && role >= 2 && role <= 4) {
            // ---- stage s -> s + 1, whole block
            const int stage = role - 2;
            const int k = it - 1 - 2 * stage;
            const int buf = k % SDRM_K2_NBUF;
            const int row = stage * SDRM_K2_SLOTS + slot_h;
            float *ring = rings + ((size_t) stage * b.dc_group + slot_h) * rpitch;
            const uint32_t base = bslot + (uint32_t) n_lane;
            int from = (int) base - (int) L.s.L;
            from += from < 0 ? (int) L.s.rcap : 0;
            float t[K2_P], v[K2_P], dl[K2_P];
            k2_load_p(ts + row * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK + n_lane, t);
            const float cp = check[(row * SDRM_K2_NBUF + buf) * SDRM_K2_LPS + L.q];
            k2_load_p(ring + from, dl);
#ifdef SDRM_K2_DEBUG_STAGE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long d1 = __builtin_amdgcn_s_memtime();
#endif
            k2_quotients(L, t, cp, v);
#ifdef SDRM_K2_DEBUG_STAGE
            asm volatile("" : "+v"(v[0]), "+v"(v[15]));
            const unsigned long long d2 = __builtin_amdgcn_s_memtime();
#endif
            k2_store_p(ring + base, v);
            if (base < SDRM_K2_MIRROR) {
                k2_store_p(ring + L.s.rcap + base, v);
            }
#ifdef SDRM_K2_DEBUG_STAGE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long d3 = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
            for (int i = 0; i < K2_P; i++) t[i] = sdrm_boxcar_term(v[i], dl[i]);
            k2_store_p(ts + (row + SDRM_K2_SLOTS) * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK + n_lane, t);
            bslot += SDRM_K2_BLK;
            bslot -= bslot >= L.s.rcap ? L.s.rcap : 0u;
#ifdef SDRM_K2_DEBUG_STAGE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long d4 = __builtin_amdgcn_s_memtime();
            if (stamp && role == 2) {
                dbg[0] += d1 - w0; dbg[1] += d2 - d1; dbg[2] += d3 - d2; dbg[3] += d4 - d3;
            }
#endif
        } else if (FAST && role == 5) {
            // ---- output, whole block
            const int k = it - 7;
            const int buf = k % SDRM_K2_NBUF;
            const int row = 3 * SDRM_K2_SLOTS + slot_h;
            const int n0 = k * SDRM_K2_BLK + n_lane;
            float t[K2_P], v[K2_P], xd[K2_P], o[K2_P];
            k2_load_p(ts + row * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK + n_lane, t);
            const float cp = check[(row * SDRM_K2_NBUF + buf) * SDRM_K2_LPS + L.q];
            k2_load_p(xdt + (slot_h * 2 + (k & 1)) * SDRM_K2_BLK + n_lane, xd);
            k2_quotients(L, t, cp, v);
            float probe = 0.0f;
#pragma unroll
            for (int i = 0; i < K2_P; i++) {
                o[i] = xd[i] - v[i];
                probe = fmaf(o[i], 0.0f, probe);  // NaN as soon as one of them is not finite
                amax = fmaxf(amax, fabsf(o[i]));  // (v_max3_f32 with |.| modifiers: two outputs per instruction)
            }
            odd |= probe != probe;
            if (HAND) {
                publish(k, n0, o, K2_P, true, odd | !(amax < tame5));
            } else {
                k2_store_p(L.out + n0, o);
            }
        } else if (FAST) {
        } else
        if (role == 0) {
            const int k = it - 2 * (lane >> 4);
            if (k >= 0 && k < nb) {
                const int buf = k % SDRM_K2_NBUF;
                const k2_f4 *row = reinterpret_cast<const k2_f4 *>(ts + lane * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK);
                k2_f4 t[16];
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    t[g] = row[g];
                }
                float cp[SDRM_K2_LPS];
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    if ((4 * g) % K2_P == 0) cp[(4 * g) / K2_P] = acc;
                    acc = acc + t[g].x;
                    acc = acc + t[g].y;
                    acc = acc + t[g].z;
                    acc = acc + t[g].w;
                }
                k2_f4 *cdst = reinterpret_cast<k2_f4 *>(check + (lane * SDRM_K2_NBUF + buf) * SDRM_K2_LPS);
#pragma unroll
                for (int g = 0; g < SDRM_K2_LPS / 4; g++) {
                    cdst[g] = k2_f4{cp[4 * g], cp[4 * g + 1], cp[4 * g + 2], cp[4 * g + 3]};
                }
            }
        } else if (role <= 4) {
            // ---- stage s -> s + 1
            const int stage = role - 2;
            const int k = it - 1 - 2 * stage;
            if (k >= 0 && k < nb && L.on) {
                const int buf = k % SDRM_K2_NBUF;
                const int row = stage * SDRM_K2_SLOTS + slot_h;
                float *ring = rings + ((size_t) stage * b.dc_group + slot_h) * rpitch;
                const int n0 = k * SDRM_K2_BLK + n_lane;
                const uint32_t base = bslot + (uint32_t) n_lane;
                int from = (int) base - (int) L.s.L;
                from += from < 0 ? (int) L.s.rcap : 0;
                const bool dl_aligned = __all((from & 3) == 0);
                float t[K2_P], v[K2_P], dl[K2_P];
                k2_load_p(ts + row * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK + n_lane, t);
                const float cp = check[(row * SDRM_K2_NBUF + buf) * SDRM_K2_LPS + L.q];
                if (early) {
                    k2_load_p_any(ring + from, dl_aligned, dl);
                }
                k2_quotients(L, t, cp, v);
                if (n0 < nz) {
                    k2_store_p(ring + base, v);
                    if (base < SDRM_K2_MIRROR) {
                        k2_store_p(ring + L.s.rcap + base, v);
                    }
                    if (!early) {
                        k2_load_p_any(ring + from, dl_aligned, dl);
                    }
#pragma unroll
                    for (int i = 0; i < K2_P; i++) t[i] = (n0 + i < nz) ? sdrm_boxcar_term(v[i], dl[i]) : 0.0f;
                } else {
#pragma unroll
                    for (int i = 0; i < K2_P; i++) t[i] = 0.0f;
                }
                k2_store_p(ts + (row + SDRM_K2_SLOTS) * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK + n_lane, t);
                bslot += SDRM_K2_BLK;
                bslot -= bslot >= L.s.rcap ? L.s.rcap : 0u;
            }
        } else {
            // ---- output: x[n - 2(L-1)] - v3[n]
            const int k = it - 7;
            if (k >= 0 && k < nb) {
                const int n0 = k * SDRM_K2_BLK + n_lane;
                float o[K2_P] = {};
                int valid = 0;
                if (L.on) {
                    const int buf = k % SDRM_K2_NBUF;
                    const int row = 3 * SDRM_K2_SLOTS + slot_h;
                    float t[K2_P], v[K2_P];
                    k2_load_p(ts + row * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK + n_lane, t);
                    k2_quotients(L, t, check[(row * SDRM_K2_NBUF + buf) * SDRM_K2_LPS + L.q], v);
                    if (n0 < nz) {
                        float xd[K2_P];
                        k2_load_p(xdt + (slot_h * 2 + (k & 1)) * SDRM_K2_BLK + n_lane, xd);
                        float probe = 0.0f;
#pragma unroll
                        for (int i = 0; i < K2_P; i++) {
                            o[i] = xd[i] - v[i];
                            probe = fmaf(o[i], 0.0f, probe);  // NaN as soon as one of them is not finite
                        }
                        valid = nz - n0 < K2_P ? nz - n0 : K2_P;
                        if (valid == K2_P) {
                            odd |= probe != probe;
#pragma unroll
                            for (int i = 0; i < K2_P; i++) {
                                amax = fmaxf(amax, fabsf(o[i]));
                            }
                            if (!HAND) {
                                k2_store_p(L.out + n0, o);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < K2_P; i++) {
                                if (i < valid) {
                                    odd |= !(fabsf(o[i]) < INFINITY);
                                    amax = fmaxf(amax, fabsf(o[i]));
                                    if (!HAND) {
                                        L.out[n0 + i] = o[i];
                                    }
                                }
                            }
                        }
                    }
                }
                if (HAND) {
                    publish(k, n0, o, valid, false, odd | !(amax < tame5));
                }
            }
        }
    };
    auto run = [&](auto role_c) {
        constexpr int R = decltype(role_c)::value;
        // interior iterations of this role: its block lies wholly inside every slot's call (and, for the stage helpers,
        // the delays allow the straight-line form)
        int lo = 0, hi = 0;
        if (R == 0) {
            lo = 6, hi = nb_full;
        } else if (R == 2 && fast_wave) {
            lo = 1 + 2 * (role_rt - 2), hi = nb_full + lo;
        } else if (R == 5 && all_on) {
            lo = 7, hi = nb_full + 7;
        }
        hi = min(hi, n_it);
        lo = min(lo, hi);
        auto one = [&](auto fast_c, int it) {
            const unsigned long long w0 = stamp ? __builtin_amdgcn_s_memtime() : 0;
            body(role_c, fast_c, it, w0);
            if (stamp) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the role's LDS traffic is part of its work
                busy += __builtin_amdgcn_s_memtime() - w0;
            }
            K2_HANDOVER();
        };
        int it = 0;
        for (; it < lo; it++) {
            one(std::false_type{}, it);
        }
        for (; it < hi; it++) {
            one(std::true_type{}, it);
        }
        for (; it < n_it; it++) {
            one(std::false_type{}, it);
        }
    };
    if (role == 0) {
        run(std::integral_constant<int, 0>{});
    } else if (role == 1) {
        // ---- feeder.  The only role that reads global memory in the loop, and it writes none.  A lane takes four 4-sample
        // pieces 16 samples apart (a load instruction then covers 64 contiguous bytes of every channel row) of
        //   a = x[n], b = x[n-L] for block it + 1 of stage 0, and c = x[n - 2(L-1)] for block it - 6 of the output role;
        // iteration `it` consumes what was loaded two iterations earlier (two register sets, so that a memory latency
        // longer than an iteration stays hidden) and issues the loads for iteration it + 2 behind it.
        struct feed_set {
            float a[K2_P], b[K2_P], c[K2_P];
        };
        typedef float k2_f2 __attribute__((ext_vector_type(2)));
        const int piece = 4 * SDRM_K2_LPS;  // samples between two pieces of a lane
        // any iteration: operands fetched and used on the spot (the first and last blocks, ragged or partial groups)
        auto feed_generic = [&](int it) {
            const int k = it + 1, ko = it - 6;
            await_z((k + 1) * SDRM_K2_BLK);
            if (!L.on) {
                return;
            }
            for (int g = 0; g < K2_V4; g++) {
                const int n = k * SDRM_K2_BLK + g * piece + 4 * L.q;
                if (k < nb) {
                    float va[4] = {0.0f, 0.0f, 0.0f, 0.0f}, vb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (n < nz) {
                        x4(n, va);
                        x4(n - (int) L.s.L, vb);
                    }
                    k2_f4 t;
                    t.x = (n < nz) ? sdrm_boxcar_term(va[0], vb[0]) : 0.0f;
                    t.y = (n + 1 < nz) ? sdrm_boxcar_term(va[1], vb[1]) : 0.0f;
                    t.z = (n + 2 < nz) ? sdrm_boxcar_term(va[2], vb[2]) : 0.0f;
                    t.w = (n + 3 < nz) ? sdrm_boxcar_term(va[3], vb[3]) : 0.0f;
                    *reinterpret_cast<k2_f4 *>(ts + slot_h * SDRM_K2_TSPITCH + (k % SDRM_K2_NBUF) * SDRM_K2_BLK + g * piece + 4 * L.q) = t;
                }
                const int no = ko * SDRM_K2_BLK + g * piece + 4 * L.q;
                if (ko >= 0 && ko < nb && no < nz) {
                    float vc[4];
                    x4(no - (int) L.s.HX, vc);
                    *reinterpret_cast<k2_f4 *>(xdt + (slot_h * 2 + (ko & 1)) * SDRM_K2_BLK + g * piece + 4 * L.q) = k2_f4{vc[0], vc[1], vc[2], vc[3]};
                }
            }
        };
        // interior iterations [f0, f1): whole blocks of every slot, nothing from the carried samples, x[n-L] 16-byte aligned
        auto feed_load = [&](int it, feed_set &S) {
            await_z((it + 2) * SDRM_K2_BLK);
            const float *za = L.z + (it + 1) * SDRM_K2_BLK + 4 * L.q, *zb = za - (int) L.s.L;
            const float *zc = L.z + (it - 6) * SDRM_K2_BLK + 4 * L.q - (int) L.s.HX;
            if (HAND) {
                const size_t ia = z_row + (size_t) ((it + 1) * SDRM_K2_BLK + 4 * L.q), ib = ia - L.s.L;
                const size_t ic = z_row + (size_t) ((it - 6) * SDRM_K2_BLK + 4 * L.q) - L.s.HX;
#pragma unroll
                for (int g = 0; g < K2_V4; g++) {
                    hand_load4(z_rsrc, ia + (size_t) (g * piece), S.a + 4 * g);
                    hand_load4(z_rsrc, ib + (size_t) (g * piece), S.b + 4 * g);
                }
#pragma unroll
                for (int g = 0; g < K2_V4; g++) {
                    hand_load2(z_rsrc, ic + (size_t) (g * piece), S.c + 4 * g);
                    hand_load2(z_rsrc, ic + (size_t) (g * piece) + 2, S.c + 4 * g + 2);
                }
                return;
            }
#pragma unroll
            for (int g = 0; g < K2_V4; g++) {
                const k2_f4 ta = *reinterpret_cast<const k2_f4 *>(za + g * piece);
                const k2_f4 tb = *reinterpret_cast<const k2_f4 *>(zb + g * piece);
                S.a[4 * g] = ta.x; S.a[4 * g + 1] = ta.y; S.a[4 * g + 2] = ta.z; S.a[4 * g + 3] = ta.w;
                S.b[4 * g] = tb.x; S.b[4 * g + 1] = tb.y; S.b[4 * g + 2] = tb.z; S.b[4 * g + 3] = tb.w;
            }
#pragma unroll
            for (int g = 0; g < K2_V4; g++) {  // the delay 2(L-1) is even: 8-byte pieces
                const k2_f2 lo = *reinterpret_cast<const k2_f2 *>(zc + g * piece), hi = *reinterpret_cast<const k2_f2 *>(zc + g * piece + 2);
                S.c[4 * g] = lo.x; S.c[4 * g + 1] = lo.y; S.c[4 * g + 2] = hi.x; S.c[4 * g + 3] = hi.y;
            }
        };
        auto feed_use = [&](int it, const feed_set &S) {
            float *dt = ts + slot_h * SDRM_K2_TSPITCH + ((it + 1) % SDRM_K2_NBUF) * SDRM_K2_BLK + 4 * L.q;
            float *dx = xdt + (slot_h * 2 + ((it - 6) & 1)) * SDRM_K2_BLK + 4 * L.q;
#pragma unroll
            for (int g = 0; g < K2_V4; g++) {
                k2_f4 t;
                t.x = sdrm_boxcar_term(S.a[4 * g], S.b[4 * g]);
                t.y = sdrm_boxcar_term(S.a[4 * g + 1], S.b[4 * g + 1]);
                t.z = sdrm_boxcar_term(S.a[4 * g + 2], S.b[4 * g + 2]);
                t.w = sdrm_boxcar_term(S.a[4 * g + 3], S.b[4 * g + 3]);
                *reinterpret_cast<k2_f4 *>(dt + g * piece) = t;
                *reinterpret_cast<k2_f4 *>(dx + g * piece) = k2_f4{S.c[4 * g], S.c[4 * g + 1], S.c[4 * g + 2], S.c[4 * g + 3]};
            }
        };
        int f0 = max((l_max + SDRM_K2_BLK - 1) / SDRM_K2_BLK - 1, (hx_max + SDRM_K2_BLK - 1) / SDRM_K2_BLK + 6), f1 = nb_full - 1;
        if (!(all_on && l_mod4) || f1 - f0 < 8) {
            f0 = f1 = 0;
        }
        unsigned long long w0 = 0;
        auto begin_iter = [&]() { w0 = stamp ? __builtin_amdgcn_s_memtime() : 0; };
        auto end_iter = [&]() {
            if (stamp) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                busy += __builtin_amdgcn_s_memtime() - w0;
            }
            K2_HANDOVER();
        };
        int it = 0;
        for (; it < n_it && it < f0; it++) {
            begin_iter();
            feed_generic(it);
            end_iter();
        }
        if (f1 > f0) {
            // two register sets: iteration `it` uses what was loaded two iterations earlier, then reloads its set for it + 2
            feed_set ev, od;
            feed_load(it, ev);
            feed_load(it + 1, od);
            while (it + 3 < f1) {
                begin_iter();
                feed_use(it, ev);
                feed_load(it + 2, ev);
                end_iter();
                begin_iter();
                feed_use(it + 1, od);
                feed_load(it + 3, od);
                end_iter();
                it += 2;
            }
            begin_iter();
            feed_use(it, ev);
            end_iter();
            begin_iter();
            feed_use(it + 1, od);
            end_iter();
            it += 2;
        }
        for (; it < n_it; it++) {
            begin_iter();
            feed_generic(it);
            end_iter();
        }
    } else if (role <= 4) {
        run(std::integral_constant<int, 2>{});
    } else {
        run(std::integral_constant<int, 5>{});
    }
    __syncthreads();

    // ---- state back: running sums, ring tails, carried samples of x
    if (role == 0) {
        const sdrm_k2_slot s = slots[lane & (SDRM_K2_SLOTS - 1)];
        if (s.chan >= 0 && !s.alias) {
            sdrm_k2_state_acc(b.dc_state + b.params[s.chan].dc_state_off, b.dc_hx_cap, b.dc_l_cap)[lane >> 4] = acc;
        }
    } else if (role >= 2 && role <= 4) {
        for (int i = 0; i < SDRM_K2_P; i++) {
            const int sl = sub * SDRM_K2_P + i;
            const sdrm_k2_slot s = slots[sl];
            if (sl < b.dc_group && s.chan >= 0 && !s.alias) {
                float *st = b.dc_state + b.params[s.chan].dc_state_off;
                sdrm_k2_ring_save(rings + ((size_t) (role - 2) * b.dc_group + sl) * rpitch, s,
                                  sdrm_k2_state_tail(st, role - 2, b.dc_hx_cap, b.dc_l_cap), lane, 64);
            }
        }
    }
    if (role == 5 && L.on) {
        // NaN/Inf: the clock stage takes its general (NaN-aware) symbol for this channel; beyond the tame amplitude
        // (sdrm_kernels.h "wild channels"; an Inf raises both): it runs the channel's call from global memory
        const uint32_t bits = (odd ? SDRM_FLAG_NONFINITE : 0u) | (!(amax < sdrm_tame_level(b.params[L.s.chan], true)) ? SDRM_FLAG_WILD : 0u);
        if (bits && !HAND) {  // (with the hand-off every block has raised its own, and the clock stage may have consumed and cleared them)
            atomicOr(b.nonfinite + L.s.chan, bits);
        }
        if (HAND) {  // the last block (and those flags) are out: the channel's count reaches its end
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (L.q == 0) {
                __hip_atomic_store(b.hand_prog + L.s.chan, ((unsigned long long) b.epoch << 32) | (uint32_t) nz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (HAND) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // what follows reads the call's last samples plainly: nothing stale from this XCD's L2
    }
    // hx <- last HX samples of (hx ++ z): a slot at a time by all threads; when the call is shorter than HX the new array
    // overlaps the old one shifted by nz, so every round reads before anybody writes (and later rounds read further up)
    for (int sl = 0; sl < b.dc_group; sl++) {
        const sdrm_k2_slot s = slots[sl];
        if (s.chan < 0 || s.nz == 0 || s.alias) {
            continue;
        }
        float *hx = b.dc_state + b.params[s.chan].dc_state_off;
        const float *z = b.z + (size_t) s.chan * b.z_stride;
        for (uint32_t j0 = 0; j0 < s.HX; j0 += blockDim.x) {
            const uint32_t j = j0 + tid;
            float v = 0.0f;
            if (j < s.HX) {
                v = sdrm_k2_hx_source(s, z, hx, j);
            }
            __syncthreads();
            if (j < s.HX) {
                hx[j] = v;
            }
        }
    }
    if (b.k3_stamps != nullptr && blockIdx.x == 0 && tid == 0) {
        unsigned long long *k2s = b.k3_stamps + SDRM_STAMP_K3_WAVES(b.n_channels) * 4 + 8;  // after the K3 and K1 records
        k2s[8] = __builtin_amdgcn_s_memtime() - t_begin;
        k2s[9] = (unsigned long long) n_it;
    }
    if (stamp && lane == 0 && (wave == 0 || (wave - 1) % SDRM_K2_WPR == 0)) {
        b.k3_stamps[SDRM_STAMP_K3_WAVES(b.n_channels) * 4 + 8 + role] = busy;  // six roles: words 0..5 of the DC record
    }
#ifdef SDRM_K2_DEBUG_STAGE
    if (stamp && lane == 0 && role == 2) {  // stage 0's segments over the other roles' words (debug build only)
        b.k3_stamps[SDRM_STAMP_K3_WAVES(b.n_channels) * 4 + 8 + 6] = dbg[0] | (dbg[1] << 32);
        b.k3_stamps[SDRM_STAMP_K3_WAVES(b.n_channels) * 4 + 8 + 7] = dbg[2] | (dbg[3] << 32);
    }
#endif
    tl_mark(b, 1, 1);
}

KernelLaunch describe_dc(const DeviceBatch &b) {
    KernelLaunch k;
    if (!b.any_dc) {
        return k;
    }
    k.lds = b.dc_lds;
    static lds_grant granted, granted_hand;
    if (b.handoff) {
        allow_lds(k2_dc<true>, k.lds, &granted_hand);
        k.func = reinterpret_cast<const void *>(k2_dc<true>);
    } else {
        allow_lds(k2_dc<false>, k.lds, &granted);
        k.func = reinterpret_cast<const void *>(k2_dc<false>);
    }
    k.grid = dim3((unsigned) ((b.n_channels + b.dc_group - 1) / b.dc_group));
    k.block = dim3(64 * SDRM_K2_WAVES);
    return k;
}

void launch_dc(const DeviceBatch &b, hipStream_t s) {
    void *args[] = {(void *) &b};
    launch_described(describe_dc(b), args, s);
}

// ================================================================================================ K3

// One lane per channel, LANES (16, 32 or 64: sdrm_k3_shape_for) channels per CONSUMER wave (reference
// src/dsp/clock_recovery_mm.c:78-139 per lane); a second PRODUCER wave of the same workgroup stages the samples, so the
// two overlap.
// Producer: per step, a block (a quarter ring: 256 or 64 samples) of each of the workgroup's channels goes from global
// memory (coalesced 64-sample row segments, lane = time, prefetched one step ahead into registers) into per-channel LDS
// rings of pair elements {x[e], x[e+1]} (sdrm_kernels.h; channel pitch 22 mod 64 floats: the producer's lane = time
// writes walk through a ring, the consumer's lane = channel reads spread over the banks).  A ring holds 4 steps; mirror
// elements at both ends keep every window contiguous.  Consumer: each lane runs its own loop while staged samples last
// (lanes drop out of the exec mask as they run out); a symbol's 8 window samples are one base address plus constant
// offsets, and the next symbol's operands are fetched before the current symbol's float soft bit is stored (the int8
// conversion is k3_quantize's, behind this kernel).  Without NaN/Inf in the wave's channels the consumer runs the
// hand-scheduled loop below (k3_drain_finite), otherwise the C++ form of the same arithmetic.  One barrier per step
// hands block k to the consumer while block k+1 is written.
size_t k3_lds_bytes(int lanes, int ring, int plain) {
    const size_t rings = (size_t) lanes * (plain ? 1 : 2) * (SDRM_K3_PRE + ring + SDRM_K3_POST);
    return (rings + 129 * SDRM_K3_BANKPITCH + 4 + 4 * SDRM_K3_WAVE) * sizeof(float);
}

// Order of work inside a symbol:
//   o = ((((((((0 + w0 t0) + w1 t1) + ...) + w7 t7)   (mmse_fir_interpolator.c:188-191, fir_filter.c:116-121)
//   mm = slice(last) * o - slice(o) * last   (clock_recovery_mm.c:115): the first term is o with last's sign bit xor-ed in
//   (taken from the previous symbol in the shadow of the loads), the second an exact product with copysign(1, o), so one
//   v_fma (-slice(o) * last + first term) rounds once, exactly where the subtraction does;
//   gain_omega * mm and gain_mu * mm are one v_pk_mul_f32, (omega - mid) + lim and (omega - mid) - lim one
//   v_pk_add_f32 (per component the same IEEE operation as the scalar forms; op_sel broadcasts the common operand)
//   omega += gain_omega * mm; omega = mid + clip(omega - mid, lim)   (:119-120, branchless_clip :74-76); the clip's
//   final 0.5 * y and the addition of mid are one v_fma: halving is exact (were y the smallest denormals it would not
//   be, but then both forms return mid), so the fused form rounds once, where the addition does
//   mu = mu + omega + gain_mu * mm; ii += floor(mu); mu -= floor(mu)   (:121-123)
//   operands of the next symbol
//   while (ii < limit && oo < cap)   (:103): compared right behind the loads, combined at the symbol's end (nothing
//   waits for the compares, and they cost the dependent chain in front of the loads nothing: 316 -> 312 cycles)
//   int8 soft bit of the symbol just computed (fsk_demod.c:106), in the shadow of the loads
// one symbol of the hand-scheduled loop (see k3_drain_finite); exec is narrowed at its end
// ACC / LAST: the register the symbol is accumulated in and the one that holds the previous symbol; consecutive symbols
// swap them, which saves the copy.  CAPCMP / CAPAND: the `oo < output_len` half of the loop condition, empty in the
// variant used while the output buffer cannot fill up within the call.
// WIN: how the window's eight samples are addressed and loaded.  Pair elements: slot * 8 bytes, two ds_read2_b64 (elements
// s, s+2 | s+4, s+6; offsets in 8-byte units, PRE = 3 folded in); plain samples: slot * 4 bytes, four ds_read2_b32.
#define K3_WIN_PAIR_ADDR "v_lshl_add_u32 v64, v64, 3, %[col]\n\t"
#define K3_WIN_PAIR_LOADS \
    "ds_read_b128 v[66:69], v65\n\t" \
    "ds_read2_b64 v[74:77], v64 offset0:3 offset1:5\n\t" \
    "ds_read_b128 v[70:73], v65 offset:16\n\t" \
    "ds_read2_b64 v[78:81], v64 offset0:7 offset1:9\n\t"
#define K3_WIN_PAIR_WAIT1 "s_waitcnt lgkmcnt(2)\n\t"
#define K3_WIN_PLAIN_ADDR "v_lshl_add_u32 v64, v64, 2, %[col]\n\t"
#define K3_WIN_PLAIN_LOADS \
    "ds_read_b128 v[66:69], v65\n\t" \
    "ds_read2_b32 v[74:75], v64 offset0:3 offset1:4\n\t" \
    "ds_read2_b32 v[76:77], v64 offset0:5 offset1:6\n\t" \
    "ds_read_b128 v[70:73], v65 offset:16\n\t" \
    "ds_read2_b32 v[78:79], v64 offset0:7 offset1:8\n\t" \
    "ds_read2_b32 v[80:81], v64 offset0:9 offset1:10\n\t"
#define K3_WIN_PLAIN_WAIT1 "s_waitcnt lgkmcnt(3)\n\t"
#define K3_SYMBOL_ASM(ACC, LAST, CAPCMP, CAPAND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
    WIN_WAIT1 \
    "v_pk_mul_f32 v[66:67], v[66:67], v[74:75]\n\t" \
    "v_pk_mul_f32 v[68:69], v[68:69], v[76:77]\n\t" \
    "v_add_f32 " ACC ", 0, v66\n\t" \
    "v_add_f32 " ACC ", v67, " ACC "\n\t" \
    "v_add_f32 " ACC ", v68, " ACC "\n\t" \
    "v_add_f32 " ACC ", v69, " ACC "\n\t" \
    "s_waitcnt lgkmcnt(0)\n\t" \
    "v_pk_mul_f32 v[70:71], v[70:71], v[78:79]\n\t" \
    "v_pk_mul_f32 v[72:73], v[72:73], v[80:81]\n\t" \
    "v_add_f32 " ACC ", v70, " ACC "\n\t" \
    "v_add_f32 " ACC ", v71, " ACC "\n\t" \
    "v_add_f32 " ACC ", v72, " ACC "\n\t" \
    "v_add_f32 " ACC ", v73, " ACC "\n\t" \
    "v_xor_b32 v84, " ACC ", v86\n\t" \
    "v_bfi_b32 v83, %[mask], 1.0, " ACC "\n\t" \
    "v_fma_f32 v84, -v83, " LAST ", v84\n\t" \
    "v_pk_mul_f32 v[64:65], v[84:85], %[gg] op_sel:[0,0] op_sel_hi:[0,1]\n\t" \
    "v_add_f32 %[omega], %[omega], v64\n\t" \
    "v_sub_f32 v84, %[omega], %[mid]\n\t" \
    "v_pk_add_f32 v[84:85], v[84:85], %[ll] op_sel:[0,0] op_sel_hi:[0,1]\n\t" \
    "v_sub_f32_e64 v85, |v84|, |v85|\n\t" \
    "v_fma_f32 %[omega], v85, 0.5, %[mid]\n\t" \
    "v_add_f32 %[mu], %[mu], %[omega]\n\t" \
    "v_add_f32 %[mu], %[mu], v65\n\t" \
    "v_floor_f32 v85, %[mu]\n\t" \
    "v_sub_f32 %[mu], %[mu], v85\n\t" \
    "v_add_f32 %[pm], %[pm], v85\n\t" \
    "v_fma_f32 v65, %[mu], %[c128], %[magic]\n\t" \
    "v_and_b32 v64, %[m255], %[pm]\n\t" \
    "v_mad_u32_u24 v65, v65, %[rowb], %[bias]\n\t" \
    WIN_ADDR \
    WIN_LOADS \
    "v_cmp_lt_f32 vcc, %[pm], %[limm]\n\t" \
    CAPCMP \
    "v_and_b32 v86, %[sgn], " ACC "\n\t" \
    "global_store_dword %[off], " ACC ", %[out]\n\t" \
    "v_add_u32 %[off], 4, %[off]\n\t" \
    CAPAND \
    "s_and_b64 exec, exec, vcc\n\t"
#define K3_CAPCMP "v_cmp_ne_u32 s[74:75], %[off], %[offlast]\n\t"
#define K3_CAPAND "s_and_b64 vcc, vcc, s[74:75]\n\t"

// The FINITE symbol loop of the clock stage, scheduled by hand (same operations, same order per lane as
// sdrm_k3_fetch<true> + sdrm_k3_step<true>, which is what the CPU emulation runs).  Why by hand: one wave issues one
// instruction per ~4.4 cycles whatever its dependences, so the symbol time is the instruction count plus whatever LDS
// latency is left exposed.  Here: 36 VALU instructions (35 of them in front of the loads) (the compiler's form: 51; one more and a second SALU instruction
// while the output buffer could fill up), one SALU instruction for the loop, FOUR operand loads (two 16-byte reads for
// the MMSE row, two ds_read2_b64 for the window's pair elements), two waits, and the previous symbol's float soft bit
// stored behind the loads.
//   v64..v88 are scratch (named, so that halves of the 64-bit pairs can be addressed); everything else is allocated
//   by the compiler.  exec is narrowed as lanes run out of samples and restored on exit.
#ifndef SDRM_K3_LOOP_SKEW
#define SDRM_K3_LOOP_SKEW 0
#endif
#define K3_STORE_SLACK 32   // store instructions of the consumer that may still be in flight at a hand-over
#ifdef SDRM_K3_NO_FUSED_INT8  // A/B builds only (profiles/r04_1024_ab.txt): every shape leaves the conversion to k3_quantize
#define K3_FUSED_INT8(G) false
#define K3_FUSED_INT8_RING(ring) false
#else
#define K3_FUSED_INT8(G) (G::block >= 256)  // the staging wave converts the soft bits to int8 (else: k3_quantize)
#define K3_FUSED_INT8_RING(ring) ((ring) / 4 >= 256)
#endif
#define K3_STR2(x) #x
#define K3_STR(x) K3_STR2(x)
#define K3_LOOP_SKEW ".rept " K3_STR(SDRM_K3_LOOP_SKEW) "\n\ts_nop 0\n\t.endr\n\t"
template <bool CAP, bool PLAIN>
__device__ __forceinline__ void k3_drain_finite(sdrm_k3_lane &L, uint32_t lim, uint32_t col_addr, uint32_t bank_addr,
                                                uint32_t &off, uint32_t off_end, const float *out_base, uint32_t ring_mask) {
    // row address = bank + rowbytes * rint(mu * 128): the low 24 bits of (mu * 128 + 1.5 * 2^23) are 0x400000 + row, so a
    // 24-bit multiply-add with this bias lands on the row (the sum wraps modulo 2^32)
    const uint32_t bias = bank_addr - 0x400000u * (SDRM_K3_BANKPITCH * 4u);
    float mu = L.st.mu, omega = L.st.omega, last = L.st.last;
    int ii = L.st.ii, inc = L.st.inc;
    unsigned long long saved_exec;
    // operand pairs of the two packed instructions: {gain_omega, gain_mu} * mm, (omega - mid) + {lim, -lim}
    typedef float k3_f2 __attribute__((ext_vector_type(2)));
    const k3_f2 gains = {L.k.gain_omega, L.k.gain_mu};
    const k3_f2 limits = {L.k.omega_lim, -L.k.omega_lim};
    // Inside the loop the position is a FLOAT: pm = 1.5 * 2^23 + (ii - kept).  Positions stay below 2^22 (checked by the
    // caller), so every pm is exact, `pm += floor(mu)` is the integer addition, the low mantissa bits of pm ARE the ring
    // slot (ii - kept) & mask (negative positions included: 0x400000 + p keeps p's low bits), and `pm < limm` is
    // `ii < lim`: the conversion of floor(mu) to an integer, the integer addition and the subtraction of `kept` leave
    // the symbol's dependent chain (two instructions less per symbol).
    float pm = SDRM_RINT_MAGIC + (float) (ii - L.kept);
    const float limm = SDRM_RINT_MAGIC + (float) ((int) lim - L.kept);
#define K3_DRAIN_ASM(CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
    asm volatile( \
        "s_mov_b64 %[sv], exec\n\t" \
        /* operands of the first symbol: MMSE row (2 x 16 bytes) and the window's four pair elements */ \
        "v_sub_u32 v64, %[ii], %[kept]\n\t" \
        "v_and_b32 v64, %[m255], v64\n\t" \
        WIN_ADDR \
        "v_fma_f32 v65, %[mu], %[c128], %[magic]\n\t" \
        "v_mad_u32_u24 v65, v65, %[rowb], %[bias]\n\t" \
        WIN_LOADS \
        "v_mov_b32 v87, %[last]\n\t" \
        "v_and_b32 v86, %[sgn], %[last]\n\t" \
        "v_mov_b32 v88, %[off]\n" \
        /* the loop starts on a 64-byte boundary (+ SDRM_K3_LOOP_SKEW s_nops): a hand-scheduled stream of 4- and 8-byte */ \
        /* instructions runs ~15 % slower when code in front of it moves it by 4 bytes (2.58 -> 2.96 ms per chunk seen) */ \
        ".p2align 6\n\t" \
        K3_LOOP_SKEW \
        "1:\n\t" \
        /* eight symbols per trip: the taken branch back costs a lone wave ~25 cycles */ \
        K3_SYMBOL_ASM("v82", "v87", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execz 2f\n\t" \
        K3_SYMBOL_ASM("v87", "v82", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execz 2f\n\t" \
        K3_SYMBOL_ASM("v82", "v87", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execz 2f\n\t" \
        K3_SYMBOL_ASM("v87", "v82", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execz 2f\n\t" \
        K3_SYMBOL_ASM("v82", "v87", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execz 2f\n\t" \
        K3_SYMBOL_ASM("v87", "v82", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execz 2f\n\t" \
        K3_SYMBOL_ASM("v82", "v87", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execz 2f\n\t" \
        K3_SYMBOL_ASM("v87", "v82", CMP, AND, WIN_ADDR, WIN_LOADS, WIN_WAIT1) \
        "s_cbranch_execnz 1b\n" \
        "2:\n\t" \
        "s_waitcnt lgkmcnt(0)\n\t" \
        "s_mov_b64 exec, %[sv]\n\t" \
        /* back to integers: the advance of the lane's last symbol (its floor is still in v85) and the position */ \
        "v_cvt_i32_f32 %[inc], v85\n\t" \
        "v_sub_f32 v64, %[pm], %[magic]\n\t" \
        "v_cvt_i32_f32 v64, v64\n\t" \
        "v_add_u32 %[ii], v64, %[kept]\n\t" \
        /* a lane's last symbol sits in v82 after an odd number of symbols, else in v87 */ \
        "v_xor_b32 v88, v88, %[off]\n\t" \
        "v_and_b32 v88, 4, v88\n\t" \
        "v_cmp_eq_u32 vcc, 4, v88\n\t" \
        "s_nop 1\n\t" \
        "v_cndmask_b32 %[last], v87, v82, vcc\n\t" \
        : [mu] "+v"(mu), [omega] "+v"(omega), [last] "+v"(last), [ii] "+v"(ii), [inc] "+v"(inc), [off] "+v"(off), \
          [pm] "+v"(pm), [sv] "=&s"(saved_exec) \
        : [kept] "v"(L.kept), [limm] "v"(limm), [col] "v"(col_addr), [magic] "v"(SDRM_RINT_MAGIC), [gg] "v"(gains), \
          [ll] "v"(limits), [mid] "v"(L.k.omega_mid), \
          [offlast] "v"(off_end - 4u), [c128] "v"(128.0f), [mask] "s"(0x7fffffffu), [bias] "s"(bias), [out] "s"(out_base), \
          [m255] "s"(ring_mask), [rowb] "n"(SDRM_K3_BANKPITCH * 4), [sgn] "s"(0x80000000u) \
        : "memory", "vcc", "s74", "s75", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", \
          "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88")
    if (PLAIN) {
        if (CAP) {
            K3_DRAIN_ASM(K3_CAPCMP, K3_CAPAND, K3_WIN_PLAIN_ADDR, K3_WIN_PLAIN_LOADS, K3_WIN_PLAIN_WAIT1);
        } else {
            K3_DRAIN_ASM("", "", K3_WIN_PLAIN_ADDR, K3_WIN_PLAIN_LOADS, K3_WIN_PLAIN_WAIT1);
        }
    } else {
        if (CAP) {
            K3_DRAIN_ASM(K3_CAPCMP, K3_CAPAND, K3_WIN_PAIR_ADDR, K3_WIN_PAIR_LOADS, K3_WIN_PAIR_WAIT1);
        } else {
            K3_DRAIN_ASM("", "", K3_WIN_PAIR_ADDR, K3_WIN_PAIR_LOADS, K3_WIN_PAIR_WAIT1);
        }
    }
#undef K3_DRAIN_ASM
    L.st.mu = mu;
    L.st.omega = omega;
    L.st.last = last;
    L.st.ii = ii;
    L.st.inc = inc;
}

// HAND: the in-call hand-off (the stages in front are still running; DeviceBatch::handoff) -- a build of its own (two shapes:
// 16 x 1024 and 32 x 512, what batches small enough for the hand-off take), so that the ordinary builds carry none of its code
template <int LANES, int RING, bool PLAIN, bool HAND = false>
__global__ __launch_bounds__(128) void k3_clock(DeviceBatch b) {
    typedef sdrm_k3_geom<LANES, RING, PLAIN> G;
    extern __shared__ __attribute__((aligned(16))) float k3_lds[];
    float *bank_rev = k3_lds;                       // [129*8] at LDS offset 0: a row is two aligned ds_read_b128
    float *ring = bank_rev + ((129 * SDRM_K3_BANKPITCH + 3) & ~3);  // [LANES][CPITCH]
    int *nz_sh = reinterpret_cast<int *>(ring + G::lanes * G::cpitch);  // [64] samples per channel
    int *dc_sh = nz_sh + SDRM_K3_WAVE;                                           // [64] reads dcout (1) or z (0)
    int *safe_sh = dc_sh + SDRM_K3_WAVE;                                         // [64] float soft bits the staging wave may convert
    int *flag_sh = safe_sh + SDRM_K3_WAVE;                                       // [64] hand-off: the call's SDRM_FLAG_* bits seen so far, per channel
    if (b.placed != nullptr && threadIdx.x == 0) {
        atomicAdd(b.placed + 1, 1u);  // this workgroup has its CU (k_hold_until)
    }
    const int lane = threadIdx.x & 63;
    const bool producer = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)) != 0;
    const int c0 = blockIdx.x * G::lanes;
    const int c = c0 + lane;
    const bool active = lane < G::lanes && c < b.n_channels;
    const int nrows = b.n_channels - c0 < G::lanes ? b.n_channels - c0 : G::lanes;
    for (int k = threadIdx.x; k < 129 * 8; k += 128) {
        bank_rev[(k >> 3) * SDRM_K3_BANKPITCH + (k & 7)] = b.mmse_bank[(k & ~7) + 7 - (k & 7)];  // rows reversed once: tap j meets window sample j
    }
    tl_mark(b, 2, 0);
    sdrm_k3_lane L;
    L.kept = 0;
    L.nz = 0;
    L.oo = 0;
    L.cap = 0;
    L.st.mu = 0.0f;
    L.st.omega = 0.0f;
    L.st.last = 0.0f;
    L.st.ii = 0;
    L.st.inc = 0;
    L.k.omega_mid = L.k.omega_lim = L.k.gain_omega = L.k.gain_mu = 0.0f;
    float *my_col = ring + (lane & (G::lanes - 1)) * G::cpitch;  // this channel's ring
    sdrm_clock_state *cs = b.clock_state + (active ? c : 0);
    bool clean = true;
    uint32_t flagged = 0;
    // a channel that takes no part in this call keeps its state as it is: the lane stays out of everything below
    // ... and so does a generic channel (k3_clock_generic runs it behind this kernel and writes its output count)
    const bool absent = active && (b.ctl[c].absent != 0 || b.params[c].generic != 0);
    // a WILD channel (sdrm_kernels.h: a sample beyond the amplitude up to which the timing loop provably advances, in this
    // call or among what the last one left behind) takes no part in the ring-based loops either: its samples are staged like
    // the others' (the staging stays uniform) but its lane never steps, and sdrm_k3_rescue runs its call behind the loops
    bool wild = false;
    if (!producer) {
        int uses_dc = 0;
        if (active && !absent) {
            const sdrm_chan_params p = b.params[c];
            L.nz = (int) b.ctl[c].nz;
            uses_dc = p.dc_len != 0;
            // (in-call hand-off: the stages in front are still running, their flags arrive block by block through flag_sh)
            flagged = HAND ? 0u : b.nonfinite[c];
            const uint32_t carried = cs->poison;
            wild = ((flagged | carried) & SDRM_FLAG_WILD) != 0 || !(p.amp_safe > 0.0f);
            if (!wild) {
                L.k.omega_mid = p.omega_mid;
                L.k.omega_lim = p.omega_lim;
                L.k.gain_omega = p.gain_omega;
                L.k.gain_mu = p.gain_mu;
                L.cap = p.max_len;
                L.kept = (int) cs->kept;
                L.st.mu = cs->mu;
                L.st.omega = cs->omega;
                L.st.last = cs->last;
                clean = (flagged == 0) & (carried == 0);
                for (int j = 0; j < L.kept; j++) {
                    sdrm_k3_ring_put<G>(my_col, j - L.kept, cs->hist[j]);
                }
            }
        }
        // A channel that takes no part in the call (absent: a batcher slot without a buffer this round, or waiting for a client;
        // generic) says -1: it does not decide how the workgroup's blocks are staged -- its row is staged along with the others
        // (allocated memory, whatever it holds, into the ring of a lane that never steps) so that the live channels beside it keep
        // the unpredicated path (a 512-slot batcher with 3 live clients ran its clock stage in 3.8 ms instead of 2.5:
        // profiles/r05_node_schedule.txt)
        nz_sh[lane] = absent ? -1 : L.nz;
        dc_sh[lane] = (active && absent) ? (int) (b.params[c].dc_len != 0) : uses_dc;
        flag_sh[lane] = 0;
    }
    __shared__ unsigned hw_sh;  // diagnostics: where the producer wave runs (HW_ID: SIMD in bits 5:4, CU 11:8, SE 15:13)
    if (producer && lane == 0) {
        hw_sh = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_REG_HW_ID, whole register
    }
    __syncthreads();
    const int my_nz = nz_sh[lane];
    const int my_dc = dc_sh[lane];
    // a workgroup whose 64 channels all read the same stream (all with / all without DC blocker) and all have a full
    // block takes the unpredicated staging path: one base pointer, rows z_stride apart
    const bool same_src = __all(lane >= nrows || my_dc) || __all(lane >= nrows || !my_dc);
    int max_nz = 0, min_nz = 0x7fffffff;
    for (int r = 0; r < nrows; r++) {
        int v = __builtin_amdgcn_readlane(my_nz, r);
        max_nz = v > max_nz ? v : max_nz;
        min_nz = (v >= 0 && v < min_nz) ? v : min_nz;  // (-1: the row takes no part)
    }
    // ... up to the end of the LONGEST row: a row that has ended (a mixed-rate batch: 131072 / 26214 / 16384 samples per call side by
    // side) is staged on with whatever its memory holds -- the rows are z_stride long -- into a ring whose lane has finished: its
    // consumer lane wrote its state back when its own samples ended (below), before those slots are overwritten
    const bool uniform = same_src && nrows == G::lanes;
    (void) min_nz;
    const int nblocks = (max_nz + G::block - 1) / G::block;


    if (producer) {
        const float *row0 = (__builtin_amdgcn_readfirstlane(my_dc) ? b.dcout : b.z) + (size_t) c0 * b.z_stride;
        float pre[G::lanes * G::segs];  // prefetched row segments: pre[r * SEGS + h] = sample h*64 + lane of channel c0+r's next block
#define K3_ROW_SRC(r) ((__builtin_amdgcn_readlane(my_dc, (r)) ? b.dcout : b.z) + (size_t) (c0 + (r)) * b.z_stride)
#define K3_LOAD_PLAIN(ptr) (*(ptr))
        // in-call hand-off: the sample was written by a kernel that is still running, maybe on another XCD -- read it with
        // device scope (past this XCD's L2; each sample is read once, so nothing is lost by that)
#define K3_LOAD_DEVICE(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define K3_ISSUE_AS(k, LOAD)                                                                                 \
    {                                                                                                        \
        const int n_ = (k) * G::block + lane;                                                            \
        if (uniform && ((k) + 1) * G::block <= max_nz) {                                                 \
            const float *p_ = row0 + n_;                                                                      \
            _Pragma("unroll") for (int r = 0; r < G::lanes; r++) {                                       \
                _Pragma("unroll") for (int h = 0; h < G::segs; h++) {                                    \
                    pre[r * G::segs + h] = LOAD(p_ + h * SDRM_K3_WAVE);                                  \
                }                                                                                            \
                p_ += b.z_stride;                                                                             \
            }                                                                                                \
        } else {                                                                                             \
            _Pragma("unroll") for (int r = 0; r < G::lanes; r++) {                                       \
                _Pragma("unroll") for (int h = 0; h < G::segs; h++) {                                    \
                    pre[r * G::segs + h] = 0.0f;                                                         \
                    if (r < nrows && n_ + h * SDRM_K3_WAVE < __builtin_amdgcn_readlane(my_nz, r)) {           \
                        pre[r * G::segs + h] = LOAD(K3_ROW_SRC(r) + n_ + h * SDRM_K3_WAVE);              \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }
        // In-call hand-off: before block k of this call is fetched, every channel's samples up to its end must be in memory --
        // the DC blocker's count for the channels behind one (hand_prog), the front-end's tile stamps for the others -- and
        // the flags those samples raised go to the consumer wave with the block (flag_sh, read behind the hand-over barrier).
        int my_ready = 0;  // lane = channel (row): samples of this call known to be there
        const uint32_t tile_m = (HAND && lane < nrows) ? b.params[c0 + lane].tile_m : 1u;
        auto await_block = [&](int k) {
            const int cr = c0 + lane;
            const bool mine = lane < nrows;
            const int need = ((k) + 1) * G::block < my_nz ? ((k) + 1) * G::block : my_nz;
            auto look = [&]() {
                if (!mine || my_ready >= need) {
                    return true;
                }
                if (my_dc) {
                    my_ready = (int) hand_prog_of(b, cr);
                } else {
                    while (my_ready < need && hand_tile_done(b, cr, tile_m, my_ready)) {
                        my_ready = (int) (((uint32_t) my_ready / tile_m + 1u) * tile_m);
                    }
                }
                return my_ready >= need;
            };
            // (a look is a round trip to memory, ~1 us: none while what the last one saw still covers the block -- the stage in
            // front runs ahead -- and the flags are read behind a look only: whatever was raised for samples an earlier look
            // covered was read then)
            if (__all(!mine || my_ready >= need)) {
                return;
            }
            bool there = look();
            for (int looks = 0; !__all(there) && looks < SDRM_HAND_MAX_LOOKS; looks++) {
                __builtin_amdgcn_s_sleep(8);
                there = look();
            }
            if (!__all(there)) {
                atomicOr(b.counters + 1, 1u);  // gave up: the call is void (SDRM_OUT_LEN_FAILED)
                my_ready = 0x7fffffff;         // ... and nobody waits again (one bound per kernel, not one per block)
            }
            if (mine && lane < SDRM_K3_WAVE) {
                flag_sh[lane] = (int) __hip_atomic_load(b.nonfinite + cr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        };
#define K3_ISSUE(k)                                                                                          \
    if (HAND) {                                                                                              \
        await_block(k);                                                                                      \
        K3_ISSUE_AS(k, K3_LOAD_DEVICE)                                                                       \
    } else {                                                                                                 \
        K3_ISSUE_AS(k, K3_LOAD_PLAIN)                                                                        \
    }
        // transpose the prefetched rows into the ring; the full-block path writes rows directly and refreshes the
        // mirror rows with a small copy pass, the ragged path goes element by element through sdrm_k3_ring_put
#define K3_COMMIT(k)                                                                                         \
    {                                                                                                        \
        const int n_ = (k) * G::block + lane;                                                            \
        if (uniform && ((k) + 1) * G::block <= max_nz) {                                                 \
            if (G::plain) {                                                                                   \
                _Pragma("unroll") for (int h = 0; h < G::segs; h++) {                                         \
                    float *dst_ = ring + (((n_ + h * SDRM_K3_WAVE) & (G::ring - 1)) + SDRM_K3_PRE);           \
                    _Pragma("unroll") for (int r = 0; r < G::lanes; r++) {                                    \
                        dst_[r * G::cpitch] = pre[r * G::segs + h];                                            \
                    }                                                                                         \
                }                                                                                             \
                __builtin_amdgcn_wave_barrier();                                                              \
                /* mirrors, one channel per lane: the first 8 samples above the ring, the last 3 below it */ \
                if ((((k) * G::block) & (G::ring - 1)) == 0 && lane < G::lanes) {                             \
                    _Pragma("unroll") for (int j = 0; j < SDRM_K3_POST; j++) {                                \
                        my_col[j + G::ring + SDRM_K3_PRE] = my_col[j + SDRM_K3_PRE];                          \
                    }                                                                                         \
                }                                                                                             \
                if (((((k) + 1) * G::block) & (G::ring - 1)) == 0 && lane < G::lanes) {                       \
                    _Pragma("unroll") for (int j = 0; j < SDRM_K3_PRE; j++) {                                 \
                        my_col[j] = my_col[j + G::ring];                                                      \
                    }                                                                                         \
                }                                                                                             \
            } else {                                                                                          \
            /* sample n = first half of element n, second half of element n - 1 (of every channel r) */      \
            _Pragma("unroll") for (int h = 0; h < G::segs; h++) {                                        \
                const int nh_ = n_ + h * SDRM_K3_WAVE;                                                        \
                float *lo_ = ring + 2 * ((nh_ & (G::ring - 1)) + SDRM_K3_PRE);                           \
                float *hi_ = ring + 2 * (((nh_ - 1) & (G::ring - 1)) + SDRM_K3_PRE) + 1;                 \
                _Pragma("unroll") for (int r = 0; r < G::lanes; r++) {                                   \
                    lo_[r * G::cpitch] = pre[r * G::segs + h];                                      \
                    hi_[r * G::cpitch] = pre[r * G::segs + h];                                      \
                }                                                                                            \
            }                                                                                                \
            __builtin_amdgcn_wave_barrier();                                                                  \
            /* mirrors, one channel per lane: elements 0..7 above the ring when this block wrote them (and the */ \
            /* second half of the last element below it), the last three elements below the ring when it wrote those */ \
            if ((((k) * G::block) & (G::ring - 1)) == 0 && lane < G::lanes) {                  \
                _Pragma("unroll") for (int j = 0; j < 2 * SDRM_K3_POST; j++) {                                \
                    my_col[j + 2 * (G::ring + SDRM_K3_PRE)] = my_col[j + 2 * SDRM_K3_PRE];               \
                }                                                                                            \
                my_col[2 * (SDRM_K3_PRE - 1) + 1] = my_col[2 * (G::ring - 1 + SDRM_K3_PRE) + 1];         \
            }                                                                                                \
            if (((((k) + 1) * G::block) & (G::ring - 1)) == 0 && lane < G::lanes) {            \
                _Pragma("unroll") for (int j = 0; j < 2 * SDRM_K3_PRE; j++) {                                 \
                    my_col[j] = my_col[j + 2 * G::ring];                                                 \
                }                                                                                            \
            }                                                                                                \
            }                                                                                                \
        } else {                                                                                             \
            _Pragma("unroll") for (int r = 0; r < G::lanes; r++) {                                       \
                const int nz_r = r < nrows ? __builtin_amdgcn_readlane(my_nz, r) : 0;                         \
                _Pragma("unroll") for (int h = 0; h < G::segs; h++) {                                    \
                    if (n_ + h * SDRM_K3_WAVE < nz_r) {                                                       \
                        sdrm_k3_ring_put<G>(ring + r * G::cpitch, n_ + h * SDRM_K3_WAVE, pre[r * G::segs + h]); \
                    }                                                                                        \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }
        // int8 soft bits (reference src/dsp/fsk_demod.c:106).  The recursion's wave pays for every instruction it issues, so
        // it only stores the float soft bits; this wave, idle between two staging steps, converts them: whenever a channel
        // has 64 finished ones (`safe`: published by the consumer at each hand-over, behind an s_waitcnt that lets only its
        // newest stores be outstanding) one coalesced 256-byte read and one 64-byte write, the rest after the last step.
        // Sixteen channels per step, and software-pipelined: a step converts and stores what the previous step read, so
        // that no memory latency ever sits between this wave and the hand-over the consumer is waiting for.
        // Only for staging steps of 256 samples (the shapes of up to 1280 channels): with 128- or 64-sample steps the
        // consumer is back for the next block within 1-2 us, less than a memory round trip, and every load this wave has
        // in flight delays the samples it must deliver (measured: 2048 channels 4.96 -> 6.0 ms per call, 4096: 9.0 -> 17);
        // those shapes leave the conversion to the pointwise kernel k3_quantize behind the stage.
        uint32_t done = 0;  // lane = channel: soft bits read for conversion so far (a multiple of 64 until the end)
        float qv[16];
        uint32_t q_at[16], q_cnt[16];
        int q_r0 = -1;      // the group of sixteen channels whose values sit in qv (-1: none)
#pragma unroll
        for (int j = 0; j < 16; j++) {
            qv[j] = 0.0f;
            q_at[j] = q_cnt[j] = 0;
        }
        auto q_store = [&]() {
            if (q_r0 >= 0) {
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    if ((uint32_t) lane < q_cnt[j]) {
                        b.out_i8[(size_t) (c0 + q_r0 + j) * b.out_stride + q_at[j] + lane] = sdrm_soft_to_i8(qv[j]);
                    }
                }
            }
            q_r0 = -1;
        };
        // read the next 64 (last: whatever is left) finished soft bits of the channels r0 .. r0 + 15; false: nothing to read
        auto q_load = [&](int r0, bool last) {
            const uint32_t safe = lane < G::lanes ? (uint32_t) safe_sh[lane] : 0u;
            const uint32_t pend = safe - done;
            const unsigned long long todo = __ballot(last ? pend > 0u : pend >= 64u);
            if (((todo >> r0) & 0xffffull) == 0) {
                return false;
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int r = r0 + j;
                q_at[j] = (uint32_t) __builtin_amdgcn_readlane((int) done, r);
                const uint32_t left = (uint32_t) __builtin_amdgcn_readlane((int) safe, r) - q_at[j];
                q_cnt[j] = ((todo >> r) & 1ull) ? (left < 64u ? left : 64u) : 0u;
                if ((uint32_t) lane < q_cnt[j]) {
                    // Workgroup-scope load: the consumer wave of this workgroup stored these floats (write-through, complete:
                    // its s_waitcnt) and nobody on this CU has read the two cache lines of this 256-byte chunk before, so the
                    // read misses the L1 and hits the L2 the stores went through.  (An agent-scope load bypasses the L2 on
                    // this part -- one L2 per XCD -- and fetched the 27 MB per 256 channels from memory again.)
                    qv[j] = __hip_atomic_load(b.out_f32 + (size_t) (c0 + r) * b.out_stride + q_at[j] + lane, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                done = lane == r ? q_at[j] + q_cnt[j] : done;
            }
            q_r0 = r0;
            return true;
        };
        if (nblocks > 0) {
            K3_ISSUE(0)
        }
        // diagnostics (workgroup 0 with the stamps on): cycles the staging wave spends writing the ring, converting soft bits,
        // fetching the next block, and at the hand-over barrier
        const bool pstamp = b.k3_stamps != nullptr && blockIdx.x == 0;
        unsigned long long pt[4] = {0, 0, 0, 0};
        for (int k = 0; k < nblocks; k++) {
            const unsigned long long p0 = pstamp ? __builtin_amdgcn_s_memtime() : 0;
            K3_COMMIT(k)
            if (pstamp) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            const unsigned long long p1 = pstamp ? __builtin_amdgcn_s_memtime() : 0;
            if (K3_FUSED_INT8(G)) {
                q_store();
                q_load((k % (G::lanes / 16)) * 16, false);
            }
            const unsigned long long p2 = pstamp ? __builtin_amdgcn_s_memtime() : 0;
            if (k + 1 < nblocks) {
                K3_ISSUE(k + 1)
            }
            const unsigned long long p3 = pstamp ? __builtin_amdgcn_s_memtime() : 0;
            // block k is in the ring; the consumer works on it while block k+1 is written
            if (K3_FUSED_INT8(G)) {
                // LDS traffic only: the loads and stores above stay in flight across the hand-over
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                __syncthreads();
            }
            if (pstamp) {
                const unsigned long long p4 = __builtin_amdgcn_s_memtime();
                pt[0] += p1 - p0;
                pt[1] += p2 - p1;
                pt[2] += p3 - p2;
                pt[3] += p4 - p3;
            }
        }
        if (pstamp && lane == 0) {
            unsigned long long *ps = b.k3_stamps + SDRM_STAMP_K3_WAVES(b.n_channels) * 4 + 18;
            ps[0] = pt[0];
            ps[1] = pt[1];
            ps[2] = pt[2];
            ps[3] = pt[3];
        }
        if (!K3_FUSED_INT8(G)) {
            tl_mark(b, 2, 1);
            return;  // short staging steps: k3_quantize converts behind the kernel
        }
        q_store();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the consumer has stored its last soft bit (and waited for the stores)
        for (bool more = true; more;) {
            more = false;
            for (int r0 = 0; r0 < G::lanes; r0 += 16) {
                if (q_load(r0, true)) {
                    q_store();
                    more = true;
                }
            }
        }
#undef K3_ISSUE
#undef K3_ISSUE_AS
#undef K3_LOAD_PLAIN
#undef K3_LOAD_DEVICE
#undef K3_COMMIT
#undef K3_ROW_SRC
        tl_mark(b, 2, 1);
        return;
    }

    // ------------------------------------------------------------------ consumer wave
    // a long dependent chain on one wave: let it win issue arbitration against the throughput kernels sharing the SIMD
    __builtin_amdgcn_s_setprio(3);
    // the stage writes the float soft bits; k3_quantize turns them into the int8 output behind it
    float *of = b.out_f32 + (size_t) (active ? c : 0) * b.out_stride;
    bool wave_clean = __all(clean);
    const bool positions_fit = __all(L.nz < (1 << 22) - 2 * SDRM_CLOCK_HCAP);  // the hand-scheduled loop counts positions in a float's mantissa
    // Run every lane's loop as far as the staged samples allow.  A lane that cannot step now cannot step later in
    // the same block either, so the loop only ever shrinks the exec mask.  The operands of the NEXT symbol are issued
    // before the current one is quantised and stored, so part of the LDS latency hides behind that work.
    float *pf = of;
    const uint32_t end_lo = (uint32_t) (uintptr_t) (of + L.cap);
    // LDS-typed, opaque copy of this lane's ring address: the compiler then keeps it in one register and reaches the
    // window through the ds_read2 immediate offsets instead of re-adding the ring's LDS offset per symbol
    typedef const __attribute__((address_space(3))) float *lds_cf;
    lds_cf col_l = (lds_cf) my_col;
    asm volatile("" : "+v"(col_l));
#define K3_DRAIN(FIN)                                                                                        \
    if (sdrm_k3_can_step(L, lim)) {                                                                          \
        sdrm_k3_operands F;                                                                                   \
        sdrm_k3_fetch<FIN, G>(L, col_l, bank_rev, F);                                                            \
        do {                                                                                                 \
            const float soft = sdrm_k3_step<FIN>(L, F);                                                       \
            sdrm_k3_fetch<FIN, G>(L, col_l, bank_rev, F);                                                        \
            *pf++ = soft;                                                                                     \
            /* `oo < output_len`: the output pointer stands in for the symbol count (low words suffice) */    \
        } while (((uint32_t) L.st.ii < lim) & ((uint32_t) (uintptr_t) pf != end_lo));                        \
        L.oo = (uint32_t) (pf - of);                                                                          \
    }
    // the hand-scheduled loop addresses the output as uniform base (the workgroup's first channel) + 32-bit byte offset
    const float *wg_out = b.out_f32 + (size_t) c0 * b.out_stride;
    const uint32_t off_base = (uint32_t) ((active ? lane : 0) * (size_t) b.out_stride * sizeof(float));
    uint32_t off = off_base;
    const uint32_t off_end = off_base + L.cap * (uint32_t) sizeof(float);
    const uint32_t col_addr = (uint32_t) (uintptr_t) col_l;
    const uint32_t bank_addr = (uint32_t) (uintptr_t) (lds_cf) bank_rev;
    unsigned long long t_wait = 0, t_drain = 0, n_iter = 0;
    const unsigned long long real0 = b.k3_stamps ? __builtin_amdgcn_s_memrealtime() : 0;  // 100 MHz reference clock
    // what a call leaves behind for the next (reference clock_recovery_mm.c:127-135) -- written when the LANE's samples end, which
    // in a workgroup of unequal rows is before the workgroup's last block: the ring slots that hold the carried samples are
    // intact then (the block being staged meanwhile goes elsewhere) and are not later
    bool finished = false;
    auto finish_lane = [&]() {
        int from_n, new_kept;
        sdrm_k3_finish(L, &from_n, &new_kept);
        for (int j = 0; j < new_kept; j++) {
            cs->hist[j] = sdrm_k3_ring_get<G>(my_col, from_n + j);
        }
        cs->kept = (uint32_t) new_kept;
        cs->mu = L.st.mu;
        cs->omega = L.st.omega;
        cs->last = L.st.last;
        // the carried samples come from this call's stream; a loop state that is no longer finite (an Inf sample makes
        // omega and mu NaN two symbols later) keeps the channel off the finite-only fast path until it is reset
        cs->poison = (flagged != 0 || !(fabsf(L.st.mu) < INFINITY) || !(fabsf(L.st.omega) < INFINITY) || !(fabsf(L.st.last) < INFINITY)) ? SDRM_FLAG_NONFINITE : 0u;
        b.nonfinite[c] = 0;        // consumed: the slot is clean for its next use
        b.out_len[c] = L.oo;
        finished = true;
    };
    for (int k = 0; k <= nblocks; k++) {
        // k == nblocks: nothing new, only drains what the carried history alone allows (nz == 0 case)
        unsigned long long t0 = b.k3_stamps ? __builtin_amdgcn_s_memtime() : 0;
        if (k < nblocks) {
            // float soft bits the staging wave may convert now: all but those of the newest stores (one store instruction
            // per symbol of the wave's loop; stores complete in order, so behind this wait at most K3_STORE_SLACK are open)
            if (K3_FUSED_INT8(G)) {
                asm volatile("s_waitcnt vmcnt(" K3_STR(K3_STORE_SLACK) ")" ::: "memory");
                safe_sh[lane] = L.oo > K3_STORE_SLACK ? (int) (L.oo - K3_STORE_SLACK) : 0;
            }
            __syncthreads();  // block k staged
            if (HAND) {
                // what the stages in front have flagged up to this block: a channel that has turned wild stops here (its call
                // is run again from its start by sdrm_k3_rescue, the state it started from is untouched), NaN/Inf moves the
                // wave to the general form of the symbol
                const uint32_t f = active && !absent ? (uint32_t) flag_sh[lane] : 0u;
                flagged |= f;
                if ((f & SDRM_FLAG_WILD) != 0 && !wild) {
                    wild = true;
                    L.cap = 0;
                }
                clean = wild || (clean && f == 0);
                wave_clean = __all(clean);
            }
        }
        unsigned long long t1 = b.k3_stamps ? __builtin_amdgcn_s_memtime() : 0;
        int avail = (k + 1) * G::block;
        avail = avail < L.nz ? avail : L.nz;
        const uint32_t lim = active ? sdrm_k3_limit(L, avail) : 0u;
        const uint32_t oo0 = L.oo;
        if (wave_clean && positions_fit) {
            if (sdrm_k3_can_step(L, lim)) {
                // every symbol consumes at least one sample when omega cannot fall below 1, and a call never has more
                // than a ring of them staged: with that much room left in every lane's output the loop needs no output
                // test (reference clock_recovery_mm.c:103 `oo < output_len`, true throughout)
                if (__all(L.k.omega_mid - L.k.omega_lim >= 1.0f && off_end - off > 4u * ((uint32_t) G::ring + 8u))) {
                    k3_drain_finite<false, G::plain>(L, lim, col_addr, bank_addr, off, off_end, wg_out, (uint32_t) (G::ring - 1));
                } else {
                    k3_drain_finite<true, G::plain>(L, lim, col_addr, bank_addr, off, off_end, wg_out, (uint32_t) (G::ring - 1));
                }
                L.oo = (off - off_base) / (uint32_t) sizeof(float);
            }
        } else {
            K3_DRAIN(false)
        }
        if (active && !absent && !wild && !finished && (k + 1) * G::block >= L.nz) {
            finish_lane();  // this lane's samples are all staged and drained
        }
        if (b.k3_stamps) {
            unsigned long long t2 = __builtin_amdgcn_s_memtime();
            t_wait += t1 - t0;
            t_drain += t2 - t1;
            uint32_t most = 0;  // loop iterations of this block = the most symbols any lane produced in it
            for (int r = 0; r < G::lanes; r++) {
                const uint32_t v = (uint32_t) __builtin_amdgcn_readlane((int) (L.oo - oo0), r);
                most = v > most ? v : most;
            }
            n_iter += most;
        }
    }
    if (b.k3_stamps && lane == 0) {  // diagnostic only: cycles waiting for the producer vs in the symbol loops
        b.k3_stamps[blockIdx.x * 4 + 0] = (t_wait & 0xffffffffull) | ((real0 & 0xffffffffull) << 32);  // + when this wave's loop began (100 MHz ticks)
        b.k3_stamps[blockIdx.x * 4 + 1] = t_drain;
        // steps in the low half, 100 MHz ticks of the whole loop in the high half (shader clock = cycles / time)
        b.k3_stamps[blockIdx.x * 4 + 2] = (unsigned long long) nblocks | ((__builtin_amdgcn_s_memrealtime() - real0) << 32);
        // iterations, and in the upper bits the HW_ID of the consumer wave (bits 47:32) and of the producer wave (63:48)
        b.k3_stamps[blockIdx.x * 4 + 3] = (n_iter & 0xffffffffull) | ((unsigned long long) (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xffffu) << 32) |
                                          ((unsigned long long) (hw_sh & 0xffffu) << 48);
    }
#undef K3_DRAIN
    if (K3_FUSED_INT8(G)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every float soft bit is in memory
        safe_sh[lane] = (int) L.oo;
        __syncthreads();  // the staging wave converts what is left
    }
    if (absent) {
        b.out_len[c] = 0;
    } else if (active && wild) {
        const sdrm_chan_params p = b.params[c];
        const float *src = (p.dc_len ? b.dcout : b.z) + (size_t) c * b.z_stride;
        if (HAND) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the samples were written while this kernel ran: nothing stale from this XCD's L2
        }
        b.out_len[c] = sdrm_k3_rescue(p, cs, src, L.nz, (const float *) bank_rev, of, b.out_i8 + (size_t) c * b.out_stride, flagged);
        b.nonfinite[c] = 0;
        atomicAdd(b.counters + 0, 1u);  // sdrm_batch_wild_calls
    } else if (active && !finished) {
        finish_lane();
    }
    if (HAND && active && __hip_atomic_load(b.counters + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        b.out_len[c] = SDRM_OUT_LEN_FAILED;  // a hand-off wait of this batch ran into its bound: no result of this call can be trusted
    }
    if (b.k3_done != nullptr && lane == 0) {
        // the companion grid (k3_company) leaves when every workgroup of this launch has counted; nobody reads this launch's
        // results on the strength of the count (the next call's clock stage is ordered by the stream), so no fence
        __hip_atomic_fetch_add(b.k3_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    tl_mark(b, 2, 1);
}

// int8 soft bits from the float ones (reference src/dsp/fsk_demod.c:106), pointwise behind the clock stage, for the
// workgroup shapes whose staging wave has no time for it (K3_FUSED_INT8).
// grid (ceil(max_symbols / 1024), channels), 256 threads, four symbols per thread
// The grid covers the most symbols a channel in lock can have produced (sdrm_batch.hip, symbols_bound); a channel far out of
// lock can have more: the row's LAST workgroup walks on to the real count.
__global__ __launch_bounds__(256) void k3_quantize(DeviceBatch b) {
    const int c = blockIdx.y;
    const uint32_t n = b.out_len[c];
    const bool last = blockIdx.x + 1 == gridDim.x;
    for (uint32_t j = (blockIdx.x * 256 + threadIdx.x) * 4; j < n; j += 1024) {
        const float4 v = *reinterpret_cast<const float4 *>(b.out_f32 + (size_t) c * b.out_stride + j);
        int8_t *dst = b.out_i8 + (size_t) c * b.out_stride + j;
        if (j + 4 <= n) {
            char4 q;
            q.x = sdrm_soft_to_i8(v.x);
            q.y = sdrm_soft_to_i8(v.y);
            q.z = sdrm_soft_to_i8(v.z);
            q.w = sdrm_soft_to_i8(v.w);
            *reinterpret_cast<char4 *>(dst) = q;
        } else {
            const float t[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t i = 0; j + i < n; i++) {
                dst[i] = sdrm_soft_to_i8(t[i]);
            }
        }
        if (!last) {
            break;
        }
    }
}

template <int LANES, int RING, bool PLAIN>
static KernelLaunch describe_clock_as(const DeviceBatch &b) {
    KernelLaunch k;
    static lds_grant granted;
    k.lds = k3_lds_bytes(LANES, RING, PLAIN);
    allow_lds(k3_clock<LANES, RING, PLAIN>, k.lds, &granted);
    k.func = reinterpret_cast<const void *>(k3_clock<LANES, RING, PLAIN>);
    k.grid = dim3((unsigned) ((b.n_channels + LANES - 1) / LANES));
    k.block = dim3(128);
    return k;
}

sdrm_k3_shape k3_shape(const DeviceBatch &b) {
    int lanes = 0, ring = 0, plain = 0;
    sdrm_k3_parse_shape(getenv("SDRM_K3_LANES"), &lanes, &ring, &plain);  // tests and measurements: force one workgroup shape (read per launch)
    if (lanes == 0 && b.k3_lanes != 0) {  // the shape the batch measured to be its fastest (sdrm_tune.hip, sdrm_calibrate)
        lanes = b.k3_lanes;
        ring = b.k3_ring;
        plain = b.k3_plain;
    }
    return sdrm_k3_shape_for(b.n_channels, lanes, ring, plain, b.k3_carried_max);
}

sdrm_k3_shape describe_shape(const DeviceBatch &b) { return k3_shape(b); }

static KernelLaunch describe_clock_hand(const DeviceBatch &b) {
    KernelLaunch k = describe_clock_as<16, 1024, false>(b);
    static lds_grant granted;
    allow_lds(k3_clock<16, 1024, false, true>, k.lds, &granted);
    k.func = reinterpret_cast<const void *>(k3_clock<16, 1024, false, true>);
    return k;
}
static KernelLaunch describe_clock_hand32(const DeviceBatch &b) {
    KernelLaunch k = describe_clock_as<32, 512, false>(b);
    static lds_grant granted;
    allow_lds(k3_clock<32, 512, false, true>, k.lds, &granted);
    k.func = reinterpret_cast<const void *>(k3_clock<32, 512, false, true>);
    return k;
}
// the in-call hand-off exists for these two clock-stage shapes (batches of up to 2560 channels; the admission rule of
// sdrm_call.hip, at most 192 waiting workgroups, ends at 2048): blocking 1536 x 131072 call 7.10 -> 4.10 ms, 2048: 7.83 -> 6.23
bool clock_shape_hands_off(const DeviceBatch &b) {
    const sdrm_k3_shape sh = k3_shape(b);
    return !sh.plain && ((sh.lanes == 16 && sh.ring == 1024) || (sh.lanes == 32 && sh.ring == 512));
}

KernelLaunch describe_clock(const DeviceBatch &b) {
    const sdrm_k3_shape sh = k3_shape(b);
    switch ((sh.lanes * 10000 + sh.ring) * (sh.plain ? -1 : 1)) {
        case 16 * 10000 + 1024: return b.handoff ? describe_clock_hand(b) : describe_clock_as<16, 1024, false>(b);
        case 16 * 10000 + 512: return describe_clock_as<16, 512, false>(b);
        case 16 * 10000 + 256: return describe_clock_as<16, 256, false>(b);
        case 32 * 10000 + 512: return b.handoff ? describe_clock_hand32(b) : describe_clock_as<32, 512, false>(b);
        case 32 * 10000 + 256: return describe_clock_as<32, 256, false>(b);
        case -(64 * 10000 + 256): return describe_clock_as<64, 256, true>(b);
        case -(32 * 10000 + 256): return describe_clock_as<32, 256, true>(b);
        default: return describe_clock_as<64, 256, false>(b);
    }
}

KernelLaunch describe_quantize(const DeviceBatch &b) {
    KernelLaunch k;
    const sdrm_k3_shape sh = k3_shape(b);
    if (b.max_symbols == 0 || K3_FUSED_INT8_RING(sh.ring)) {  // the clock stage's staging wave has done it
        return k;
    }
    k.func = reinterpret_cast<const void *>(k3_quantize);
    k.grid = dim3((b.max_symbols + 1023) / 1024, (unsigned) b.n_channels);
    k.block = dim3(256);
    return k;
}

// Company for the clock stage.  A wave that is (nearly) alone on the chip runs a dependent chain slower than the same
// wave on a busy chip: with 256 channels the clock stage is 16 consumer waves, and some of them -- different ones from
// run to run -- took 245-260 cycles per symbol instead of 220, until other waves were issuing VECTOR instructions all over
// the chip.  Measured (tools/k3_ab.py, ms per 131072-sample chunk of 256 channels): 2.96-2.98 alone; beside 256 / 1024 /
// 2048 / 4096 one-wave workgroups that execute one v_mov per 32 s_nop 7: 2.86 / 2.73 / 2.64 / 2.58, every consumer wave
// then at 218-220 cycles per symbol; one v_mov per s_nop: 2.53 (the front-end beside it 0.70 instead of 0.55 ms, which
// matters only where it is the longer stage).  Waves that run nothing but s_nop, or s_sleep, change nothing; a solid
// stream of v_add takes the consumers' issue slots and helps nothing either; extra waves inside the clock stage's own
// workgroups (its CUs hold nothing else: its rings take the LDS) do not help: what counts is sparse vector activity on every
// compute unit -- presumably what keeps the chip's power management from parking parts of it.  Round 1's DC blocker, one
// busy workgroup per channel on every CU, was that company without anybody knowing; the lane-dense one is not.
// So small batches get a companion grid on a side stream beside each clock-stage launch: no LDS, no memory traffic but a
// look at the counter the clock stage's workgroups bump when they finish, every ~50 us, and a bound on its life.
// NOPS: s_nop 7 between two vector instructions of a companion wave (1 / 4 / 16 / 64): how sparse the company is.  The look
// at the counter comes every 48 x 64 s_nop 7 whatever NOPS is (~50 us: thousands of waves looking, keep it rare).
template <int NOPS>
__global__ __launch_bounds__(64) void k3_company(const uint32_t *done, uint32_t target, int max_rounds) {
    float a = threadIdx.x;
    for (int i = 0; i < max_rounds; i++) {
        for (int j = 0; j < 48; j++) {
#pragma unroll
            for (int r = 0; r < 64 / NOPS; r++) {  // unrolled: the instruction stream of rounds 2-3 (an assembler .rept) for NOPS = 1
                asm volatile("v_mov_b32 %0, %0" : "+v"(a));
#pragma unroll
                for (int n = 0; n < NOPS; n++) {
                    asm volatile("s_nop 7");
                }
            }
        }
        if ((int32_t) (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) {
            break;
        }
    }
}

void launch_clock_company(const DeviceBatch &b, uint32_t target, int blocks, int max_rounds, int nops, hipStream_t s) {
    // max_rounds x ~50 us bounds the grid's life whatever happens to the counter
    if (nops >= 64) {
        hipLaunchKernelGGL(k3_company<64>, dim3((unsigned) blocks), dim3(64), 0, s, b.k3_done, target, max_rounds);
    } else if (nops >= 16) {
        hipLaunchKernelGGL(k3_company<16>, dim3((unsigned) blocks), dim3(64), 0, s, b.k3_done, target, max_rounds);
    } else if (nops >= 4) {
        hipLaunchKernelGGL(k3_company<4>, dim3((unsigned) blocks), dim3(64), 0, s, b.k3_done, target, max_rounds);
    } else {
        hipLaunchKernelGGL(k3_company<1>, dim3((unsigned) blocks), dim3(64), 0, s, b.k3_done, target, max_rounds);
    }
}

unsigned clock_workgroups(const DeviceBatch &b) { return describe_clock(b).grid.x; }
unsigned dc_workgroups(const DeviceBatch &b) { return b.any_dc ? describe_dc(b).grid.x : 0u; }

void launch_clock(const DeviceBatch &b, hipStream_t s) {
    void *args[] = {(void *) &b};
    launch_described(describe_clock(b), args, s);
    launch_described(describe_quantize(b), args, s);
}

// ================================================================================================ generic channels
// DC blocker and clock recovery of the channels the LDS-resident stages are not sized for (sdrm_kernels.h, "generic
// channels"): one workgroup of one wave per channel, state in global memory.  Launched behind k2_dc / k3_clock on the same
// streams, only when the batch has such channels.

// reference src/dsp/dc_blocker.c:56-64,105-119.  A block of 64 samples per step; a boxcar of a generic channel is longer than
// that, so the delayed samples of a block were all written before the block began: the terms u[n] - u[n - L] are pointwise,
// the running sum is the in-order chain (one addition per sample, through the lanes), the quotient the IEEE division.
__global__ __launch_bounds__(64) void k2_dc_generic(DeviceBatch b) {
    const int c = b.gen_list[blockIdx.x];
    const sdrm_chan_params p = b.params[c];
    const sdrm_chunk_ctl ctl = b.ctl[c];
    if (p.dc_len == 0 || ctl.absent != 0 || ctl.nz == 0) {
        return;
    }
    const sdrm_gen_layout g = sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim);
    // (global-address-space pointers, said so: through a pointer loaded from memory the compiler emits FLAT accesses, which count in
    // two counters and made it wait for every single access -- 12 round trips per step)
    typedef __attribute__((address_space(1))) float gfloat;
    gfloat *st = (gfloat *) b.gen_state[c];
    const int lane = threadIdx.x;
    float acc[4] = {st[0], st[1], st[2], st[3]};
    uint32_t pos = sdrm_bits(st[4]), xpos = sdrm_bits(st[5]);  // where u[n - L] / x[n - 2 (L - 1)] of the call's first sample sit
    const gfloat *z = (const gfloat *) (b.z + (size_t) c * b.z_stride);
    gfloat *out = (gfloat *) (b.dcout + (size_t) c * b.z_stride);
    const float len_f = p.dc_len_f;
    const int nz = (int) ctl.nz;
    // The four boxcars are four in-order chains, each fed by the one in front -- but block by block: stage s can work on block
    // k - s while stage 0 works on block k.  So a step runs the four stages on four consecutive blocks, and their chains --
    // y[n] = t[n] + y[n - 1] through the lanes: lane 0 adds the carried sum, then sixty-three v_add_f32 whose first operand comes from the lane below (DPP:
    // lane i takes lane i - 1's sum; a lane without a source lane keeps its own; after the k-th every lane up to k holds its final
    // value and recomputes the same value from then on) -- are interleaved in one instruction stream: four independent
    // additions per sample position, no wait states between them.  What a step reads from memory (the new block's samples, the
    // delayed samples u_s[n - L] and x[n - 2 (L - 1)]) depends on nothing it computes -- a generic channel's boxcar is thousands of
    // samples long, so everything a step reads was written long before -- and is fetched one step ahead.
    // Same operations on the same values in the same order per sample as the shuffle form this replaces (24 ms per 131072-sample
    // call, on the DC stream in front of everybody's clock stage; now ~1.5: profiles/r05_dc_generic.txt).
    gfloat *xring = st + g.off_x;
    const int nb = (nz + 63) / 64;
    uint32_t pos_s[4] = {pos, pos, pos, pos};  // ring position of the block stage s works on next
    float in_s[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // its input: the block's samples (stage 0) / the stage in front's output
    float del_s[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // x[n - 2 (L - 1)] of that block, travelling with it to the output
    auto ring_index = [&](uint32_t at, uint32_t len) {
        const uint32_t i = at + (uint32_t) lane;
        return i >= len ? i - len : i;
    };
    auto advance = [&](uint32_t at, int by, uint32_t len) {
        const uint32_t i = at + (uint32_t) by;
        return i >= len ? i - len : i;
    };
    auto block_valid = [&](int k) { return nz - 64 * k < 64 ? nz - 64 * k : 64; };
    // operands of step `step` (stage s works on block step - s), fetched K2G_AHEAD steps before they are used -- one wave has
    // nothing else to hide a memory round trip behind, and a step's arithmetic is a fraction of one
#define K2G_AHEAD 4
    float x_q[K2G_AHEAD], del_q[K2G_AHEAD], old_q[K2G_AHEAD][4];
    uint32_t f_pos[4] = {pos, pos, pos, pos}, f_xpos = xpos;  // where the fetches stand (ahead of pos_s / xpos)
    // ST (a compile-time flag): the step lies in the call's steady state -- all four stages at work on full blocks -- and every
    // condition below is constant: no branch, no predicate, and the compiler can count the loads in flight instead of waiting
    // for all of them at every join (which is what the general form costs: it serves the first three, the last few and ragged steps)
    auto fetch = [&](int step, auto set_c, auto steady_c) {
        constexpr int set = decltype(set_c)::value;
        constexpr bool ST = decltype(steady_c)::value;
        if (ST || step < nb) {
            const bool on = ST || lane < block_valid(step);
            x_q[set] = on ? z[64 * step + lane] : 0.0f;
            del_q[set] = on ? xring[ring_index(f_xpos, g.XL)] : 0.0f;
            f_xpos = advance(f_xpos, ST ? 64 : block_valid(step), g.XL);
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int k = step - s;
            if (ST || (k >= 0 && k < nb)) {
                old_q[set][s] = (ST || lane < block_valid(k)) ? st[g.off_ring + (size_t) s * g.L + ring_index(f_pos[s], g.L)] : 0.0f;
                f_pos[s] = advance(f_pos[s], ST ? 64 : block_valid(k), g.L);
            }
        }
    };
    auto do_step = [&](int step, auto set_c, auto steady_c) {
        constexpr int j = decltype(set_c)::value;
        constexpr bool ST = decltype(steady_c)::value;
        if (!ST && step >= nb + 3) {
            return;
        }
        const float old4[4] = {old_q[j][0], old_q[j][1], old_q[j][2], old_q[j][3]};
        if (ST || step < nb) {
            in_s[0] = x_q[j];
            del_s[0] = del_q[j];
            if (ST || lane < block_valid(step)) {
                xring[ring_index(xpos, g.XL)] = x_q[j];
            }
            xpos = advance(xpos, ST ? 64 : block_valid(step), g.XL);
        }
        float t4[4], y4[4];
        uint32_t next_pos[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int k = step - s;
            const bool active = ST || (k >= 0 && k < nb);
            const bool on = ST || (active && lane < block_valid(k));
            if (on) {
                st[g.off_ring + (size_t) s * g.L + ring_index(pos_s[s], g.L)] = in_s[s];
            }
            t4[s] = on ? sdrm_boxcar_term(in_s[s], old4[s]) : 0.0f;
            y4[s] = t4[s] + (lane == 0 ? acc[s] : 0.0f);
            next_pos[s] = active ? advance(pos_s[s], ST ? 64 : block_valid(k), g.L) : pos_s[s];
        }
        // (behind this step's stores in program order: they touch other addresses, thousands of samples apart)
        fetch(step + K2G_AHEAD, set_c, steady_c);
        // (the wave-wide shift wave_shr:1 costs ~20 cycles an instruction on this part; the shift inside a row of sixteen lanes runs at
        // the full rate.  So: fifteen row shifts make row 0 final, one row_bcast:15 hands its last sum to row 1 -- to the whole row:
        // lane 16 is then right, lanes 17 .. 31 are recomputed one after the other by the next fifteen shifts, as they would be
        // from any value -- and so on through the four rows; rows that are final recompute their own values.)
#define K2G_ROW_PHASE                                                                   \
    ".rept 15\n\t"                                                                      \
    "v_add_f32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    "v_add_f32_dpp %1, %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    "v_add_f32_dpp %2, %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    "v_add_f32_dpp %3, %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"                 \
    ".endr\n\t"
#define K2G_ROW_CARRY(MASK)                                                             \
    "v_add_f32_dpp %0, %0, %4 row_bcast:15 row_mask:" MASK " bank_mask:0xf\n\t"         \
    "v_add_f32_dpp %1, %1, %5 row_bcast:15 row_mask:" MASK " bank_mask:0xf\n\t"         \
    "v_add_f32_dpp %2, %2, %6 row_bcast:15 row_mask:" MASK " bank_mask:0xf\n\t"         \
    "v_add_f32_dpp %3, %3, %7 row_bcast:15 row_mask:" MASK " bank_mask:0xf\n\t"
        asm volatile("s_nop 1\n\t"  // (a DPP operand written by the instruction in front needs two wait states; the compiler cannot see into this)
                     K2G_ROW_PHASE K2G_ROW_CARRY("0x2") K2G_ROW_PHASE K2G_ROW_CARRY("0x4") K2G_ROW_PHASE K2G_ROW_CARRY("0x8") K2G_ROW_PHASE
                     : "+v"(y4[0]), "+v"(y4[1]), "+v"(y4[2]), "+v"(y4[3])
                     : "v"(t4[0]), "v"(t4[1]), "v"(t4[2]), "v"(t4[3]));
#undef K2G_ROW_PHASE
#undef K2G_ROW_CARRY
        float out_u[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int k = step - s;
            const bool active = ST || (k >= 0 && k < nb);
            if (active) {
                acc[s] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, y4[s]), (ST ? 64 : block_valid(k)) - 1));
            }
            out_u[s] = sdrm_boxcar_out(y4[s], len_f);
            pos_s[s] = next_pos[s];
        }
        if (ST || (step >= 3 && lane < block_valid(step - 3))) {
            out[64 * (step - 3) + lane] = del_s[3] - out_u[3];
        }
#pragma unroll
        for (int s = 3; s > 0; s--) {  // every block moves on to the next stage
            in_s[s] = out_u[s - 1];
            del_s[s] = del_s[s - 1];
        }
    };
    typedef std::integral_constant<int, 0> set0;
    typedef std::integral_constant<int, 1> set1;
    typedef std::integral_constant<int, 2> set2;
    typedef std::integral_constant<int, 3> set3;
#pragma unroll
    for (int j = 0; j < K2G_AHEAD; j++) {
        x_q[j] = del_q[j] = 0.0f;
        old_q[j][0] = old_q[j][1] = old_q[j][2] = old_q[j][3] = 0.0f;
    }
    fetch(0, set0{}, std::false_type{});
    fetch(1, set1{}, std::false_type{});
    fetch(2, set2{}, std::false_type{});
    fetch(3, set3{}, std::false_type{});
    // a group of four steps is steady when its first step has all four stages at work (step >= 3) and the fetch its last step
    // issues (four steps ahead) still lands on full blocks only
    const int nb_full = nz / 64;
    for (int base = 0; base < nb + 3; base += K2G_AHEAD) {
        if (base >= K2G_AHEAD && base + 2 * K2G_AHEAD <= nb_full) {
            do_step(base + 0, set0{}, std::true_type{});
            do_step(base + 1, set1{}, std::true_type{});
            do_step(base + 2, set2{}, std::true_type{});
            do_step(base + 3, set3{}, std::true_type{});
        } else {
            do_step(base + 0, set0{}, std::false_type{});
            do_step(base + 1, set1{}, std::false_type{});
            do_step(base + 2, set2{}, std::false_type{});
            do_step(base + 3, set3{}, std::false_type{});
        }
    }
#undef K2G_AHEAD
    pos = pos_s[3];
    __syncthreads();  // every lane's ring stores are issued before the state words say so
    if (lane == 0) {
        st[0] = acc[0];
        st[1] = acc[1];
        st[2] = acc[2];
        st[3] = acc[3];
        st[4] = sdrm_from_bits(pos);
        st[5] = sdrm_from_bits(xpos);
    }
}

// reference src/dsp/clock_recovery_mm.c:78-139 (+ fsk_demod.c:106): the call's samples go behind the carried ones in the
// channel's working buffer (all lanes), one lane runs the symbol loop in its NaN-aware form (a few hundred symbols), all
// lanes move what is carried to the front.
__global__ __launch_bounds__(64) void k3_clock_generic(DeviceBatch b) {
    const int c = b.gen_list[blockIdx.x];
    const sdrm_chan_params p = b.params[c];
    const sdrm_chunk_ctl ctl = b.ctl[c];
    __shared__ float bank_rev[129 * SDRM_K3_BANKPITCH];
    __shared__ int from_sh, keep_sh;
    if (ctl.absent != 0) {
        if (threadIdx.x == 0) {
            b.out_len[c] = 0;
        }
        return;
    }
    const int lane = threadIdx.x;
    for (int k = lane; k < 129 * 8; k += 64) {
        bank_rev[(k >> 3) * SDRM_K3_BANKPITCH + (k & 7)] = b.mmse_bank[(k & ~7) + 7 - (k & 7)];
    }
    const sdrm_gen_layout g = sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim);
    // (global-address-space pointers, said so: see k2_dc_generic)
    typedef __attribute__((address_space(1))) float gfloat;
    gfloat *work = (gfloat *) (b.gen_state[c] + g.off_work);  // work[3 + i] = position i of the reference's working buffer
    sdrm_clock_state *cs = b.clock_state + c;
    const int kept = (int) cs->kept;
    const int nz = (int) ctl.nz;
    const gfloat *src = (const gfloat *) ((p.dc_len ? b.dcout : b.z) + (size_t) c * b.z_stride);
    int i_copy = lane;
    for (; i_copy + 192 < nz; i_copy += 256) {  // four loads in flight
        const float v0 = src[i_copy], v1 = src[i_copy + 64], v2 = src[i_copy + 128], v3 = src[i_copy + 192];
        work[3 + kept + i_copy] = v0;
        work[3 + kept + i_copy + 64] = v1;
        work[3 + kept + i_copy + 128] = v2;
        work[3 + kept + i_copy + 192] = v3;
    }
    for (; i_copy < nz; i_copy += 64) {
        work[3 + kept + i_copy] = src[i_copy];
    }
    __syncthreads();
    if (lane == 0) {
        sdrm_k3_lane L;
        L.k.omega_mid = p.omega_mid;
        L.k.omega_lim = p.omega_lim;
        L.k.gain_omega = p.gain_omega;
        L.k.gain_mu = p.gain_mu;
        L.kept = 0;
        L.nz = kept + nz;  // positions count from the start of the carried samples
        L.oo = 0;
        L.cap = p.max_len;
        L.st.mu = cs->mu;
        L.st.omega = cs->omega;
        L.st.last = cs->last;
        L.st.ii = 0;
        L.st.inc = 0;
        const uint32_t lim = sdrm_k3_limit(L, L.nz);
        float *of = b.out_f32 + (size_t) c * b.out_stride;
        int8_t *o8 = b.out_i8 + (size_t) c * b.out_stride;
        while (sdrm_k3_can_step(L, lim)) {
            sdrm_k3_operands F;
            sdrm_k3_fetch<false, sdrm_k3_geom_linear>(L, (const float *) work, (const float *) bank_rev, F);
            const float soft = sdrm_k3_step<false>(L, F);
            of[L.oo] = soft;
            o8[L.oo] = sdrm_soft_to_i8(soft);
            L.oo++;
        }
        int from, keep;
        sdrm_k3_finish_linear(L, g.hcap, &from, &keep);
        from_sh = from;
        keep_sh = keep;
        cs->mu = L.st.mu;
        cs->omega = L.st.omega;
        cs->last = L.st.last;
        cs->kept = (uint32_t) keep;
        cs->poison = 0;
        b.nonfinite[c] = 0;
        b.out_len[c] = L.oo;
    }
    __syncthreads();
    const int from = from_sh, keep = keep_sh;
    if (from > 0) {
        for (int i0 = 0; i0 < keep; i0 += 64) {  // forward move, a block at a time: every block is read before it is written
            const float v = i0 + lane < keep ? work[3 + from + i0 + lane] : 0.0f;
            __syncthreads();
            if (i0 + lane < keep) {
                work[3 + i0 + lane] = v;
            }
            __syncthreads();
        }
    }
}

void launch_dc_generic(const DeviceBatch &b, hipStream_t s) {
    if (b.n_gen > 0 && b.any_dc) {
        hipLaunchKernelGGL(k2_dc_generic, dim3((unsigned) b.n_gen), dim3(64), 0, s, b);
    }
}
void launch_clock_generic(const DeviceBatch &b, hipStream_t s) {
    if (b.n_gen > 0) {
        hipLaunchKernelGGL(k3_clock_generic, dim3((unsigned) b.n_gen), dim3(64), 0, s, b);
    }
}

// ================================================================================================ probes

__global__ void probe_atan2(const float *y, const float *x, const float *tab, float *out, size_t n) {
    __shared__ float t[260];
    for (int k = threadIdx.x; k < 257; k += blockDim.x) {
        t[k] = tab[k];
    }
    __syncthreads();
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = sdrm_fast_atan2f_flat(y[i], x[i], t);  // the form the front-end kernel runs
    }
}

// the quotient form of the DC kernel, one value per lane (16 per lane there: the fall-back is taken per wave either way)
__global__ void probe_boxcar_div(const float *sums, float len_f, float inv_len, float *out, size_t n) {
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    const float a = i < n ? sums[i] : 0.0f;
    bool unsafe;
    float v = sdrm_boxcar_out_fast(a, len_f, inv_len, &unsafe);
    if (__any(unsafe)) {
        v = sdrm_boxcar_out(a, len_f);
    }
    if (i < n) {
        out[i] = v;
    }
}

// the front-end's discriminator phase on a stream of LPF1 outputs: thread t takes samples [15 t, 15 t + 15) with sample
// 15 t - 1 as its predecessor (zero in front of the stream), the work of sdrm_k1_phase_quad; fast[wave] = the wave took
// the short form (sdrm_quad_block_fast) rather than the flat fall-back
__global__ __launch_bounds__(256) void probe_quad(const sdrm_f2 *y, size_t n, float gain, const float *atan_tab, float *out,
                                                  uint32_t *fast) {
    __shared__ __attribute__((aligned(16))) float tab[260];
    __shared__ __attribute__((aligned(16))) float tab2[512];
    const int tid = threadIdx.x;
    for (int k = tid; k < 257; k += 256) {
        tab[k] = atan_tab[k];
    }
    tab2[2 * tid] = atan_tab[tid];
    tab2[2 * tid + 1] = atan_tab[tid + 1] - atan_tab[tid];
    __syncthreads();
    const size_t base = ((size_t) blockIdx.x * 256 + tid) * SDRM_K1_R;
    sdrm_k1_regs regs;
#pragma unroll
    for (int r = 0; r < SDRM_K1_R; r++) {
        regs.y[r].x = regs.y[r].y = 0.0f;
        if (base + r < n) {
            regs.y[r] = y[base + r];
        }
    }
    sdrm_f2 prev;
    prev.x = prev.y = 0.0f;
    if (base > 0 && base - 1 < n) {
        prev = y[base - 1];
    }
    float q[SDRM_K1_R];
    bool ok = false;
#if defined(__HIP_DEVICE_COMPILE__)
    sdrm_lds_cf tab2_l = (sdrm_lds_cf) tab2;
    asm volatile("" : "+s"(tab2_l));
    ok = __all(sdrm_quad_block_fast<SDRM_K1_R>(regs.y, prev, gain, tab2_l, q));
#endif
    if (!ok) {
        sdrm_f2 pv = prev;
#pragma unroll
        for (int r = 0; r < SDRM_K1_R; r++) {
            q[r] = sdrm_quad_sample_flat(regs.y[r], pv, gain, tab);
            pv = regs.y[r];
        }
    }
#pragma unroll
    for (int r = 0; r < SDRM_K1_R; r++) {
        if (base + r < n) {
            out[base + r] = q[r];
        }
    }
    if ((tid & 63) == 0) {
        fast[blockIdx.x * 4 + (tid >> 6)] = ok ? 1u : 0u;
    }
}

void launch_probe_quad(const sdrm_f2 *d_y, size_t n, float gain, const float *d_tab, float *d_out, uint32_t *d_fast, hipStream_t s) {
    const unsigned blocks = (unsigned) ((n + 256 * SDRM_K1_R - 1) / (256 * SDRM_K1_R));
    hipLaunchKernelGGL(probe_quad, dim3(blocks ? blocks : 1), dim3(256), 0, s, d_y, n, gain, d_tab, d_out, d_fast);
}

void launch_probe_boxcar_div(const float *d_sums, uint32_t length, float *d_out, size_t n, hipStream_t s) {
    unsigned blocks = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(probe_boxcar_div, dim3(blocks ? blocks : 1), dim3(256), 0, s, d_sums, (float) length, 1.0f / (float) length, d_out, n);
}

void launch_probe_atan2(const float *d_y, const float *d_x, const float *d_tab, float *d_out, size_t n, hipStream_t s) {
    unsigned blocks = (unsigned) ((n + 255) / 256);
    hipLaunchKernelGGL(probe_atan2, dim3(blocks ? blocks : 1), dim3(256), 0, s, d_y, d_x, d_tab, d_out, n);
}

}  // namespace sdrm
