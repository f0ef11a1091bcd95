// sdrm_plan.cpp -- see sdrm_plan.h
#include "sdrm_plan.h"

#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

namespace sdrm {

static uint32_t round_up_u32(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// LDS of the DC kernel: term rows and checkpoints of 4 stages x 16 slots, the delayed-input tile, three delay rings per channel of the group,
// the slots' constants
static size_t dc_lds_bytes_for(uint32_t l_cap, uint32_t group) {
    const uint32_t rcap = (l_cap + SDRM_K2_BLK - 1) / SDRM_K2_BLK * SDRM_K2_BLK + SDRM_K2_BLK;
    return ((size_t) SDRM_K2_ROWS * SDRM_K2_TSPITCH + (size_t) SDRM_K2_ROWS * SDRM_K2_NBUF * SDRM_K2_LPS +
            (size_t) SDRM_K2_SLOTS * 2 * SDRM_K2_BLK + 3 * (size_t) group * sdrm_k2_ring_pitch(rcap)) * sizeof(float) + SDRM_K2_SLOTS * sizeof(sdrm_k2_slot) + 64;
}

static uint32_t dc_group_for(uint32_t l_cap) {
    uint32_t group = SDRM_K2_SLOTS;
    while (group > 1 && dc_lds_bytes_for(l_cap, group) > 150 * 1024) {
        group /= 2;
    }
    return group;
}

size_t BatchPlan::dc_lds_bytes() const { return dc_lds_bytes_for(dc_l_cap, dc_group); }

// everything of a channel's device parameters that follows from its design alone (no offsets into shared storage)
static int params_from_design(const ChannelDesign &d, sdrm_chan_params &p) {
    memset(&p, 0, sizeof(p));
    p.T1 = (uint32_t) d.taps1.size();
    p.T2 = (uint32_t) d.taps2.size();
    p.decim = d.cfg.decimation;
    p.dc_len = d.dc_length;
    p.hist_len = p.T1 + p.T2 - 1;
    if (p.T2 + 2 > (uint32_t) SDRM_K1_NY || sdrm_k1_lds_bytes_for(p.T1, p.T2) > 160 * 1024) {
        // the front-end stages a tile and its T1 - 1 samples of halo, and both filters' taps, in a CU's LDS
        fprintf(stderr, "<3>low-pass filters of %u / %u taps do not fit a tile\n", p.T1, p.T2);
        return -ENOTSUP;
    }
    // a tile computes SDRM_K1_NY LPF1 positions: (m-1)*d + T2 + 1 of them are needed for m outputs
    uint32_t by_halo = (uint32_t) ((SDRM_K1_NY - 1 - (int) p.T2) / (int) p.decim + 1);
    p.tile_m = std::min<uint32_t>((uint32_t) (SDRM_K1_THREADS * SDRM_K1_RZ), by_halo);
    p.max_len = d.cfg.max_input_buffer_length;
    p.quad_gain = d.quad_gain;
    p.omega_mid = d.sps;
    p.omega_lim = d.omega_lim;
    p.gain_omega = d.gain_omega;
    p.gain_mu = d.gain_mu;
    if (p.dc_len) {
        p.dc_len_f = (float) p.dc_len;
        p.dc_inv_len = 1.0f / p.dc_len_f;  // correctly rounded reciprocal: sdrm_boxcar_out_fast
    }
    p.generic = d.generic ? 1u : 0u;
    p.amp_safe = sdrm_amp_safe(p.omega_mid, p.omega_lim, p.gain_mu);
    // Can the clock stage's input ever exceed it?  |discriminator| <= |gain| * pi (fast_atan2f.c:87-157 returns within a
    // table step of [-pi, pi]), LPF2 scales by at most sum |tap|, the DC blocker subtracts a running mean of the same stream
    // (|x - mean| <= 2 max |x|; 5 % on top for what its running sums drift over a long stream).  Only host-side bounds hang
    // on this (how many symbols a call can produce at most); the kernels compare every sample they write.
    double abs_sum = 0.0;
    for (float t : d.taps2) {
        abs_sum += fabs((double) t);
    }
    const double reach = fabs((double) d.quad_gain) * 3.1416 * 1.001 * abs_sum * (p.dc_len ? 2.05 : 1.0) * 1.001;
    p.can_wild = !(reach < (double) p.amp_safe) ? 1u : 0u;
    return 0;
}

static void append_taps(std::vector<float> &pool, const std::vector<float> &taps) {
    pool.insert(pool.end(), taps.rbegin(), taps.rend());  // reversed: fir_filter.c:25-28
    while (pool.size() % 8) {
        pool.push_back(0.0f);
    }
}

int plan_batch(const sdrm_fsk_config *cfgs, size_t n, BatchPlan &plan) {
    plan.design.resize(n);
    for (size_t c = 0; c < n; c++) {
        int code = design_channel(cfgs[c], plan.design[c]);
        if (code != 0) {
            return code;
        }
    }
    plan.params.resize(n);
    plan.phase.assign(n, 0);
    plan.parity.assign(n, 0);
    plan.zbase.assign(n, 0);
    uint32_t h_max = 0, maxlen_max = 0;
    size_t last_distinct = 0;
    for (size_t c = 0; c < n; c++) {
        const ChannelDesign &d = plan.design[c];
        sdrm_chan_params &p = plan.params[c];
        int code = params_from_design(d, p);
        if (code != 0) {
            return code;
        }
        // channels with the same filters share one copy of the taps (batches are mostly a few distinct configs)
        bool shared = false;
        const size_t probes[2] = {last_distinct, c ? c - 1 : 0};
        for (size_t e : probes) {
            if (c > 0 && e < c && plan.design[e].taps1 == d.taps1 && plan.design[e].taps2 == d.taps2) {
                p.taps1_off = plan.params[e].taps1_off;
                p.taps2_off = plan.params[e].taps2_off;
                shared = true;
                break;
            }
        }
        if (!shared) {
            last_distinct = c;
            p.taps1_off = (uint32_t) plan.tap_pool.size();
            append_taps(plan.tap_pool, d.taps1);
            p.taps2_off = (uint32_t) plan.tap_pool.size();
            append_taps(plan.tap_pool, d.taps2);
        }
        // generic channels (sdrm_kernels.h) keep their DC and clock state in global memory: they size neither the DC stage's
        // LDS rings nor the clock stage's; their DC output still goes through the batch's dcout buffers
        if (p.dc_len) {
            plan.any_dc = 1;
            if (!p.generic) {
                plan.dc_l_cap = std::max(plan.dc_l_cap, p.dc_len);
            }
        }
        if (!p.generic) {
            plan.clock_carried_max = std::max(plan.clock_carried_max, (uint32_t) (d.sps * 1.01f + 8.0f));
        }
        plan.t1_max = std::max(plan.t1_max, p.T1);
        plan.t2_max = std::max(plan.t2_max, p.T2);
        h_max = std::max(h_max, p.hist_len);
        maxlen_max = std::max(maxlen_max, p.max_len);
    }
    // every channel gets a DC state region of the batch's largest size, so that a channel can later be given any
    // configuration the batch's geometry holds (replan_channel)
    if (plan.any_dc && plan.dc_l_cap == 0) {
        plan.dc_l_cap = 32;  // every DC blocker of the batch is a generic one: the fast stage keeps a minimal geometry
    }
    plan.dc_hx_cap = plan.any_dc ? 2 * (plan.dc_l_cap - 1) : 0;
    plan.dc_region_floats = plan.any_dc ? sdrm_k2_state_floats(plan.dc_hx_cap, plan.dc_l_cap) : 0;
    // sixteen channels per DC workgroup while their delay rings fit beside the term rows (150 KB of the CU's 160)
    plan.dc_group = dc_group_for(plan.dc_l_cap);
    if (plan.any_dc && dc_lds_bytes_for(plan.dc_l_cap, plan.dc_group) > 160 * 1024) {
        fprintf(stderr, "<3>DC blocker of %u samples does not fit the device\n", plan.dc_l_cap);
        return -ENOTSUP;
    }
    for (size_t c = 0; c < n; c++) {
        plan.params[c].dc_state_off = (uint32_t) (c * plan.dc_region_floats);
    }
    plan.dc_state_floats = n * plan.dc_region_floats;
    // behind the shared taps, one private slot per channel for configurations assigned later
    plan.private_taps_base = plan.tap_pool.size();
    plan.private_taps_slot = (size_t) round_up_u32(plan.t1_max, 8) + round_up_u32(plan.t2_max, 8);
    plan.hist_stride = round_up_u32(h_max, 8);
    plan.z_stride = round_up_u32(maxlen_max + 64, 64);
    plan.out_stride = round_up_u32(maxlen_max + 64, 64);
    plan.in_stride = round_up_u32(std::max<uint32_t>(maxlen_max, 1), 64);
    return 0;
}

int plan_growth(const BatchPlan &plan, const sdrm_fsk_config &cfg, GeometryGrowth &g) {
    ChannelDesign d;
    int code = design_channel(cfg, d);
    if (code != 0) {
        return code;
    }
    sdrm_chan_params p;
    code = params_from_design(d, p);
    if (code != 0) {
        return code;
    }
    if (p.max_len > plan.in_stride || p.max_len + 64 > plan.z_stride) {
        fprintf(stderr, "<3>configuration does not fit the batch it is assigned to (buffer of %u samples, the batch holds %u)\n",
                p.max_len, plan.in_stride);
        return -ENOTSUP;
    }
    g.t1_max = std::max(plan.t1_max, p.T1);
    g.t2_max = std::max(plan.t2_max, p.T2);
    g.hist_stride = std::max(plan.hist_stride, round_up_u32(p.hist_len, 8));
    g.dc_l_cap = std::max(plan.dc_l_cap, p.generic ? 0u : p.dc_len);  // a generic channel's boxcar lives in global memory
    g.any_dc = plan.any_dc || p.dc_len != 0;
    if (g.any_dc && g.dc_l_cap == 0) {
        g.dc_l_cap = 32;
    }
    g.needed = g.t1_max != plan.t1_max || g.t2_max != plan.t2_max || g.hist_stride != plan.hist_stride ||
               g.dc_l_cap != plan.dc_l_cap || g.any_dc != plan.any_dc;
    if (g.any_dc && dc_lds_bytes_for(g.dc_l_cap, dc_group_for(g.dc_l_cap)) > 160 * 1024) {
        fprintf(stderr, "<3>DC blocker of %u samples does not fit the device\n", g.dc_l_cap);
        return -ENOTSUP;
    }
    return 0;
}

void apply_growth(BatchPlan &plan, const GeometryGrowth &g, std::vector<size_t> &moved) {
    const size_t n = plan.params.size();
    plan.t1_max = g.t1_max;
    plan.t2_max = g.t2_max;
    plan.hist_stride = g.hist_stride;
    plan.any_dc = g.any_dc;
    plan.dc_l_cap = g.dc_l_cap;
    plan.dc_hx_cap = plan.any_dc ? 2 * (plan.dc_l_cap - 1) : 0;
    plan.dc_region_floats = plan.any_dc ? sdrm_k2_state_floats(plan.dc_hx_cap, plan.dc_l_cap) : 0;
    plan.dc_group = dc_group_for(plan.dc_l_cap);
    plan.dc_state_floats = n * plan.dc_region_floats;
    plan.private_taps_slot = (size_t) round_up_u32(plan.t1_max, 8) + round_up_u32(plan.t2_max, 8);
    moved.clear();
    for (size_t c = 0; c < n; c++) {
        sdrm_chan_params &p = plan.params[c];
        p.dc_state_off = (uint32_t) (c * plan.dc_region_floats);
        if (p.taps1_off >= plan.private_taps_base) {
            // every channel that lives in a private slot is listed, also when the slot size stays: the caller builds a new
            // tap buffer and must fill all of them again
            p.taps1_off = (uint32_t) (plan.private_taps_base + c * plan.private_taps_slot);
            p.taps2_off = p.taps1_off + round_up_u32(plan.t1_max, 8);
            moved.push_back(c);
        }
    }
}

int replan_channel(BatchPlan &plan, size_t c, const sdrm_fsk_config &cfg, std::vector<float> &taps_slot) {
    if (c >= plan.params.size()) {
        return -1;
    }
    ChannelDesign d;
    int code = design_channel(cfg, d);
    if (code != 0) {
        return code;
    }
    sdrm_chan_params p;
    code = params_from_design(d, p);
    if (code != 0) {
        return code;
    }
    // the batch's geometry (LDS sizes, strides, buffers) was fixed when it was created
    if (p.T1 > plan.t1_max || p.T2 > plan.t2_max || p.hist_len > plan.hist_stride || p.max_len > plan.in_stride ||
        p.max_len + 64 > plan.z_stride || (p.dc_len && (!plan.any_dc || (!p.generic && p.dc_len > plan.dc_l_cap)))) {
        fprintf(stderr, "<3>configuration does not fit the batch it is assigned to (filters of %u / %u taps, DC length %u, "
                        "buffer %u)\n", p.T1, p.T2, p.dc_len, p.max_len);
        return -ENOTSUP;
    }
    p.dc_state_off = (uint32_t) (c * plan.dc_region_floats);
    p.taps1_off = (uint32_t) (plan.private_taps_base + c * plan.private_taps_slot);
    p.taps2_off = p.taps1_off + round_up_u32(plan.t1_max, 8);
    taps_slot.assign(plan.private_taps_slot, 0.0f);
    std::copy(d.taps1.rbegin(), d.taps1.rend(), taps_slot.begin());
    std::copy(d.taps2.rbegin(), d.taps2.rend(), taps_slot.begin() + round_up_u32(plan.t1_max, 8));
    plan.design[c] = d;
    plan.params[c] = p;
    if (!p.generic) {
        plan.clock_carried_max = std::max(plan.clock_carried_max, (uint32_t) (d.sps * 1.01f + 8.0f));  // the clock stage follows with its ring
    }
    plan.phase[c] = 0;
    plan.parity[c] = 0;
    plan.zbase[c] = 0;
    return 0;
}

uint32_t plan_call(BatchPlan &plan, const size_t *lens, sdrm_chunk_ctl *ctl) {
    uint32_t max_tiles = 0;
    const size_t n_ch = plan.params.size();
    for (size_t c = 0; c < n_ch; c++) {
        const sdrm_chan_params &p = plan.params[c];
        size_t n = lens ? lens[c] : 0;
        const bool absent = n == SDRM_LEN_ABSENT;
        if (absent) {
            n = 0;
        }
        if (n > p.max_len) {
            fprintf(stderr, "<3>requested buffer %zu is more than max: %zu\n", n, (size_t) p.max_len);
            n = 0;
        }
        sdrm_chunk_ctl &k = ctl[c];
        k.n_in = (uint32_t) n;
        k.i0 = plan.phase[c];
        // LPF2 emits at stream positions that are multiples of d (fir_filter.c:100-107 carries the phase in
        // history_offset); i0 is where the next one falls inside this call's input
        k.nz = (k.n_in > k.i0) ? (k.n_in - k.i0 + p.decim - 1) / p.decim : 0;
        k.tiles = (k.nz + p.tile_m - 1) / p.tile_m;
        k.parity = plan.parity[c];
        k.zbase = plan.zbase[c];
        k.nco_off = 0;
        k.nco_cnt = 0;
        k.absent = absent ? 1u : 0u;
        k.pre = 0;
        plan.phase[c] = k.i0 + k.nz * p.decim - k.n_in;
        plan.parity[c] ^= 1u;
        plan.zbase[c] += k.nz;
        max_tiles = std::max(max_tiles, k.tiles);
    }
    return max_tiles;
}

int plan_nco(const BatchPlan &plan, const sdrm_nco_segment *segs, size_t n_segs, sdrm_chunk_ctl *ctl,
             std::vector<sdrm_nco_seg> &table) {
    table.clear();
    const size_t n_ch = plan.params.size();
    size_t i = 0;
    while (i < n_segs) {
        const uint32_t c = segs[i].channel;
        if (c >= n_ch || ctl[c].nco_cnt != 0) {
            fprintf(stderr, "<3>nco segments must be grouped by channel (segment %zu, channel %u)\n", i, c);
            return -1;
        }
        const uint64_t fs = plan.design[c].cfg.sampling_freq;
        uint64_t total = 0;
        ctl[c].nco_off = (uint32_t) table.size();
        while (i < n_segs && segs[i].channel == c) {
            sdrm_nco_seg s;
            s.len = segs[i].len;
            // reference src/dsp/sig_source.c:44: `M_2PI * (float) freq / source->rx_sampling_freq` in fp32
            const float two_pi = (float) (2 * 3.14159265358979323846);
            s.step = two_pi * (float) segs[i].freq_hz / fs;
            table.push_back(s);
            total += s.len;
            i++;
        }
        ctl[c].nco_cnt = (uint32_t) (table.size() - ctl[c].nco_off);
        if (total != ctl[c].n_in) {
            if (ctl[c].n_in == 0) {  // oversize / empty input was dropped: so is its correction
                table.resize(ctl[c].nco_off);
                ctl[c].nco_cnt = 0;
                continue;
            }
            fprintf(stderr, "<3>nco segments of channel %u cover %llu samples, input has %u\n", c,
                    (unsigned long long) total, ctl[c].n_in);
            return -1;
        }
    }
    return 0;
}

// reference src/dsp/doppler.c:128-180, one batch per loop turn
size_t DopplerPlanner::plan(uint32_t channel, size_t input_len, sdrm_nco_segment *out, size_t cap) {
    size_t done = 0, count = 0;
    while (done < input_len && count < cap) {
        const size_t remaining = input_len - done;
        size_t batch;
        if (interval < remaining + in_interval) {
            if (in_interval >= interval) {
                batch = interval < remaining ? (size_t) interval : remaining;
            } else {
                batch = (size_t) (interval - in_interval);
            }
        } else {
            batch = remaining;
        }
        if (in_interval >= interval) {
            in_interval = 0;
            cur = (next == 0) ? fn(user, second++) : next;  // :147-160 (0 doubles as "not evaluated yet")
            next = fn(user, second++);
            slope = (next - cur) / interval;                 // :166 linear interpolation inside the second
        } else {
            cur += slope * (double) batch;                   // :168
        }
        in_interval += batch;
        out[count].channel = channel;
        out[count].len = (uint32_t) batch;
        out[count].freq_hz = (int64_t) cur;                  // :180 truncation to integer Hz
        count++;
        done += batch;
    }
    return count;
}

}  // namespace sdrm
