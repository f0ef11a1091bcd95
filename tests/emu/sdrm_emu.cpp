// sdrm_emu.cpp -- TEST INFRASTRUCTURE: drives the per-thread kernel bodies of sdr-modem_amd/csrc/sdrm_kernels.h on
// the host, thread by thread and phase by phase (a __syncthreads() becomes the end of a loop over thread ids), with
// the same planning code the C-ABI uses (sdrm_plan.cpp).  It lets the CPU-only test suite check tiling, history
// hand-off, decimation phase, ring indexing and the clock loop's block-wise execution against the oracle without
// a GPU.  It is NOT a fallback: the product library never links or loads this file.
//
// K2 (DC blocker): the kernel's six roles are driven from the same per-lane bodies, iteration by iteration.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../sdr-modem_amd/csrc/sdrm_plan.h"
#include "../../sdr-modem_amd/csrc/sdrm_tables.h"

using namespace sdrm;

struct EmuBatch {
    BatchPlan plan;
    std::vector<sdrm_f2> hist;
    std::vector<float> z, dcout, dcstate;
    std::vector<sdrm_clock_state> clock;
    std::vector<int8_t> out8;
    std::vector<float> outf;
    std::vector<uint32_t> outlen;
    std::vector<sdrm_chunk_ctl> ctl;
    std::vector<uint32_t> nonfinite;
    uint64_t wild_calls = 0;
    std::vector<int64_t> pre_offset;  // the oscillator in front (sdrm_batch_set_pre_offset), Hz per channel, 0 = none
    std::vector<float> pre_state;
    std::vector<std::vector<sdrm_f2>> pre_mixed;
    std::vector<float> nco_state;
    std::vector<std::vector<sdrm_f2>> mixed;
    std::vector<sdrm_nco_seg> nco_table;
    std::vector<std::vector<float>> gen;  // generic channels (sdrm_kernels.h): the channel's state region, empty for the others
};

static void emu_sync_generic(EmuBatch *b, long only) {
    const BatchPlan &pl = b->plan;
    b->gen.resize(pl.params.size());
    for (size_t c = 0; c < pl.params.size(); c++) {
        if (only >= 0 && (size_t) only != c) continue;
        const sdrm_chan_params &p = pl.params[c];
        b->gen[c].clear();
        if (p.generic) b->gen[c].assign(sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim).total, 0.0f);
    }
}

extern "C" int emu_create(const sdrm_fsk_config *cfgs, size_t n, EmuBatch **out) {
    EmuBatch *b = new EmuBatch();
    int code = plan_batch(cfgs, n, b->plan);
    if (code != 0) {
        delete b;
        return code;
    }
    const BatchPlan &pl = b->plan;
    b->hist.assign(n * 2 * (size_t) pl.hist_stride, sdrm_f2{0.0f, 0.0f});
    b->z.assign(n * (size_t) pl.z_stride, 0.0f);
    b->dcout.assign(n * (size_t) pl.z_stride, 0.0f);
    b->dcstate.assign(pl.dc_state_floats + 8, 0.0f);
    b->plan.tap_pool.resize(pl.private_taps_base + n * pl.private_taps_slot + 16, 0.0f);  // room for reassigned channels
    b->clock.resize(n);
    for (size_t c = 0; c < n; c++) {
        memset(&b->clock[c], 0, sizeof(sdrm_clock_state));
        b->clock[c].mu = 0.5f;
        b->clock[c].omega = pl.design[c].sps;
    }
    b->out8.assign(n * (size_t) pl.out_stride, 0);
    b->outf.assign(n * (size_t) pl.out_stride, 0.0f);
    b->outlen.assign(n, 0);
    b->ctl.resize(n);
    b->nonfinite.assign(n, 0);
    b->nco_state.assign(n, 0.0f);
    b->mixed.resize(n);
    emu_sync_generic(b, -1);
    *out = b;
    return 0;
}

extern "C" void emu_destroy(EmuBatch *b) { delete b; }

// the emulation's counterpart of sdrm_batch_reset_channel (same planning code, state vectors instead of device memory)
extern "C" int emu_reset_channel(EmuBatch *b, size_t c, const sdrm_fsk_config *cfg) {
    BatchPlan &pl = b->plan;
    if (c >= pl.params.size()) return -1;
    const sdrm_fsk_config use = cfg ? *cfg : pl.design[c].cfg;
    GeometryGrowth growth;
    int code = plan_growth(pl, use, growth);
    if (code != 0) return code;
    if (growth.needed) {
        // the emulation's counterpart of grow_geometry (sdrm_batch.hip): the same planning calls, vectors instead of device memory
        const size_t n = pl.params.size();
        const uint32_t old_hist = pl.hist_stride, old_hx = pl.dc_hx_cap, old_l = pl.dc_l_cap;
        const size_t old_region = pl.dc_region_floats;
        const bool had_dc = pl.any_dc != 0;
        std::vector<size_t> moved;
        apply_growth(pl, growth, moved);
        std::vector<sdrm_f2> hist2(n * 2 * (size_t) pl.hist_stride, sdrm_f2{0.0f, 0.0f});
        for (size_t r = 0; r < n * 2; r++) std::copy(b->hist.begin() + r * old_hist, b->hist.begin() + (r + 1) * old_hist, hist2.begin() + r * pl.hist_stride);
        b->hist.swap(hist2);
        std::vector<float> dc2(pl.dc_state_floats + 8, 0.0f);
        if (had_dc) {
            for (size_t ch = 0; ch < n; ch++) {
                const float *src = b->dcstate.data() + ch * old_region;
                float *dst = dc2.data() + ch * pl.dc_region_floats;
                std::copy(src, src + old_hx, dst);
                for (int ring = 0; ring < 3; ring++) std::copy(src + old_hx + (size_t) ring * old_l, src + old_hx + (size_t) (ring + 1) * old_l, dst + pl.dc_hx_cap + (size_t) ring * pl.dc_l_cap);
                std::copy(src + old_hx + 3 * (size_t) old_l, src + old_hx + 3 * (size_t) old_l + 8, dst + pl.dc_hx_cap + 3 * (size_t) pl.dc_l_cap);
            }
        }
        b->dcstate.swap(dc2);
        pl.tap_pool.resize(pl.private_taps_base + n * pl.private_taps_slot + 16, 0.0f);
        for (size_t ch : moved) {
            const ChannelDesign &d = pl.design[ch];
            std::copy(d.taps1.rbegin(), d.taps1.rend(), pl.tap_pool.begin() + pl.params[ch].taps1_off);
            std::copy(d.taps2.rbegin(), d.taps2.rend(), pl.tap_pool.begin() + pl.params[ch].taps2_off);
        }
    }
    std::vector<float> slot;
    code = replan_channel(pl, c, use, slot);
    if (code != 0) return code;
    std::copy(slot.begin(), slot.end(), pl.tap_pool.begin() + pl.params[c].taps1_off);
    std::fill(b->hist.begin() + c * 2 * pl.hist_stride, b->hist.begin() + (c + 1) * 2 * pl.hist_stride, sdrm_f2{0.0f, 0.0f});
    if (pl.dc_region_floats)
        std::fill(b->dcstate.begin() + c * pl.dc_region_floats, b->dcstate.begin() + (c + 1) * pl.dc_region_floats, 0.0f);
    memset(&b->clock[c], 0, sizeof(sdrm_clock_state));
    b->clock[c].mu = 0.5f;
    b->clock[c].omega = pl.design[c].sps;
    b->nonfinite[c] = 0;
    b->nco_state[c] = 0.0f;
    if (c < b->pre_offset.size()) {
        b->pre_offset[c] = 0;
        b->pre_state[c] = 0.0f;
    }
    emu_sync_generic(b, (long) c);
    return 0;
}

static void emu_front(EmuBatch *b, const sdrm_f2 *const *inputs) {
    const BatchPlan &pl = b->plan;
    const size_t C = pl.params.size();
    std::vector<sdrm_f2> xs(SDRM_K1_NY + pl.t1_max);
    std::vector<float> qs(SDRM_K1_NY + SDRM_K1_QPAD), zs(SDRM_K1_NY);
    std::vector<sdrm_f2> bnd(SDRM_K1_THREADS);
    std::vector<float> tab(260);
    std::vector<sdrm_k1_regs> regs(SDRM_K1_THREADS);
    for (size_t c = 0; c < C; c++) {
        const sdrm_chan_params &p = pl.params[c];
        const sdrm_chunk_ctl &ctl = b->ctl[c];
        const sdrm_f2 *in = inputs[c];
        const sdrm_f2 *hist = b->hist.data() + (c * 2 + ctl.parity) * pl.hist_stride;
        for (uint32_t tile = 0; tile < ctl.tiles; tile++) {
            // poison the "LDS" so that any read of a slot the kernel did not write shows up as NaN in the outputs
            for (auto &v : xs) v = sdrm_f2{NAN, NAN};
            for (auto &v : qs) v = NAN;
            for (auto &v : zs) v = NAN;
            const sdrm_k1_tile t = sdrm_k1_tile_setup(p, ctl, (int) tile);
            for (int tid = 0; tid < SDRM_K1_THREADS; tid++)
                sdrm_k1_phase_load(tid, t, in, hist, (int) p.hist_len, sdrm_atan_tab, xs.data(), tab.data());
            for (int tid = 0; tid < SDRM_K1_THREADS; tid++)
                sdrm_k1_phase_lpf1(tid, t, p, pl.tap_pool.data() + p.taps1_off, xs.data(), bnd.data(), regs[tid]);
            for (int tid = 0; tid < SDRM_K1_THREADS; tid++)
                sdrm_k1_phase_quad(tid, t, p, tab.data(), nullptr, bnd.data(), regs[tid], qs.data() + 1);
            for (int tid = 0; tid < SDRM_K1_THREADS; tid++)
                sdrm_k1_phase_lpf2(tid, t, p, pl.tap_pool.data() + p.taps2_off, qs.data() + 1, zs.data(), &b->nonfinite[c]);
            for (int tid = 0; tid < SDRM_K1_THREADS; tid++)
                sdrm_k1_phase_store(tid, t, zs.data(), b->z.data() + c * pl.z_stride);
        }
        sdrm_f2 *next = b->hist.data() + (c * 2 + (ctl.parity ^ 1u)) * pl.hist_stride;
        for (int tid = 0; tid < 256; tid++) sdrm_hist_roll(tid, 256, p, ctl, in, hist, next);
    }
}

static void emu_front_any(EmuBatch *b, const sdrm_f2 *const *inputs) { emu_front(b, inputs); }

static void emu_dc(EmuBatch *b) {
    const BatchPlan &pl = b->plan;
    if (!pl.any_dc) return;
    const int C = (int) pl.params.size();
    const int G = (int) pl.dc_group;
    const uint32_t rcap_max = (pl.dc_l_cap + SDRM_K2_BLK - 1) / SDRM_K2_BLK * SDRM_K2_BLK + SDRM_K2_BLK;
    const uint32_t rpitch = sdrm_k2_ring_pitch(rcap_max);
    std::vector<float> ts(SDRM_K2_ROWS * SDRM_K2_TSPITCH), check(SDRM_K2_ROWS * SDRM_K2_NBUF * SDRM_K2_LPS), rings(3 * (size_t) G * rpitch);
    for (int c0 = 0; c0 < C; c0 += G) {
        std::fill(ts.begin(), ts.end(), 0.0f);
        std::fill(check.begin(), check.end(), NAN);
        std::fill(rings.begin(), rings.end(), NAN);
        sdrm_k2_slot slots[SDRM_K2_SLOTS];
        int nb = 0;
        for (int sl = 0; sl < SDRM_K2_SLOTS; sl++) {
            sdrm_k2_slot &s = slots[sl];
            s.chan = -1;
            s.L = 1; s.A = 64; s.rcap = 128; s.nz = 0; s.HX = 0; s.Lf = 1.0f; s.invL = 1.0f; s.alias = 0;
            const int c = c0 + sl;
            if (sl < G && c < C && pl.params[c].dc_len != 0 && b->ctl[c].absent == 0 && pl.params[c].generic == 0) {
                sdrm_k2_slot_setup(s, c, pl.params[c], b->ctl[c].nz);
                nb = std::max(nb, (int) ((b->ctl[c].nz + SDRM_K2_BLK - 1) / SDRM_K2_BLK));
            }
        }
        if (nb == 0) continue;
        sdrm_k2_fill_aliases(slots, G);  // as the kernel: empty slots beside live ones become replicas
        float acc[SDRM_K2_ROWS];
        for (int r = 0; r < SDRM_K2_ROWS; r++) {
            const sdrm_k2_slot &s = slots[r & (SDRM_K2_SLOTS - 1)];
            acc[r] = s.chan >= 0 ? sdrm_k2_state_acc(b->dcstate.data() + pl.params[s.chan].dc_state_off, pl.dc_hx_cap, pl.dc_l_cap)[r >> 4] : 0.0f;
        }
        for (int ring = 0; ring < 3; ring++)
            for (int sl = 0; sl < G; sl++)
                if (slots[sl].chan >= 0)
                    for (int lane = 0; lane < 64; lane++)
                        sdrm_k2_ring_load(rings.data() + ((size_t) ring * G + sl) * rpitch, slots[sl],
                                          sdrm_k2_state_tail(b->dcstate.data() + pl.params[slots[sl].chan].dc_state_off, ring, pl.dc_hx_cap, pl.dc_l_cap), lane, 64);
        auto chan_z = [&](const sdrm_k2_slot &s) { return b->z.data() + (size_t) s.chan * pl.z_stride; };
        auto chan_hx = [&](const sdrm_k2_slot &s) { return b->dcstate.data() + pl.params[s.chan].dc_state_off; };
        // role 0 = chain (one wave), roles 1..5 = feeder, stages 0..2, output (SDRM_K2_WPR waves each, a wave covers P slots)
        auto role = [&](int role_id, int it) {
            for (int lane = 0; lane < (role_id == 0 ? 64 : 64 * SDRM_K2_WPR); lane++) {
                const int slot_h = lane / SDRM_K2_LPS, q = lane % SDRM_K2_LPS;
                const sdrm_k2_slot &hs = slots[slot_h];
                const int wave = role_id;
                if (wave == 0) {
                    const int k = it - 2 * (lane >> 4);
                    if (k >= 0 && k < nb) {
                        const int buf = k % SDRM_K2_NBUF;
                        acc[lane] = sdrm_k2_chain_block(ts.data() + lane * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK,
                                                        check.data() + (lane * SDRM_K2_NBUF + buf) * SDRM_K2_LPS, acc[lane]);
                    }
                } else if (wave == 1) {
                    const int k = it + 1;
                    if (k < nb && hs.chan >= 0)
                        sdrm_k2_feed(hs, k, q, chan_z(hs), chan_hx(hs), ts.data() + slot_h * SDRM_K2_TSPITCH + (k % SDRM_K2_NBUF) * SDRM_K2_BLK);
                } else {
                    const int stage = wave - 2, k = it - 1 - 2 * stage;
                    if (k >= 0 && k < nb && hs.chan >= 0) {
                        const int buf = k % SDRM_K2_NBUF, row = stage * SDRM_K2_SLOTS + slot_h;
                        const float *in_buf = ts.data() + row * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK;
                        const float cp = check[(row * SDRM_K2_NBUF + buf) * SDRM_K2_LPS + q];
                        if (stage < 3) {
                            sdrm_k2_transition(hs, k, q, in_buf, cp, rings.data() + ((size_t) stage * G + slot_h) * rpitch,
                                               ts.data() + (row + SDRM_K2_SLOTS) * SDRM_K2_TSPITCH + buf * SDRM_K2_BLK);
                        } else {
                            b->nonfinite[hs.chan] |= sdrm_k2_output(hs, k, q, in_buf, cp, chan_z(hs), chan_hx(hs), b->dcout.data() + (size_t) hs.chan * pl.z_stride,
                                                                    sdrm_tame_level(pl.params[hs.chan], true));
                        }
                    }
                }
            }
        };
        for (int sl = 0; sl < G; sl++)
            if (slots[sl].chan >= 0)
                for (int q = 0; q < SDRM_K2_LPS; q++) sdrm_k2_feed(slots[sl], 0, q, chan_z(slots[sl]), chan_hx(slots[sl]), ts.data() + sl * SDRM_K2_TSPITCH);
        for (int it = 0; it < nb + 7; it++) {
            if (it & 1) {
                for (int w = 5; w >= 0; w--) role(w, it);
            } else {
                for (int w = 0; w < 6; w++) role(w, it);
            }
        }
        for (int r = 0; r < SDRM_K2_ROWS; r++) {
            const sdrm_k2_slot &s = slots[r & (SDRM_K2_SLOTS - 1)];
            if (s.chan >= 0 && !s.alias) sdrm_k2_state_acc(b->dcstate.data() + pl.params[s.chan].dc_state_off, pl.dc_hx_cap, pl.dc_l_cap)[r >> 4] = acc[r];
        }
        for (int ring = 0; ring < 3; ring++)
            for (int sl = 0; sl < G; sl++)
                if (slots[sl].chan >= 0 && !slots[sl].alias)
                    for (int lane = 0; lane < 64; lane++)
                        sdrm_k2_ring_save(rings.data() + ((size_t) ring * G + sl) * rpitch, slots[sl],
                                          sdrm_k2_state_tail(b->dcstate.data() + pl.params[slots[sl].chan].dc_state_off, ring, pl.dc_hx_cap, pl.dc_l_cap), lane, 64);
        const uint32_t T = 64 * SDRM_K2_WAVES;
        for (int sl = 0; sl < G; sl++) {
            const sdrm_k2_slot &s = slots[sl];
            if (s.chan < 0 || s.nz == 0 || s.alias) continue;
            float *hx = b->dcstate.data() + pl.params[s.chan].dc_state_off;
            std::vector<float> v(T);
            for (uint32_t j0 = 0; j0 < s.HX; j0 += T) {  // a round: every thread reads, barrier, every thread writes
                for (uint32_t t = 0; t < T && j0 + t < s.HX; t++) v[t] = sdrm_k2_hx_source(s, chan_z(s), hx, j0 + t);
                for (uint32_t t = 0; t < T && j0 + t < s.HX; t++) hx[j0 + t] = v[t];
            }
        }
    }
}

template <int LANES, int RING, bool PLAIN>
static void emu_clock_as(EmuBatch *b) {
    typedef sdrm_k3_geom<LANES, RING, PLAIN> G;
    const BatchPlan &pl = b->plan;
    const int C = (int) pl.params.size();
    std::vector<float> ring(G::lanes * G::cpitch);
    float bank_rev[129 * SDRM_K3_BANKPITCH];
    for (int k = 0; k < 129 * 8; k++) bank_rev[(k >> 3) * SDRM_K3_BANKPITCH + (k & 7)] = (&sdrm_mmse_bank[0][0])[(k & ~7) + 7 - (k & 7)];
    for (int c0 = 0; c0 < C; c0 += G::lanes) {
        for (auto &v : ring) v = NAN;
        sdrm_k3_lane lanes[G::lanes];
        bool clean[G::lanes];
        uint32_t flagged[G::lanes];
        bool absent[G::lanes];
        bool wild[G::lanes];
        int max_nz = 0;
        const int nl = C - c0 < G::lanes ? C - c0 : G::lanes;
        for (int l = 0; l < nl; l++) {
            const int c = c0 + l;
            const sdrm_chan_params &p = pl.params[c];
            sdrm_clock_state &cs = b->clock[c];
            sdrm_k3_lane &L = lanes[l];
            L.k = sdrm_mm_consts{p.omega_mid, p.omega_lim, p.gain_omega, p.gain_mu};
            absent[l] = b->ctl[c].absent != 0 || p.generic != 0;  // generic channels: emu_clock_generic, behind this stage
            L.cap = absent[l] ? 0u : p.max_len;  // an absent channel never steps and keeps its state
            L.nz = (int) b->ctl[c].nz;
            L.kept = absent[l] ? 0 : (int) cs.kept;
            L.oo = 0;
            L.st.mu = cs.mu;
            L.st.omega = cs.omega;
            L.st.last = cs.last;
            L.st.ii = 0;
            L.st.inc = 0;
            flagged[l] = b->nonfinite[c];
            clean[l] = flagged[l] == 0 && cs.poison == 0;
            // as the kernel: a wild channel's samples are staged, its lane never steps, sdrm_k3_rescue runs its call
            wild[l] = !absent[l] && ((((flagged[l] | cs.poison) & SDRM_FLAG_WILD) != 0) || !(p.amp_safe > 0.0f));
            if (wild[l]) {
                L.cap = 0;
                L.kept = 0;
            }
            float *col = ring.data() + l * G::cpitch;
            for (int j = 0; j < L.kept; j++) sdrm_k3_ring_put<G>(col, j - L.kept, cs.hist[j]);
            max_nz = L.nz > max_nz ? L.nz : max_nz;
        }
        const int nblocks = (max_nz + G::block - 1) / G::block;
        for (int k = 0; k <= nblocks; k++) {
            if (k < nblocks) {
                for (int r = 0; r < nl; r++) {
                    const int cr = c0 + r;
                    const float *src = (pl.params[cr].dc_len ? b->dcout.data() : b->z.data()) + (size_t) cr * pl.z_stride;
                    float *col = ring.data() + r * G::cpitch;
                    for (int n = k * G::block; n < (k + 1) * G::block && n < lanes[r].nz; n++)
                        sdrm_k3_ring_put<G>(col, n, src[n]);
                }
            }
            for (int l = 0; l < nl; l++) {
                sdrm_k3_lane &L = lanes[l];
                int avail = (k + 1) * G::block;
                avail = avail < L.nz ? avail : L.nz;
                const int c = c0 + l;
                const float *col = ring.data() + l * G::cpitch;
                const uint32_t lim = sdrm_k3_limit(L, avail);
                while (sdrm_k3_can_step(L, lim)) {
                    // the GPU picks the window/step flavour per wave; every flavour must give the same values, so the
                    // emulation lets each lane take the cheapest one its own state allows
                    sdrm_k3_operands F;
                    float soft;
                    if (clean[l]) {
                        sdrm_k3_fetch<true, G>(L, col, bank_rev, F);
                        soft = sdrm_k3_step<true>(L, F);
                    } else {
                        sdrm_k3_fetch<false, G>(L, col, bank_rev, F);
                        soft = sdrm_k3_step<false>(L, F);
                    }
                    b->out8[(size_t) c * pl.out_stride + L.oo] = clean[l] ? sdrm_soft_to_i8_finite(soft) : sdrm_soft_to_i8(soft);
                    b->outf[(size_t) c * pl.out_stride + L.oo] = soft;
                    L.oo++;
                }
            }
        }
        for (int l = 0; l < nl; l++) {
            const int c = c0 + l;
            sdrm_k3_lane &L = lanes[l];
            sdrm_clock_state &cs = b->clock[c];
            if (absent[l]) {
                b->outlen[c] = 0;
                continue;
            }
            if (wild[l]) {
                const sdrm_chan_params &p = pl.params[c];
                const float *src = (p.dc_len ? b->dcout.data() : b->z.data()) + (size_t) c * pl.z_stride;
                b->outlen[c] = sdrm_k3_rescue(p, &cs, src, L.nz, (const float *) bank_rev, b->outf.data() + (size_t) c * pl.out_stride,
                                              b->out8.data() + (size_t) c * pl.out_stride, flagged[l]);
                b->nonfinite[c] = 0;
                b->wild_calls++;
                continue;
            }
            int from_n, new_kept;
            sdrm_k3_finish(L, &from_n, &new_kept);
            const float *col = ring.data() + l * G::cpitch;
            float tmp[SDRM_CLOCK_HCAP];
            for (int j = 0; j < new_kept; j++) tmp[j] = sdrm_k3_ring_get<G>(col, from_n + j);
            for (int j = 0; j < new_kept; j++) cs.hist[j] = tmp[j];
            cs.kept = (uint32_t) new_kept;
            cs.mu = L.st.mu;
            cs.omega = L.st.omega;
            cs.last = L.st.last;
            cs.poison = (flagged[l] != 0 || !(fabsf(lanes[l].st.mu) < INFINITY) || !(fabsf(lanes[l].st.omega) < INFINITY) ||
                         !(fabsf(lanes[l].st.last) < INFINITY)) ? SDRM_FLAG_NONFINITE : 0u;  // as the kernel: a non-finite loop state stays off the fast path
            b->nonfinite[c] = 0;
            b->outlen[c] = L.oo;
        }
    }
}

// the workgroup shape the library would launch for this batch (same policy, same SDRM_K3_LANES override)
static void emu_clock(EmuBatch *b) {
    int lanes = 0, ring = 0, plain = 0;
    sdrm_k3_parse_shape(getenv("SDRM_K3_LANES"), &lanes, &ring, &plain);
    const sdrm_k3_shape sh = sdrm_k3_shape_for((int) b->plan.params.size(), lanes, ring, plain, (int) b->plan.clock_carried_max);
    switch ((sh.lanes * 10000 + sh.ring) * (sh.plain ? -1 : 1)) {
        case 16 * 10000 + 1024: emu_clock_as<16, 1024, false>(b); break;
        case 16 * 10000 + 512: emu_clock_as<16, 512, false>(b); break;
        case 16 * 10000 + 256: emu_clock_as<16, 256, false>(b); break;
        case 32 * 10000 + 512: emu_clock_as<32, 512, false>(b); break;
        case 32 * 10000 + 256: emu_clock_as<32, 256, false>(b); break;
        case -(64 * 10000 + 256): emu_clock_as<64, 256, true>(b); break;
        case -(32 * 10000 + 256): emu_clock_as<32, 256, true>(b); break;
        default: emu_clock_as<64, 256, false>(b); break;
    }
}

// generic channels, as k2_dc_generic / k3_clock_generic run them (same layout helpers, same per-sample bodies; the block
// structure of the DC kernel is kept so that ring positions and the partial last block are exercised)
static void emu_dc_generic(EmuBatch *b) {
    const BatchPlan &pl = b->plan;
    for (size_t c = 0; c < pl.params.size(); c++) {
        const sdrm_chan_params &p = pl.params[c];
        const sdrm_chunk_ctl &ctl = b->ctl[c];
        if (!p.generic || p.dc_len == 0 || ctl.absent != 0 || ctl.nz == 0) continue;
        const sdrm_gen_layout g = sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim);
        float *st = b->gen[c].data();
        float acc[4] = {st[0], st[1], st[2], st[3]};
        uint32_t pos = sdrm_bits(st[4]), xpos = sdrm_bits(st[5]);
        const float *z = b->z.data() + c * pl.z_stride;
        float *out = b->dcout.data() + c * pl.z_stride;
        const int nz = (int) ctl.nz;
        for (int n0 = 0; n0 < nz; n0 += 64) {
            const int valid = nz - n0 < 64 ? nz - n0 : 64;
            float u[64], x[64];
            for (int l = 0; l < valid; l++) x[l] = u[l] = z[n0 + l];
            for (int s = 0; s < 4; s++) {
                float *ring = st + g.off_ring + (size_t) s * g.L;
                float t[64];
                for (int l = 0; l < valid; l++) {  // pointwise, every lane at once
                    uint32_t idx = pos + (uint32_t) l;
                    idx = idx >= g.L ? idx - g.L : idx;
                    const float old = ring[idx];
                    ring[idx] = u[l];
                    t[l] = sdrm_boxcar_term(u[l], old);
                }
                float run = acc[s];
                for (int l = 0; l < valid; l++) {  // the in-order chain
                    run = t[l] + run;
                    u[l] = sdrm_boxcar_out(run, p.dc_len_f);
                }
                acc[s] = run;
            }
            float *xring = st + g.off_x;
            for (int l = 0; l < valid; l++) {
                uint32_t xi = xpos + (uint32_t) l;
                xi = xi >= g.XL ? xi - g.XL : xi;
                const float delayed = xring[xi];
                xring[xi] = x[l];
                out[n0 + l] = delayed - u[l];
            }
            pos += (uint32_t) valid;
            pos = pos >= g.L ? pos - g.L : pos;
            xpos += (uint32_t) valid;
            xpos = xpos >= g.XL ? xpos - g.XL : xpos;
        }
        for (int s = 0; s < 4; s++) st[s] = acc[s];
        st[4] = sdrm_from_bits(pos);
        st[5] = sdrm_from_bits(xpos);
    }
}

static void emu_clock_generic(EmuBatch *b) {
    const BatchPlan &pl = b->plan;
    float bank_rev[129 * SDRM_K3_BANKPITCH];
    for (int k = 0; k < 129 * 8; k++) bank_rev[(k >> 3) * SDRM_K3_BANKPITCH + (k & 7)] = (&sdrm_mmse_bank[0][0])[(k & ~7) + 7 - (k & 7)];
    for (size_t c = 0; c < pl.params.size(); c++) {
        const sdrm_chan_params &p = pl.params[c];
        if (!p.generic) continue;
        const sdrm_chunk_ctl &ctl = b->ctl[c];
        if (ctl.absent != 0) {
            b->outlen[c] = 0;
            continue;
        }
        const sdrm_gen_layout g = sdrm_gen_layout_for(p.dc_len, p.omega_mid, p.max_len, p.decim);
        float *work = b->gen[c].data() + g.off_work;
        sdrm_clock_state &cs = b->clock[c];
        const int kept = (int) cs.kept, nz = (int) ctl.nz;
        const float *src = (p.dc_len ? b->dcout.data() : b->z.data()) + c * pl.z_stride;
        for (int i = 0; i < nz; i++) work[3 + kept + i] = src[i];
        sdrm_k3_lane L;
        L.k = sdrm_mm_consts{p.omega_mid, p.omega_lim, p.gain_omega, p.gain_mu};
        L.kept = 0;
        L.nz = kept + nz;
        L.oo = 0;
        L.cap = p.max_len;
        L.st.mu = cs.mu;
        L.st.omega = cs.omega;
        L.st.last = cs.last;
        L.st.ii = 0;
        L.st.inc = 0;
        const uint32_t lim = sdrm_k3_limit(L, L.nz);
        while (sdrm_k3_can_step(L, lim)) {
            sdrm_k3_operands F;
            sdrm_k3_fetch<false, sdrm_k3_geom_linear>(L, (const float *) work, (const float *) bank_rev, F);
            const float soft = sdrm_k3_step<false>(L, F);
            b->outf[c * pl.out_stride + L.oo] = soft;
            b->out8[c * pl.out_stride + L.oo] = sdrm_soft_to_i8(soft);
            L.oo++;
        }
        int from, keep;
        sdrm_k3_finish_linear(L, g.hcap, &from, &keep);
        if (from > 0) memmove(work + 3, work + 3 + from, sizeof(float) * (size_t) keep);
        cs.mu = L.st.mu;
        cs.omega = L.st.omega;
        cs.last = L.st.last;
        cs.kept = (uint32_t) keep;
        cs.poison = 0;
        b->nonfinite[c] = 0;
        b->outlen[c] = L.oo;
    }
}

extern "C" uint64_t emu_wild_calls(const EmuBatch *b) { return b->wild_calls; }

extern "C" int emu_set_pre_offset(EmuBatch *b, size_t c, int64_t freq_hz) {
    const size_t C = b->plan.params.size();
    if (c >= C) return -1;
    b->pre_offset.resize(C, 0);
    b->pre_state.resize(C, 0.0f);
    b->pre_mixed.resize(C);
    b->pre_offset[c] = freq_hz;
    b->pre_state[c] = 0.0f;
    return 0;
}

// the oscillator in front of everything else: one batch over the channel's whole input of the call (as the device plans it)
static const sdrm_f2 *emu_pre_mix(EmuBatch *b, size_t c, const sdrm_f2 *in) {
    if (c >= b->pre_offset.size() || b->pre_offset[c] == 0 || b->ctl[c].n_in == 0 || b->ctl[c].absent) return in;
    const float two_pi = (float) (2 * 3.14159265358979323846);
    const float step = two_pi * (float) b->pre_offset[c] / b->plan.design[c].cfg.sampling_freq;
    std::vector<sdrm_f2> &m = b->pre_mixed[c];
    m.resize(b->ctl[c].n_in);
    float phase = b->pre_state[c];
    for (uint32_t n = 0; n < b->ctl[c].n_in; n++) {
        m[n] = sdrm_nco_mix(in[n], sdrm_nco_sample(phase));
        phase = sdrm_nco_advance(phase, step);
    }
    b->pre_state[c] = phase;
    return m.data();
}

// inputs[c]: interleaved cf32, lens[c] samples.  Outputs: per channel pointers into emu-owned memory.
extern "C" int emu_process(EmuBatch *b, const float *const *inputs, const size_t *lens, const int8_t **out8,
                           const float **outf, size_t *outlens) {
    const size_t C = b->plan.params.size();
    plan_call(b->plan, lens, b->ctl.data());
    std::vector<const sdrm_f2 *> ins(C);
    for (size_t c = 0; c < C; c++) ins[c] = emu_pre_mix(b, c, reinterpret_cast<const sdrm_f2 *>(inputs[c]));
    emu_front_any(b, ins.data());
    emu_dc(b);
    emu_dc_generic(b);
    emu_clock(b);
    emu_clock_generic(b);
    for (size_t c = 0; c < C; c++) {
        out8[c] = b->out8.data() + c * b->plan.out_stride;
        outf[c] = b->outf.data() + c * b->plan.out_stride;
        outlens[c] = b->outlen[c];
    }
    return 0;
}

// same call with the NCO pre-mix in front (K0: lane-per-channel phase chain, then the pointwise mix)
extern "C" int emu_process_nco(EmuBatch *b, const float *const *inputs, const size_t *lens, const sdrm_nco_segment *segs,
                               size_t n_segs, const int8_t **out8, const float **outf, size_t *outlens) {
    const size_t C = b->plan.params.size();
    plan_call(b->plan, lens, b->ctl.data());
    if (plan_nco(b->plan, segs, n_segs, b->ctl.data(), b->nco_table) != 0) {
        return -1;
    }
    std::vector<const sdrm_f2 *> ins(C);
    for (size_t c = 0; c < C; c++) {
        const sdrm_chunk_ctl &k = b->ctl[c];
        const sdrm_f2 *in = emu_pre_mix(b, c, reinterpret_cast<const sdrm_f2 *>(inputs[c]));
        if (k.nco_cnt == 0) {
            ins[c] = in;
            continue;
        }
        std::vector<sdrm_f2> &m = b->mixed[c];
        m.resize(k.n_in);
        float phase = b->nco_state[c];
        uint32_t n = 0;
        for (uint32_t s = 0; s < k.nco_cnt; s++) {
            const sdrm_nco_seg &sg = b->nco_table[k.nco_off + s];
            for (uint32_t i = 0; i < sg.len; i++, n++) {
                m[n] = sdrm_nco_mix(in[n], sdrm_nco_sample(phase));
                phase = sdrm_nco_advance(phase, sg.step);
            }
        }
        b->nco_state[c] = phase;
        ins[c] = m.data();
    }
    emu_front_any(b, ins.data());
    emu_dc(b);
    emu_dc_generic(b);
    emu_clock(b);
    emu_clock_generic(b);
    for (size_t c = 0; c < C; c++) {
        out8[c] = b->out8.data() + c * b->plan.out_stride;
        outf[c] = b->outf.data() + c * b->plan.out_stride;
        outlens[c] = b->outlen[c];
    }
    return 0;
}

extern "C" size_t emu_mixed(EmuBatch *b, size_t c, float *dst, size_t cap) {
    const std::vector<sdrm_f2> &m = b->mixed[c];
    for (size_t i = 0; i < m.size() && i < cap; i++) {
        dst[2 * i] = m[i].x;
        dst[2 * i + 1] = m[i].y;
    }
    return m.size();
}

// the Doppler planner (host code of the product) for CPU tests
struct EmuShift {
    const double *v;
    size_t n;
};
static double emu_shift_cb(void *user, uint64_t k) {
    EmuShift *s = (EmuShift *) user;
    return s->n ? s->v[k < s->n ? k : s->n - 1] : 0.0;
}
extern "C" size_t emu_doppler_plan_stream(uint64_t fs, const double *shifts, size_t n_shifts, const size_t *call_lens,
                                          size_t n_calls, sdrm_nco_segment *out, size_t cap, size_t *per_call_counts) {
    EmuShift sh{shifts, n_shifts};
    DopplerPlanner p;
    p.interval = fs;
    p.in_interval = fs;
    p.fn = emu_shift_cb;
    p.user = &sh;
    size_t total = 0;
    for (size_t i = 0; i < n_calls; i++) {
        size_t k = p.plan(0, call_lens[i], out + total, cap - total);
        per_call_counts[i] = k;
        total += k;
    }
    return total;
}

// the branch-free arctangent the front-end kernel runs (sdrm_fast_atan2f_flat), on the host
extern "C" void emu_fast_atan2f_flat(const float *y, const float *x, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i] = sdrm_fast_atan2f_flat(y[i], x[i], sdrm_atan_tab);
    }
}

// the same in the reference's own shape (sdrm_fast_atan2f, the if-tree)
extern "C" void emu_fast_atan2f_tree(const float *y, const float *x, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i] = sdrm_fast_atan2f(y[i], x[i], sdrm_atan_tab);
    }
}

// stage taps for inspection
extern "C" size_t emu_taps(EmuBatch *b, size_t c, int stage, float *dst, size_t cap) {
    const std::vector<float> &t = stage == 2 ? b->plan.design[c].taps2 : b->plan.design[c].taps1;
    for (size_t i = 0; i < t.size() && i < cap; i++) dst[i] = t[i];
    return t.size();
}

extern "C" void emu_info(EmuBatch *b, size_t c, sdrm_fsk_info *info) {
    const ChannelDesign &d = b->plan.design[c];
    info->taps1_len = (uint32_t) d.taps1.size();
    info->taps2_len = (uint32_t) d.taps2.size();
    info->dc_length = d.dc_length;
    info->quad_gain = d.quad_gain;
    info->sps = d.sps;
    info->gain_omega = d.gain_omega;
    info->gain_mu = d.gain_mu;
    info->omega_lim = d.omega_lim;
}

// sdrm_boxcar_out_fast against the division it stands for: every fp32 significand at biased exponent `bexp`, both signs.
// Returns the number of quotients that differ where the short form did not raise its flag; *unsafe_count = flagged ones.
extern "C" uint64_t emu_check_boxcar_div(uint32_t length, uint32_t bexp, uint64_t *unsafe_count) {
    const float lf = (float) length, inv = 1.0f / lf;
    uint64_t bad = 0, uns = 0;
    for (uint32_t sign = 0; sign < 2; sign++) {
        for (uint32_t m = 0; m < (1u << 23); m++) {
            const float a = sdrm_from_bits((sign << 31) | (bexp << 23) | m);
            bool unsafe;
            const float got = sdrm_boxcar_out_fast(a, lf, inv, &unsafe);
            if (unsafe) {
                uns++;
                continue;
            }
            const float want = sdrm_boxcar_out(a, lf);
            if (sdrm_bits(got) != sdrm_bits(want) && !(got != got && want != want)) bad++;
        }
    }
    if (unsafe_count) *unsafe_count = uns;
    return bad;
}

// sdrm_nco_advance_nomask (what the phase kernel's hand-written loop computes) against the reference's two-test wrap
// sdrm_nco_advance: `n` pairs (phase, step), both within +-2 pi.  Returns how many results differ in their bits.
extern "C" uint64_t emu_check_nco_advance(const float *phase, const float *step, size_t n) {
    uint64_t bad = 0;
    for (size_t i = 0; i < n; i++) {
        const float want = sdrm_nco_advance(phase[i], step[i]);
        const float got = sdrm_nco_advance_nomask(phase[i], step[i], sdrm_nco_wrap_bigs(step[i]), sdrm_nco_wrap_negw(step[i]));
        if (sdrm_bits(got) != sdrm_bits(want)) bad++;
    }
    return bad;
}
// the same along the recursion itself: `len` steps from `phase0`; returns the first sample where the two forms part (len: never)
extern "C" size_t emu_check_nco_run(float phase0, float step, size_t len) {
    float a = phase0, b = phase0;
    const float bigs = sdrm_nco_wrap_bigs(step), negw = sdrm_nco_wrap_negw(step);
    for (size_t i = 0; i < len; i++) {
        a = sdrm_nco_advance(a, step);
        b = sdrm_nco_advance_nomask(b, step, bigs, negw);
        if (sdrm_bits(a) != sdrm_bits(b)) return i;
    }
    return len;
}

// NCO sample helpers for the CPU suite: the double-double sin/cos, and a sweep that counts, over `n` consecutive fp32
// phases starting at `first_bits`, how often the fragile path was taken and how often the result differs from the host
// libm's (float) cos / (float) sin
extern "C" void emu_sincos_dd(double x, double *out4) {
    sdrm_dd s, c;
    sdrm_sincos_dd(x, &s, &c);
    out4[0] = s.hi; out4[1] = s.lo; out4[2] = c.hi; out4[3] = c.lo;
}
extern "C" void emu_nco_sample(float phase, float *out2) {
    const sdrm_f2 v = sdrm_nco_sample(phase);
    out2[0] = v.x; out2[1] = v.y;
}
extern "C" void emu_nco_sweep(uint32_t first_bits, uint32_t n, uint32_t stride, uint64_t *fragile, uint64_t *differ,
                              float *fragile_phases, uint32_t cap) {
    uint64_t fr = 0, df = 0;
    for (uint32_t i = 0; i < n; i++) {
        const float ph = sdrm_from_bits(first_bits + i * stride);
        const double x = (double) ph, c = cos(x), s = sin(x);
        if (sdrm_f32_rounding_is_fragile(c) || sdrm_f32_rounding_is_fragile(s)) {
            if (fr < cap) fragile_phases[fr] = ph;
            fr++;
        }
        const sdrm_f2 v = sdrm_nco_sample(ph);
        if (sdrm_bits(v.x) != sdrm_bits((float) c) || sdrm_bits(v.y) != sdrm_bits((float) s)) df++;
    }
    *fragile = fr;
    *differ = df;
}
