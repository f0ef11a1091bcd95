"""CPU-only checks of the HIP path's LOGIC: the per-thread kernel bodies (sdrm_kernels.h) and the host planning
(sdrm_plan.cpp, sdrm_design.cpp) are driven on the host by tests/emu and must reproduce the oracle bit for bit --
same tiling, history hand-off, decimation phase, DC rings and block-wise clock loop the GPU will execute.
(The arithmetic itself -- no FMA, IEEE divide, denormals on gfx950 -- is what the `-m gpu` tests check.)"""
import os

import numpy as np
import pytest

import emu_api
import orc
import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd import siggen  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_both(cfg, iq, chunks, maxlen):
    """cfg = (fs, baud, dev, decim, tw, dc). chunks = list of chunk lengths. Returns per-call (oracle, emu) outputs."""
    o = orc.Fsk(*cfg, maxlen)
    e = emu_api.EmuBatch([cfg + (maxlen,)])
    assert o.code == 0 and e.code == 0
    pos = 0
    for n in chunks:
        part = iq[pos:pos + n]
        pos += n
        o8, of = o.process(part)
        e8, ef = e.process([part])
        assert len(o8) == len(e8[0]), (pos, len(o8), len(e8[0]))
        assert np.array_equal(of.view(np.uint32), ef[0].view(np.uint32)), (pos, np.abs(of - ef[0]).max())
        assert np.array_equal(o8, e8[0])
    return pos


CONFIGS = [
    (48000, 9600, 5000, 1, 2000, True),
    (48000, 9600, 5000, 1, 2000, False),
    (48000, 4800, 5000, 2, 2000, True),
    (48000, 4800, 5000, 2, 2000, False),
    (240000, 19200, 5000, 5, 2000, True),
    (48000, 1200, 5000, 8, 2000, True),
    (192000, 40000, 5000, 1, 2000, True),
]


def atan_cases():
    """(y, x) pairs: every octant, the axes and diagonals, signed zeros, denormals, the table's knots and the double-typed
    threshold TAN_MAP_RES from both sides, quotients that underflow, Inf and NaN"""
    rng = np.random.default_rng(3)
    y = np.concatenate([rng.standard_normal(200000), [0, 0, 1, -1, 1e-3, -1e-9, 0.0, -0.0, np.inf, np.nan, 1e-42, 3e38, 1e-39,
                                                      -1e-45, 1e-45, -0.0, 0.0, 1, -1, 1, -1, np.nan, np.inf, -np.inf, 2.0]])
    x = np.concatenate([rng.standard_normal(200000), [0, -1, 1, -1, 1, -1, -0.0, 0.0, 1.0, 1.0, 1e-40, 3e38, 3e-39,
                                                      3e38, -3e38, 1.0, -1.0, 1, 1, -1, -1, np.nan, np.inf, np.inf, -0.0]])
    k = np.arange(1, 256, dtype=np.float64) / 255.0
    res = np.float32(0.003921569)
    edge = [res, np.nextafter(res, np.float32(1)), np.nextafter(res, np.float32(0)), 0.0039215689, 0.00392157]
    ys = [y, k, k * (1 + 1e-7), k * (1 - 1e-7), edge]
    xs = [x, np.ones(3 * 255 + len(edge))]
    y = np.concatenate(ys).astype(np.float32)
    x = np.concatenate(xs).astype(np.float32)
    # and every sign / swap of the lot
    return (np.concatenate([y, -y, y, -y, x, -x, x, -x]), np.concatenate([x, x, -x, -x, y, y, -y, -y]))


def test_branch_free_arctangent_equals_the_reference_form():
    """the front-end kernel evaluates fast_atan2f without branches (sdrm_fast_atan2f_flat: one division with selected
    operands, offset + signed base); bit-identical to the oracle's transcription of src/math/fast_atan2f.c:87-157"""
    y, x = atan_cases()
    out = np.zeros(len(y), np.float32)
    emu_api.lib().emu_fast_atan2f_flat(y.ctypes.data, x.ctypes.data, out.ctypes.data, len(y))
    f = orc.lib().orc_fast_atan2f
    want = np.array([f(a, b) for a, b in zip(y, x)], dtype=np.float32)
    both_nan = np.isnan(out) & np.isnan(want)
    assert np.array_equal(out.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan])
    assert np.array_equal(np.isnan(out), np.isnan(want))
    # and the if-tree form kept next to it (sdrm_fast_atan2f) says the same
    tree = np.zeros(len(y), np.float32)
    emu_api.lib().emu_fast_atan2f_tree(y.ctypes.data, x.ctypes.data, tree.ctypes.data, len(y))
    both_nan = np.isnan(out) & np.isnan(tree)
    assert np.array_equal(out.view(np.uint32)[~both_nan], tree.view(np.uint32)[~both_nan])


@pytest.mark.parametrize("cfg", CONFIGS, ids=[str(c) for c in CONFIGS])
def test_design_matches_oracle(cfg):
    o = orc.Fsk(*cfg, 4096)
    e = emu_api.EmuBatch([cfg + (4096,)])
    inf, t1, t2 = o.info()
    einf = e.info(0)
    for f in ("taps1_len", "taps2_len", "dc_length", "quad_gain", "sps", "gain_omega", "gain_mu", "omega_lim"):
        assert getattr(inf, f) == getattr(einf, f), f
    assert np.array_equal(t1.view(np.uint32), e.taps(0, 1).view(np.uint32))
    assert np.array_equal(t2.view(np.uint32), e.taps(0, 2).view(np.uint32))


@pytest.mark.parametrize("cfg", CONFIGS[:4], ids=[str(c) for c in CONFIGS[:4]])
def test_lucky7_stream_bit_exact(cfg):
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    run_both(cfg, iq, [4096] * 23 + [96000 - 23 * 4096], 4096)


def test_ragged_chunks_cross_tiles_and_phases():
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    chunks = [0, 1, 1, 1, 7, 0, 100, 3839, 3840, 3841, 5000, 9000, 1, 2, 8191, 12000, 64, 63, 65, 1]
    for cfg in [(48000, 9600, 5000, 1, 2000, True), (48000, 4800, 5000, 2, 2000, True), (48000, 4800, 5000, 3, 2000, True)]:
        assert run_both(cfg, iq, chunks, 12000) <= len(iq)


@pytest.mark.parametrize("lanes", ["16", "32", "64", "16x512", "16x256", "32x256", "64x256p", "32x256p"])
def test_clock_stage_workgroup_shapes(lanes, monkeypatch):
    """the clock stage's workgroup shapes (channels x ring length, pair-element or plain rings; SDRM_K3_LANES forces one)
    stage and drain the same samples: ragged chunks around every step size"""
    monkeypatch.setenv("SDRM_K3_LANES", lanes)
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    chunks = [0, 1, 7, 100, 255, 256, 257, 1023, 1024, 1025, 3839, 5000, 9000, 1, 8191, 12000, 64, 63, 65, 1]
    for cfg in [(48000, 9600, 5000, 1, 2000, True), (48000, 4800, 5000, 2, 2000, False)]:
        assert run_both(cfg, iq, chunks, 12000) <= len(iq)


def test_two_hundred_samples_per_symbol():
    """240 kHz / 1200 baud without decimation (the reference's defaults for a 1200-baud satellite on a 240 kHz stream):
    200 samples per symbol, a 1033-tap LPF1, a 6400-sample DC boxcar (one channel per DC workgroup), up to 210 samples
    carried by the clock stage between calls -- which re-emits a symbol at chunk edges there (clock_recovery_mm.c:127-133),
    so the chunking is the oracle's.  Refused (-ENOTSUP) until round 3."""
    iq = siggen.gmsk_channel(5, 66000, fs=240000, baud=1200)
    run_both((240000, 1200, 5000, 1, 2000, True), iq, [16384, 1000, 16384, 7, 20000, 12225], 20000)
    run_both((240000, 1200, 5000, 1, 2000, False), iq[:30000], [9000, 300, 20700], 20700)


def test_any_samples_per_symbol_through_the_generic_stages():
    """The reference accepts any samples per symbol (src/dsp/fsk_demod.c:53-63).  Beyond what the LDS-resident stages are sized
    for -- more than ~244 samples per symbol, or a DC boxcar longer than 7712 samples -- a channel keeps the fast front-end and
    runs its DC blocker and clock recovery in their generic forms (state in global memory; -ENOTSUP until round 4):
    240 kHz / 600 baud (400 samples per symbol, 1091-tap LPF1, a 12800-sample boxcar, up to 412 carried samples) and
    240 kHz / 900 baud without DC blocker (266.7), over ragged calls including empty and tiny ones, and in one batch with an
    ordinary channel, whose stream must not notice its neighbour."""
    iq = siggen.gmsk_channel(6, 130000, fs=240000, baud=600)
    chunks = [16384, 1000, 0, 16384, 7, 20000, 1, 1, 300, 20000, 20000, 20000, 15923]
    assert run_both((240000, 600, 5000, 1, 2000, True), iq, chunks, 20000) == 130000
    iq2 = siggen.gmsk_channel(7, 60000, fs=240000, baud=900)
    run_both((240000, 900, 5000, 1, 2000, False), iq2, [9000, 300, 20700, 5, 20000, 9995], 20700)
    # the longest filters that fit a tile's LDS (5899 + 2891 taps: halo longer than the tile, history longer than the calls)
    iq3 = siggen.gmsk_channel(9, 30000, fs=2400000, baud=9600)
    run_both((2400000, 9600, 5000, 1, 2000, True), iq3, [4096, 4096, 100, 4096, 7, 8192, 3000], 8192)
    run_both((240000, 900, 5000, 1, 2000, True), iq2, [20700, 20700, 18600], 20700)  # 8534-sample boxcar: DC generic, clock too
    # one batch: generic, ordinary, generic without DC
    cfgs = [(240000, 600, 5000, 1, 2000, True, 8192), (48000, 9600, 5000, 1, 2000, True, 8192), (240000, 900, 5000, 1, 2000, False, 8192)]
    sigs = [siggen.gmsk_channel(20 + i, 3 * 8192, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    e = emu_api.EmuBatch(cfgs)
    assert e.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs]
    for lens in ([8192, 8192, 8192], [100, 8192, 0], [8092, 3000, 8192]):
        parts, nxt = [], []
        for i, n in enumerate(lens):
            used = getattr(test_any_samples_per_symbol_through_the_generic_stages, "_pos", [0, 0, 0])
            parts.append(sigs[i][used[i]:used[i] + n])
            nxt.append(used[i] + n)
        test_any_samples_per_symbol_through_the_generic_stages._pos = nxt
        e8, ef = e.process(parts)
        for i, o in enumerate(oracles):
            o8, of = o.process(parts[i])
            assert np.array_equal(o8, e8[i]) and np.array_equal(of.view(np.uint32), ef[i].view(np.uint32)), (i, lens)
    del test_any_samples_per_symbol_through_the_generic_stages._pos
    # a slot handed from an ordinary client to a generic one and back (the batch neither grows its LDS rings nor keeps the state)
    assert e.reset_channel(1, (240000, 600, 5000, 1, 2000, True, 8192)) == 0
    o = orc.Fsk(240000, 600, 5000, 1, 2000, True, 8192)
    e8, ef = e.process([emu_api.ABSENT, sigs[0][:8192], emu_api.ABSENT])
    assert np.array_equal(o.process(sigs[0][:8192])[0], e8[1])
    assert e.reset_channel(1, (48000, 9600, 5000, 1, 2000, True, 8192)) == 0
    o = orc.Fsk(48000, 9600, 5000, 1, 2000, True, 8192)
    e8, ef = e.process([emu_api.ABSENT, sigs[1][:8192], emu_api.ABSENT])
    assert np.array_equal(o.process(sigs[1][:8192])[0], e8[1])


def noise(seed, n, sigma):
    rng = np.random.default_rng(seed)
    return (rng.normal(0, sigma, n) + 1j * rng.normal(0, sigma, n)).astype(np.complex64)


def run_counting(cfg, parts, maxlen):
    """the stream `parts` (list of complex64 calls) through oracle and emulation, bit for bit; returns (symbols, wild channel-calls)"""
    o = orc.Fsk(*cfg, maxlen)
    e = emu_api.EmuBatch([cfg + (maxlen,)])
    assert o.code == 0 and e.code == 0, (cfg, o.code, e.code)
    total = 0
    for k, part in enumerate(parts):
        o8, of = o.process(part)
        e8, ef = e.process([part])
        assert len(o8) == len(e8[0]), (cfg, k, len(o8), len(e8[0]))
        assert np.array_equal(of.view(np.uint32), ef[0].view(np.uint32)), (cfg, k)
        assert np.array_equal(o8, e8[0]), (cfg, k)
        total += len(o8)
    return total, e.wild_calls()


@pytest.mark.parametrize("cfg", [(48000, 9600, 1, 1, 2000, True), (240000, 19200, 4, 2, 2000, True), (48000, 19200, 4, 2, 2000, True),
                                 (48000, 9600, -3, 1, 2000, False)], ids=str)
def test_timing_loop_that_walks_backwards_through_its_buffer(cfg):
    """A discriminator gain in the thousands (a deviation of a few Hz) on noise: the timing error is thousands of samples, the
    reference's loop walks BACKWARDS through its working buffer (`ii += (int) floorf(mu)` with mu < 0,
    src/dsp/clock_recovery_mm.c:121-122; defined while ii stays in [0, working_len)), or stands still and fills its output
    buffer.  An LDS ring remembers two staging blocks: until round 5 the ring-based loop read overwritten slots there and
    returned WRONG symbols without an error (round-4 review: oracle 185 symbols, emulation 135 on the first case).  Now the
    stage in front of the clock stage flags every sample beyond the amplitude up to which the loop provably advances
    (sdrm_amp_safe), and a flagged channel's call is run from global memory (sdrm_k3_rescue): same bits as the oracle."""
    symbols, wild = run_counting(cfg, [noise(99, 5000, 0.7)], 5000)
    assert wild == 1 and symbols > 0
    # a stream of calls: full, ragged, empty; the state a wild call leaves behind (carried samples, last symbol) is wild too
    parts = [noise(100 + k, n, 0.7) for k, n in enumerate([5000, 1, 0, 4999, 7, 2500, 5000])]
    symbols, wild = run_counting(cfg, parts, 5000)
    assert wild == len(parts)


def test_a_wild_channel_returns_to_the_ring_based_loop_when_its_signal_does():
    """wild calls (noise at a gain of 7600), then a clean carrier with a little noise: after the call whose carried samples and
    last symbol are tame again, the channel is back on the LDS-resident loop -- bit-equal throughout"""
    cfg = (48000, 9600, 1, 1, 2000, False)
    quiet = [(np.exp(2j * np.pi * 1e-6 * np.arange(4000 * k, 4000 * (k + 1))) + noise(7 + k, 4000, 1e-5)).astype(np.complex64) for k in range(6)]
    parts = [noise(1, 4000, 0.5), noise(2, 4000, 0.5)] + quiet
    symbols, wild = run_counting(cfg, parts, 4000)
    assert 2 <= wild <= 4, wild  # the two noise calls, and at most the two calls their filter tails and carried samples reach


def test_ordinary_signals_never_leave_the_ring_based_loop():
    """every reference fixture configuration on noisy GMSK, and on noise alone at full scale: no call is flagged"""
    for i, cfg in enumerate(CONFIGS):
        iq = siggen.gmsk_channel(40 + i, 3 * 8192, fs=cfg[0], baud=cfg[1])
        symbols, wild = run_counting(cfg, [iq[:8192], iq[8192:16384], noise(i, 8192, 1.0)], 8192)
        assert wild == 0, (cfg, wild)


def test_fewer_than_one_sample_per_symbol():
    """48 kHz / 9600 baud decimated by 7 (0.71 samples per symbol): the reference accepts it (src/dsp/fsk_demod.c:53-63 has no
    range check) and produces more symbols than it has samples, until its output buffer is full; -ENOTSUP until round 5.
    The timing loop is not tame at any amplitude there (omega - limit < 1): such a channel always runs from global memory."""
    iq = siggen.gmsk_channel(3, 20000, fs=48000, baud=9600)
    for cfg in [(48000, 9600, 5000, 7, 2000, True), (48000, 9600, 5000, 5, 2000, False), (48000, 19200, 5000, 4, 2000, True),
                (48000, 9600, 5000, 50, 2000, True)]:
        symbols, wild = run_counting(cfg, [iq[:8000], iq[8000:8003], iq[8003:16003], iq[16003:20000]], 8000)
        assert wild == 4 and symbols > 0, (cfg, symbols, wild)


def test_low_deviation_fuzz_against_the_oracle():
    """deviation log-uniform in 1 .. 1000 Hz (discriminator gains 7 .. 38000), every sample rate / decimation / DC choice,
    noise over three decades with and without a carrier, ragged multi-call streams: oracle == emulation bit for bit, whichever
    of the two clock-stage forms each call takes (both must be taken)"""
    rng = np.random.default_rng(2025)
    wild_total = calls_total = 0
    for it in range(250):
        fs, baud = [(48000, 9600), (48000, 4800), (240000, 19200), (48000, 19200), (192000, 40000), (48000, 1200), (96000, 9600)][rng.integers(7)]
        cfg = (fs, baud, int(np.exp(rng.uniform(0, np.log(1000)))) * int(rng.choice([1, 1, 1, -1])), int(rng.choice([1, 1, 2, 4, 5, 8])), 2000,
               bool(rng.integers(2)))
        maxlen = int(rng.choice([3000, 5000, 8192]))
        sigma = float(np.exp(rng.uniform(np.log(1e-3), np.log(2))))
        parts = []
        for k in range(int(rng.integers(1, 5))):
            n = int(rng.choice([maxlen, maxlen, rng.integers(0, maxlen + 1), rng.integers(0, 50)]))
            x = noise(int(rng.integers(1 << 30)), n, sigma)
            if it % 3 == 1:
                x = (x + np.exp(2j * np.pi * rng.uniform(-0.1, 0.1) * np.arange(n))).astype(np.complex64)
            parts.append(x)
        symbols, wild = run_counting(cfg, parts, maxlen)
        wild_total += wild
        calls_total += len(parts)
    assert 0.2 * calls_total < wild_total < calls_total, (wild_total, calls_total)


def test_the_bound_on_an_interpolated_symbol_covers_the_filter_bank():
    """SDRM_MMSE_ABS_SUM (sdrm_kernels.h) >= max over the bank's rows of sum |tap|: what sdrm_amp_safe's proof uses"""
    import re
    root = os.path.dirname(GOLDEN)
    text = open(os.path.join(root, "..", "sdr-modem_amd", "csrc", "sdrm_tables.h")).read()
    body = text[text.index("sdrm_mmse_bank[129][8]"):]
    body = body[:body.index("};")]
    bank = np.array(re.findall(r"-?\d+\.\d+(?:e[+-]\d+)?", body), dtype=np.float64).reshape(129, 8)
    hdr = open(os.path.join(root, "..", "sdr-modem_amd", "csrc", "sdrm_kernels.h")).read()
    bound = float(re.search(r"#define SDRM_MMSE_ABS_SUM ([0-9.]+)f", hdr).group(1))
    assert np.abs(bank).sum(axis=1).max() <= bound < np.abs(bank).sum(axis=1).max() * 1.001


def test_synthetic_gmsk_large_chunk():
    iq = siggen.gmsk_channel(0, 70000)
    run_both((48000, 9600, 5000, 1, 2000, True), iq, [65536, 4464], 65536)


def test_mixed_rate_configs_with_decimation():
    for cfg, fs, baud in [((240000, 19200, 5000, 5, 2000, True), 240000, 19200),
                          ((48000, 1200, 5000, 8, 2000, True), 48000, 1200)]:
        iq = siggen.gmsk_channel(3, 30000, fs=fs, baud=baud)
        run_both(cfg, iq, [10000, 5, 19995], 20000)


@pytest.mark.parametrize("cfg,name", [((192000, 40000, 5000, 1, 2000, True), "nusat.cf32"),
                                      ((240000, 9600, 5000, 1, 2000, True), "inputnan.cf32")])
def test_reference_fixtures_nusat_and_nan(cfg, name):
    iq = np.fromfile(os.path.join(GOLDEN, name), dtype=np.complex64)
    n = len(iq)
    chunks = [4096] * (n // 4096) + ([n % 4096] if n % 4096 else [])
    run_both(cfg, iq, chunks, 4096)


def test_sps_ge_8_tail_quirk_is_chunk_faithful():
    """48 kHz / 1200 baud / decim 1 => sps 40: the reference re-emits a symbol at chunk edges (SURVEY finding 3)."""
    iq = siggen.gmsk_channel(5, 40000, fs=48000, baud=1200)
    cfg = (48000, 1200, 5000, 1, 2000, True)
    run_both(cfg, iq, [4096] * 9, 4096)
    run_both(cfg, iq, [1000] * 30, 4096)


def test_nan_and_inf_inputs_no_dc():
    rng = np.random.default_rng(11)
    iq = siggen.gmsk_channel(7, 20000)
    iq[5000] = np.nan
    iq[9000] = np.inf + 0j
    iq[9001] = 1e-41 + 1e-42j
    run_both((48000, 9600, 5000, 1, 2000, False), iq, [4096] * 4, 4096)


def test_inf_sample_followed_by_many_clean_calls():
    """An Inf (not NaN) sample makes omega and mu NaN a few symbols later; the calls after it carry no flagged sample any
    more, but the loop state is still not finite: the channel has to stay on the general (NaN-aware) path until it is reset
    (the finite-only loop assumes every symbol advances by at least one sample).  Same stream as the oracle throughout."""
    iq = siggen.gmsk_channel(9, 10 * 4096)
    iq[3000] = np.inf + 0j
    run_both((48000, 9600, 5000, 1, 2000, False), iq, [4096] * 10, 4096)
    run_both((48000, 9600, 5000, 1, 2000, True), iq, [4096] * 10, 4096)


def test_batch_of_mixed_channels_is_independent():
    cfgs = [(48000, 9600, 5000, 1, 2000, True, 8192), (48000, 4800, 5000, 2, 2000, False, 8192),
            (240000, 19200, 5000, 5, 2000, True, 8192)] * 23  # 69 channels: two K3 waves, one partial
    sigs = [siggen.gmsk_channel(i, 9000, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    e = emu_api.EmuBatch(cfgs)
    assert e.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs]
    for lo, hi in [(0, 5000), (5000, 5003), (5003, 9000)]:
        lens = [(hi - lo) if i % 7 else max(0, hi - lo - 17) for i in range(len(cfgs))]  # ragged per channel
        parts = [s[lo:lo + n] for s, n in zip(sigs, lens)]
        e8, ef = e.process(parts)
        for i, o in enumerate(oracles):
            o8, of = o.process(parts[i])
            assert np.array_equal(o8, e8[i]), i
            assert np.array_equal(of.view(np.uint32), ef[i].view(np.uint32)), i


def test_oversize_input_is_dropped(capfd):
    e = emu_api.EmuBatch([(48000, 9600, 5000, 1, 2000, True, 100)])
    o = orc.Fsk(48000, 9600, 5000, 1, 2000, True, 100)
    iq = siggen.gmsk_channel(1, 300)
    for part in (iq[:100], iq[100:201], iq[201:300]):
        o8, of = o.process(part)
        e8, ef = e.process([part])
        assert np.array_equal(o8, e8[0])
    assert "more than max: 100" in capfd.readouterr().err


def test_create_errors_match_reference():
    assert emu_api.EmuBatch([(48000, 48000, 5000, 1, 2000, True, 4096)]).code == -1  # cutoff > fs/2
    assert emu_api.EmuBatch([(0, 4800, 5000, 1, 2000, True, 4096)]).code == -1
    assert emu_api.EmuBatch([(48000, 4800, 5000, 1, 0, True, 4096)]).code == -1


# ---------------------------------------------------------------- next row f-1: Doppler planner + NCO in front of the path

import json  # noqa: E402

DOPPLER = json.load(open(os.path.join(GOLDEN, "doppler_shifts_lucky7.json")))


def test_doppler_planner_matches_oracle_batching():
    lens = [2000] * 30 + [47000, 1, 95000, 48000, 7]
    ours = emu_api.doppler_plan_stream(48000, DOPPLER["shifts_hz"], lens)
    d = orc.Doppler(48000, DOPPLER["shifts_hz"], 100000)
    for n, got in zip(lens, ours):
        assert got == d.plan(n), n


def test_nco_then_demod_matches_oracle_doppler_then_demod():
    """lucky7.cf32 is the uncorrected recording of the reference's Doppler test: correct + demodulate."""
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.cf32"), dtype=np.complex64)
    cfg = (48000, 4800, 5000, 2, 2000, True)
    chunk = 20000
    e = emu_api.EmuBatch([cfg + (chunk,)])
    o = orc.Fsk(*cfg, chunk)
    d = orc.Doppler(48000, DOPPLER["shifts_hz"], chunk)
    calls = [chunk] * 4 + [len(iq) - 4 * chunk]
    plans = emu_api.doppler_plan_stream(48000, DOPPLER["shifts_hz"], calls)
    pos = 0
    for n, plan in zip(calls, plans):
        part = iq[pos:pos + n]
        pos += n
        mixed = d.process(part.view(np.float32))
        o8, of = o.process(mixed)
        e8, ef = e.process([part], [(0, ln, f) for ln, f in plan])
        assert np.array_equal(e.mixed(0).view(np.uint32), mixed.view(np.uint32))
        assert np.array_equal(of.view(np.uint32), ef[0].view(np.uint32))
        assert np.array_equal(o8, e8[0])


def test_channel_reassignment_matches_a_fresh_demodulator():
    """sdrm_batch_reset_channel's planning and state reset (shared host code, kernel emulation): a channel that has
    streamed with one configuration is given another (shorter filters, other rates, DC blocker off / on) and then
    behaves like a freshly created demodulator; its neighbours keep streaming undisturbed; what does not fit the
    batch's geometry is refused."""
    big = (48000, 4800, 5000, 2, 2000, True, 4096)       # 157-tap LPF1: the batch's largest filters
    cfgs = [big, (48000, 9600, 5000, 1, 2000, True, 4096), big]
    e = emu_api.EmuBatch(cfgs)
    assert e.code == 0
    sigs = [siggen.gmsk_channel(40 + i, 3 * 4096, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    keep = [orc.Fsk(*c) for c in cfgs]
    got, _ = e.process([s[:4096] for s in sigs])
    for i in range(3):
        assert np.array_equal(got[i], keep[i].process(sigs[i][:4096])[0])
    new_cfg = (48000, 9600, 5000, 1, 2000, False, 4096)
    assert e.reset_channel(1, new_cfg) == 0
    assert e.reset_channel(2) == 0                        # same configuration, new stream
    assert e.reset_channel(0, (48000, 4800, 5000, 2, 2000, True, 8192)) != 0  # a longer buffer than the batch's: refused, channel 0 untouched
    fresh = {1: orc.Fsk(*new_cfg), 2: orc.Fsk(*big)}
    new_sig = {1: siggen.gmsk_channel(77, 2 * 4096, fs=48000, baud=9600), 2: siggen.gmsk_channel(78, 2 * 4096, fs=48000, baud=4800)}
    for k in range(2):
        parts = [sigs[0][(k + 1) * 4096:(k + 2) * 4096], new_sig[1][k * 4096:(k + 1) * 4096], new_sig[2][k * 4096:(k + 1) * 4096]]
        got, _ = e.process(parts)
        assert np.array_equal(got[0], keep[0].process(parts[0])[0])
        assert np.array_equal(got[1], fresh[1].process(parts[1])[0])
        assert np.array_equal(got[2], fresh[2].process(parts[2])[0])


def test_absent_channel_keeps_its_state_while_an_empty_call_does_not():
    """A round of the batcher may run without some client's buffer.  That channel is ABSENT from the call
    (SDRM_LEN_ABSENT): no output, state untouched -- its stream stays the one the client put.  An EMPTY call is something
    else: the reference's clock stage answers it from the samples it carries and, with 8 or more samples per symbol,
    re-emits a symbol (clock_recovery_mm.c:127-135), so interleaving empty calls changes the stream (and the oracle fed
    the same empty calls agrees with that)."""
    cfg = (96000, 1200, 5000, 5, 4000, False)   # 16 samples per symbol
    other = (48000, 9600, 5000, 1, 2000, True)
    sig = siggen.gmsk_channel(9, 14000, fs=96000, baud=1200)
    sig2 = siggen.gmsk_channel(10, 14000)
    sizes = [3000, 3000, 4096, 500, 1, 3000]
    o = orc.Fsk(*cfg, 4096)
    want = []
    p = 0
    for n in sizes:
        want.append(o.process(sig[p:p + n])[0])
        p += n
    # the same buffers with calls in between in which the channel is absent (the other channel goes on)
    e = emu_api.EmuBatch([cfg + (4096,), other + (4096,)])
    got, p, q = [], 0, 0
    for n in sizes:
        e.process([emu_api.ABSENT, sig2[q:q + 700]])
        q += 700
        got.append(e.process([sig[p:p + n], emu_api.ABSENT])[0][0])
        p += n
        r8, _ = e.process([emu_api.ABSENT, emu_api.ABSENT])
        assert len(r8[0]) == 0 and len(r8[1]) == 0
    for k in range(len(sizes)):
        assert np.array_equal(got[k], want[k]), k
    # empty calls in between are NOT neutral at 16 samples per symbol -- and the oracle says the same
    e2 = emu_api.EmuBatch([cfg + (4096,)])
    o2 = orc.Fsk(*cfg, 4096)
    p, extra = 0, 0
    for n in sizes:
        a = e2.process([sig[p:p + n]])[0][0]
        b = o2.process(sig[p:p + n])[0]
        assert np.array_equal(a, b)
        p += n
        a = e2.process([None])[0][0]
        b = o2.process(sig[0:0])[0]
        assert np.array_equal(a, b)
        extra += len(a)
    assert extra > 0


def test_boxcar_quotient_short_form_is_the_ieee_division():
    """sdrm_boxcar_out_fast (csrc/sdrm_core.h): q0 = a * RN(1/L), one FMA for the residual, one for the correction -- must be
    a / L bit for bit whenever it does not ask for the division proper (reference src/dsp/dc_blocker.c:63).  Every
    significand, both signs, at a normal exponent for the lengths of the named configurations; at the exponents where the
    quotient turns denormal the flag must be up instead.  (tools/dc_div_sweep.cpp: all lengths 32..3968, 74e9 quotients.)"""
    import ctypes as C
    lib = emu_api.lib()
    lib.emu_check_boxcar_div.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.emu_check_boxcar_div.restype = C.c_uint64
    for length in (80, 154, 160, 400, 800, 1280):
        uns = C.c_uint64()
        assert lib.emu_check_boxcar_div(length, 127, C.byref(uns)) == 0, length
        assert uns.value == 0
        assert lib.emu_check_boxcar_div(length, 1, C.byref(uns)) == 0, length   # quotients below the normal range
        assert uns.value == 2 * (1 << 23)
        assert lib.emu_check_boxcar_div(length, 255, C.byref(uns)) == 0          # infinities and NaN
        assert uns.value == 2 * (1 << 23)


def test_nco_wrap_without_a_compare_returns_the_reference_bits():
    """sdrm_nco_advance_nomask (csrc/sdrm_core.h: add, fma-with-clamp, fma -- the phase kernel's three instructions per
    sample) against the reference's two-test wrap (src/dsp/sig_source.c:47-53) for |phase|, |step| <= 2 pi: random pairs,
    every float within 64 ulp of the boundaries +-2 pi reached from both sides, zeros of both signs, denormals, and the
    recursion itself over long runs at Doppler-sized and at extreme steps."""
    import ctypes as C
    lib = emu_api.lib()
    lib.emu_check_nco_advance.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.emu_check_nco_advance.restype = C.c_uint64
    lib.emu_check_nco_run.argtypes = [C.c_float, C.c_float, C.c_size_t]
    lib.emu_check_nco_run.restype = C.c_size_t
    two_pi = np.float32(6.28318530717958647692)
    rng = np.random.default_rng(11)
    ph = [rng.uniform(-two_pi, two_pi, 1 << 20).astype(np.float32)]
    st = [rng.uniform(-two_pi, two_pi, 1 << 20).astype(np.float32)]
    # sums that land within a few ulp of +-2 pi: phase = +-2 pi - step + k ulp
    steps = np.concatenate([rng.uniform(-two_pi, two_pi, 4096), 10.0 ** rng.uniform(-8, 0.79, 4096) * rng.choice([-1, 1], 4096)]).astype(np.float32)
    for sign in (-1.0, 1.0):
        edge = np.float32(sign) * two_pi
        for k in range(-64, 65):
            target = (edge.view(np.int32) + np.int32(k)).view(np.float32)   # k ulp beyond / inside the boundary
            p = (target - steps).astype(np.float32)
            keep = np.abs(p) <= two_pi
            ph.append(p[keep])
            st.append(steps[keep])
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, two_pi, -two_pi, np.nextafter(two_pi, np.float32(0)),
                        -np.nextafter(two_pi, np.float32(0)), 3.1415927, -3.1415927], dtype=np.float32)
    gp, gs = np.meshgrid(special, special)
    ph.append(gp.ravel())
    st.append(gs.ravel())
    p = np.ascontiguousarray(np.concatenate(ph))
    s_ = np.ascontiguousarray(np.concatenate(st))
    assert p.size == s_.size and p.size > 2_000_000
    assert lib.emu_check_nco_advance(p.ctypes.data, s_.ctypes.data, p.size) == 0
    for phase0, step in ((0.0, 2e-4), (1.0, -0.13089969), (-6.2831855, 6.2831855), (6.2831855, -6.2831855), (0.5, 3.1415927),
                         (0.0, -1e-7), (-0.0, -0.0), (2.0, 0.0), (6.2831855, 4.7e-7), (-6.2831855, -4.7e-7)):
        assert lib.emu_check_nco_run(phase0, step, 3_000_000) == 3_000_000, (phase0, step)


def test_nco_sample_is_the_correctly_rounded_float_of_the_exact_cosine():
    """sdrm_nco_sample (csrc/sdrm_core.h; reference src/dsp/sig_source.c:46): where (float) cos((double) phase) hangs on the
    last bits of the double, the sample is re-evaluated in double-double and rounded once.  Here on the host: the
    double-double sin/cos against 300-bit arithmetic, and a sweep of 2^25 consecutive phases -- every one equal to the
    host libm's float, the fragile ones checked to be the correctly rounded float of the exact value."""
    import ctypes as C
    import mpmath as mp
    mp.mp.prec = 300
    lib = emu_api.lib()
    lib.emu_sincos_dd.argtypes = [C.c_double, C.POINTER(C.c_double)]
    lib.emu_nco_sweep.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32]
    lib.emu_nco_sample.argtypes = [C.c_float, C.POINTER(C.c_float)]
    rng = np.random.default_rng(3)
    xs = list(rng.uniform(-7, 7, 300).astype(np.float32)) + [1e-30, -1e-20, np.float32(np.pi / 2), np.float32(np.pi), 6.2831855,
                                                             -6.2831855, 1000.5, -54321.25, 1048575.0]
    for x in xs:
        out = (C.c_double * 4)()
        lib.emu_sincos_dd(float(x), out)
        for got, true in ((mp.mpf(out[0]) + mp.mpf(out[1]), mp.sin(mp.mpf(float(x)))), (mp.mpf(out[2]) + mp.mpf(out[3]), mp.cos(mp.mpf(float(x))))):
            assert abs(got - true) <= abs(true) * mp.mpf(2) ** -100, (float(x), float(got))
    fragile_total, phases = 0, []
    for first in (0x3f000000, 0x40490fdb - (1 << 22), 0xc0000000, 0x40c00000):  # around 0.5, pi, -2, 6
        fr, df = C.c_uint64(), C.c_uint64()
        buf = np.zeros(256, np.float32)
        lib.emu_nco_sweep(first, 1 << 23, 1, C.byref(fr), C.byref(df), buf.ctypes.data, 256)
        assert df.value == 0  # the host libm's float everywhere
        fragile_total += fr.value
        phases += list(buf[:min(fr.value, 256)])
    assert fragile_total > 0  # the sweep did meet roundings that hang on the double's last bits (2^-29 x 33 per value)
    for ph in phases:
        out = (C.c_float * 2)()
        lib.emu_nco_sample(float(ph), out)
        for got, fn in ((out[0], mp.cos), (out[1], mp.sin)):
            true = fn(mp.mpf(float(ph)))
            f = np.float32(float(true))
            best = min([np.nextafter(f, np.float32(-np.inf)), f, np.nextafter(f, np.float32(np.inf))], key=lambda v: abs(mp.mpf(float(v)) - true))
            assert np.float32(got) == best, float(ph)



def test_batch_geometry_grows_with_a_new_clients_configuration():
    """plan_growth / apply_growth (shared with the device path's grow_geometry): a batch of small no-DC channels takes, slot
    by slot, the first DC blocker, a 240 kHz client with 397 / 289-tap filters and a 400-sample boxcar, a 1200-baud client
    with a 1280-sample boxcar; the channels that keep their clients continue bit for bit through every re-layout of the raw
    histories, the DC states and the private tap slots; a longer buffer than the batch's is refused."""
    n = 3000
    base = (48000, 9600, 5000, 1, 2000, False, n)
    e = emu_api.EmuBatch([base] * 4)
    assert e.code == 0
    sigs = [siggen.gmsk_channel(700 + i, 10 * n, fs=48000, baud=9600) for i in range(4)]
    orcs = [orc.Fsk(*base) for _ in range(4)]
    pos = [0] * 4

    def calls(k):
        for _ in range(k):
            lens = [n, n - 17, 1000, n]
            parts = [s[p:p + ln] for s, p, ln in zip(sigs, pos, lens)]
            for i, ln in enumerate(lens):
                pos[i] += ln
            e8, ef = e.process(parts)
            for i in range(4):
                o8, of = orcs[i].process(parts[i])
                assert np.array_equal(o8, e8[i]) and np.array_equal(of.view(np.uint32), ef[i].view(np.uint32)), (i, pos[i])

    def hand_over(ch, cfg, seed):
        assert e.reset_channel(ch, cfg) == 0, cfg
        orcs[ch] = orc.Fsk(*cfg)
        sigs[ch] = siggen.gmsk_channel(seed, 10 * n, fs=cfg[0], baud=cfg[1])
        pos[ch] = 0

    calls(2)
    hand_over(1, (48000, 9600, 5000, 1, 2000, True, n), 801)
    calls(1)
    hand_over(3, (240000, 19200, 5000, 1, 2000, True, n), 803)
    calls(2)
    hand_over(0, (48000, 1200, 5000, 1, 2000, True, n), 800)
    hand_over(3, base, 813)
    calls(2)
    assert e.reset_channel(2, (48000, 9600, 5000, 1, 2000, True, 2 * n)) != 0
    calls(1)


def test_few_live_channels_in_a_large_batch_keep_their_streams():
    """A server's batcher is sized for its busiest hour: most slots have no buffer in a round (ABSENT).  Since round 5 an empty slot
    of a DC workgroup is a REPLICA of the group's longest live slot (it computes the same and saves nothing: sdrm_k2_fill_aliases),
    so that the live channels keep the stage's straight-line code.  20 channels of three kinds, presence changing from call to call
    (a channel that sits out a call continues where it left off), lengths ragged: every live channel's stream equals the oracle's."""
    kinds = [(48000, 9600, 5000, 1, 2000, True), (240000, 19200, 5000, 5, 2000, True), (48000, 1200, 5000, 8, 2000, True),
             (48000, 9600, 5000, 1, 2000, False)]
    maxlen = 6000
    cfgs = [kinds[i % 4] + (maxlen,) for i in range(20)]
    e = emu_api.EmuBatch(cfgs)
    assert e.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs]
    sigs = [siggen.gmsk_channel(900 + i, 8 * maxlen, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    pos = [0] * 20
    rng = np.random.default_rng(17)
    patterns = [[0], [0, 1, 2], [17], [3, 4, 19], list(range(20)), [16, 18], [5], [0, 19], [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15]]
    for live in patterns:
        parts = []
        for i in range(20):
            if i in live:
                n = int(rng.choice([maxlen, maxlen, 4097, 300, 0]))
                parts.append(sigs[i][pos[i]:pos[i] + n])
                pos[i] += n
            else:
                parts.append(emu_api.ABSENT)
        e8, ef = e.process(parts)
        for i in range(20):
            if i in live:
                o8, of = oracles[i].process(parts[i])
                assert np.array_equal(o8, e8[i]) and np.array_equal(of.view(np.uint32), ef[i].view(np.uint32)), (i, live)
            else:
                assert len(e8[i]) == 0
