"""`-m gpu`: the HIP path, called through the C-ABI (libsdrmodem_hip.so), against the CPU oracle.

Bar (BASELINE.json north_star): float soft bits within 1e-4 RMS of the CPU reference.  The exact-mode kernels are
built to be BIT-IDENTICAL (tolerance 0 on the fp32 soft bits and on the int8 output), which is what these tests
assert; the RMS figure is checked as well so the stated tolerance is visible in the test."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import orc
import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd import binding, siggen  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RMS_TOL = 1e-4  # north_star tolerance on the float soft bits


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    L = binding.load()
    assert L.sdrm_device_count() > 0, "these tests need an MI355X; the library has no CPU path"


def assert_same(of, gf, o8, g8, where=""):
    assert len(of) == len(gf) and len(o8) == len(g8), (where, len(of), len(gf))
    if len(of):
        rms = float(np.sqrt(np.mean((of.astype(np.float64) - gf.astype(np.float64)) ** 2)))
        assert rms <= RMS_TOL, (where, rms)
    assert np.array_equal(of.view(np.uint32), gf.view(np.uint32)), (where, "float soft bits differ")
    assert np.array_equal(o8, g8), (where, "int8 soft bits differ")


def run_stream(cfg, iq, chunks, maxlen):
    o = orc.Fsk(*cfg, maxlen)
    g = binding.Batch([cfg + (maxlen,)], keep_soft=True)
    assert o.code == 0 and g.code == 0
    pos = 0
    total = 0
    for n in chunks:
        part = iq[pos:pos + n]
        pos += n
        o8, of = o.process(part)
        g8 = g.process([part])[0]
        gf = g.last_soft(0)
        assert_same(of, gf, o8, g8, where="pos %d" % pos)
        total += len(o8)
    g.close()
    return total


# ---------------------------------------------------------------- stage probes

def test_probe_fast_atan2_bit_exact():
    """the branch-free arctangent of the front-end kernel (sdrm_fast_atan2f_flat) on the device: every octant, axes,
    signed zeros, denormals, the table knots, the double-typed threshold, Inf/NaN -- bit-identical to the oracle"""
    from test_kernel_logic_cpu import atan_cases
    y, x = atan_cases()
    out = np.zeros(len(y), np.float32)
    assert binding.load().sdrm_probe_atan2(y.ctypes.data, x.ctypes.data, out.ctypes.data, len(y)) == 0
    f = orc.lib().orc_fast_atan2f
    want = np.array([f(a, b) for a, b in zip(y, x)], dtype=np.float32)
    both_nan = np.isnan(out) & np.isnan(want)
    assert np.array_equal(out.view(np.uint32)[~both_nan], want.view(np.uint32)[~both_nan])
    assert np.array_equal(np.isnan(out), np.isnan(want))


def test_probe_discriminator_short_form_and_its_fall_back():
    """The front-end evaluates x[n] conj(x[n-1]) -> fast_atan2f -> gain (reference src/dsp/quadrature_demod.c:57-73) in a
    short form -- the division as the seven fused operations of the IEEE expansion without its range scaling, packed two
    samples per instruction -- whenever every operand of a wave lies in [2^-60, 2^60], and in the general form otherwise.
    Streams over thirty decades of amplitude (both forms get whole waves), amplitudes at the range's two ends, real-valued
    input (every sample on an axis: a zero numerator), zeros, signed zeros, denormals, Inf and NaN: the oracle's floats,
    bit for bit, and both forms really taken."""
    L = binding.load()
    rng = np.random.default_rng(2026)
    wave = 64 * 15
    parts = []
    for k in range(220):  # one amplitude per wave of 960 samples
        amp = np.float32(10.0 ** rng.uniform(-16, 16))
        parts.append(((rng.standard_normal(wave) + 1j * rng.standard_normal(wave)) * amp).astype(np.complex64))
    for e in (-31, -30, -29, 29, 30, 31):  # products next to 2^-60 and 2^60
        parts.append(((rng.standard_normal(wave) + 1j * rng.standard_normal(wave)) * np.float32(2.0 ** e)).astype(np.complex64))
    ramp = np.zeros(wave, np.complex64)
    ramp.real = (np.arange(wave) % 256).astype(np.float32) - 77.0   # the reference's perf input, shifted through zero
    parts.append(ramp)
    odd = (rng.standard_normal(wave) + 1j * rng.standard_normal(wave)).astype(np.complex64)
    odd[5] = 0
    odd[6] = complex(-0.0, 0.0)
    odd[7] = complex(0.0, -0.0)
    odd[100] = complex(np.nan, 1.0)
    odd[200] = complex(1.0, np.inf)
    odd[300] = complex(1e-41, -1e-42)
    odd[301] = complex(-1e-39, 1e-45)
    odd[400] = complex(3e38, -3e38)
    parts.append(odd)
    parts.append((rng.standard_normal(wave) + 1j * rng.standard_normal(wave)).astype(np.complex64))  # a clean wave behind it
    y = np.concatenate(parts)
    n = len(y)
    gain = np.float32(1.5278874)
    out = np.zeros(n, np.float32)
    fast = np.zeros((n + wave - 1) // wave, np.uint32)
    assert L.sdrm_probe_quad(y.view(np.float32).ctypes.data, n, gain, out.ctypes.data, fast.ctypes.data) == 0
    with np.errstate(all="ignore"):
        want = orc.Quad(float(gain), n).process(y.view(np.float32))
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(out), nan)
    assert np.array_equal(out.view(np.uint32)[~nan], want.view(np.uint32)[~nan])
    # amplitudes 1e-8 .. 1e8 have every |re|, |im| product inside the range (barring a sample on an axis): short form;
    # the real-valued wave, the odd one and the extreme decades: general form
    assert 60 < int(fast.sum()) < len(fast) - 20, (int(fast.sum()), len(fast))
    assert fast[-1] == 1 and fast[-2] == 0 and fast[-3] == 0


@pytest.mark.parametrize("length", [32, 80, 154, 160, 400, 1280, 3968])
def test_probe_boxcar_quotient_is_the_ieee_division(length):
    """the DC blocker's `sum / L` (reference src/dsp/dc_blocker.c:63) as the kernel computes it: reciprocal multiply + two
    FMAs, the division proper when the quotient is denormal or not finite -- the IEEE quotient bit for bit"""
    L = binding.load()
    rng = np.random.default_rng(length)
    n = 1 << 20
    a = (rng.standard_normal(n) * 10.0 ** rng.uniform(-44, 38, n)).astype(np.float32)
    edge = np.array([0.0, -0.0, 1e-45, -1e-45, 1.1754944e-38, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan,
                     length, 0.5 * length * 1.4e-45, 1.5 * length * 1.4e-45, 2.5 * length * 1.4e-45], dtype=np.float32)
    a[:len(edge)] = edge
    a[len(edge):len(edge) + 4096] = (np.arange(4096) * length * np.float32(1.4e-45) * 0.5).astype(np.float32)  # ties among denormals
    out = np.zeros(n, np.float32)
    assert L.sdrm_probe_boxcar_div(a.ctypes.data, length, out.ctypes.data, n) == 0
    with np.errstate(all="ignore"):
        want = (a / np.float32(length)).astype(np.float32)
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(out), nan)
    assert np.array_equal(out.view(np.uint32)[~nan], want.view(np.uint32)[~nan])


def test_design_parameters_match_oracle():
    cfg = (48000, 9600, 5000, 1, 2000, True)
    g = binding.Batch([cfg + (4096,)])
    o = orc.Fsk(*cfg, 4096)
    inf, t1, t2 = o.info()
    ginf = g.info(0)
    for f in ("taps1_len", "taps2_len", "dc_length", "quad_gain", "sps", "gain_omega", "gain_mu", "omega_lim"):
        assert getattr(inf, f) == getattr(ginf, f), f
    assert np.array_equal(t1.view(np.uint32), g.taps(0, 1).view(np.uint32))
    assert np.array_equal(t2.view(np.uint32), g.taps(0, 2).view(np.uint32))
    g.close()


@pytest.mark.parametrize("lanes", ["16", "32", "64", "16x512", "16x256", "32x256", "64x256p", "32x256p"])
def test_clock_stage_workgroup_shapes(lanes, monkeypatch):
    """The clock stage's workgroup shape is channels x ring length, with pair-element or plain ("p") rings: 16 x 1024 up to
    1280 channels (int8 conversion in the staging wave), 32 x 512 up to 2560, 64 x 256 plain beyond; SDRM_K3_LANES forces
    one.  Every shape, both builds (float soft bits kept / int8 only), ragged chunks of one stream and a ragged 69-channel
    mixed batch with NaN input."""
    monkeypatch.setenv("SDRM_K3_LANES", lanes)
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    chunks = [0, 1, 7, 100, 255, 256, 257, 3839, 5000, 9000, 1, 8191, 12000, 64, 63, 65, 1]
    assert run_stream((48000, 9600, 5000, 1, 2000, True), iq, chunks, 12000) > 0
    cfgs = [(48000, 9600, 5000, 1, 2000, True, 8192), (48000, 4800, 5000, 2, 2000, False, 8192),
            (240000, 19200, 5000, 5, 2000, True, 8192)] * 23
    sigs = [siggen.gmsk_channel(i, 9000, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    sigs[4][6000] = np.nan
    oracles = [orc.Fsk(*c) for c in cfgs]
    want = []
    spans = [(0, 5000), (5000, 5003), (5003, 9000)]
    for lo, hi in spans:
        lens = [(hi - lo) if i % 7 else max(0, hi - lo - 17) for i in range(len(cfgs))]
        want.append([o.process(s[lo:lo + n]) for o, s, n in zip(oracles, sigs, lens)])
    for keep_soft in (True, False):
        g = binding.Batch(cfgs, keep_soft=keep_soft)
        assert g.code == 0
        for (lo, hi), ref in zip(spans, want):
            lens = [(hi - lo) if i % 7 else max(0, hi - lo - 17) for i in range(len(cfgs))]
            g8 = g.process([s[lo:lo + n] for s, n in zip(sigs, lens)])
            for i, (o8, of) in enumerate(ref):
                assert np.array_equal(o8, g8[i]), (lanes, keep_soft, lo, i)
                if keep_soft:
                    gf = g.last_soft(i)
                    both_nan = np.isnan(of) & np.isnan(gf)
                    assert np.array_equal(of.view(np.uint32)[~both_nan], gf.view(np.uint32)[~both_nan]), (lanes, lo, i)
        g.close()


# ---------------------------------------------------------------- the reference's own fixtures (test/test_fsk_demod.c)

E2E = [
    ("lucky7", (48000, 4800, 5000, 2, 2000, True), "lucky7.expected.cf32", "lucky7.expected.s8"),
    ("lucky7_nodc", (48000, 4800, 5000, 2, 2000, False), "lucky7.expected.cf32", "lucky7.expected.nodc.s8"),
    ("nusat", (192000, 40000, 5000, 1, 2000, True), "nusat.cf32", "processed.s8"),
    ("nan", (240000, 9600, 5000, 1, 2000, True), "inputnan.cf32", "nan.s8"),
]


@pytest.mark.parametrize("name,cfg,inp,exp", E2E, ids=[e[0] for e in E2E])
def test_reference_fixtures_through_fsk_demod_api(name, cfg, inp, exp):
    """Reads like test/test_fsk_demod.c:22-50: 4096-sample buffers through fsk_demod_process, +-2 LSB vs the golden
    file (the reference's tolerance) -- and, stronger, identical to the oracle."""
    iq = np.fromfile(os.path.join(GOLDEN, inp), dtype=np.complex64)
    want = np.fromfile(os.path.join(GOLDEN, exp), dtype=np.int8)
    d = binding.FskDemod(*cfg, 4096)
    assert d.code == 0
    got = np.concatenate([d.process(iq[o:o + 4096]) for o in range(0, len(iq), 4096)])
    d.close()
    assert len(got) == len(want)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 2
    oracle8, _ = orc.demod_stream(cfg, iq, 4096)
    assert np.array_equal(got, oracle8)


# ---------------------------------------------------------------- config 2 of BASELINE.json + chunking edge cases

def test_single_channel_48k_9600_vs_cpu_soft_bits():
    iq = siggen.gmsk_channel(0, 4 * 131072)
    n = run_stream((48000, 9600, 5000, 1, 2000, True), iq, [131072] * 4, 131072)
    assert n > 4 * 26000


def test_lucky7_float_soft_bits_all_configs():
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    for cfg in [(48000, 9600, 5000, 1, 2000, True), (48000, 9600, 5000, 1, 2000, False),
                (48000, 4800, 5000, 2, 2000, True), (48000, 4800, 5000, 3, 2000, True)]:
        run_stream(cfg, iq, [4096] * 23 + [96000 - 23 * 4096], 4096)


def test_fast_fma_flag_is_refused():
    """SDRM_FLAG_FAST_FMA (rounds 2-5: fused multiply-adds in both filters) is gone: it failed the reference's own +-2 LSB
    tolerance (test/test_fsk_demod.c:47) on lucky7 without DC blocker -- 19 LSB, two hard-bit flips -- so no caller could
    ship it.  The flag's value stays reserved and sdrm_batch_create answers -ENOTSUP; the exact mode is the only mode."""
    import errno
    cfgs = binding.make_configs([(48000, 9600, 5000, 1, 2000, True, 4096)])
    h = binding.C.c_void_p()
    assert binding.load().sdrm_batch_create(cfgs, 1, -1, 2, binding.C.byref(h)) == -errno.ENOTSUP
    assert not h


def test_ragged_chunks():
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    chunks = [0, 1, 1, 1, 7, 0, 100, 3839, 3840, 3841, 5000, 9000, 1, 2, 8191, 12000, 64, 63, 65, 1]
    for cfg in [(48000, 9600, 5000, 1, 2000, True), (48000, 4800, 5000, 2, 2000, True)]:
        run_stream(cfg, iq, chunks, 12000)


def test_mixed_rate_configs():
    for cfg, fs, baud in [((240000, 19200, 5000, 5, 2000, True), 240000, 19200),
                          ((48000, 1200, 5000, 8, 2000, True), 48000, 1200),
                          ((48000, 1200, 5000, 1, 2000, True), 48000, 1200),  # sps 40: tail quirk, chunk-faithful
                          ((240000, 19200, 5000, 1, 2000, True), 240000, 19200),
                          ((240000, 1200, 5000, 1, 2000, True), 240000, 1200)]:  # sps 200, DC length 6400, LPF1 of 1033 taps
        iq = siggen.gmsk_channel(3, 40000, fs=fs, baud=baud)
        run_stream(cfg, iq, [16384, 5, 16384, 7227], 16384)


def test_nan_inf_denormal_inputs():
    iq = siggen.gmsk_channel(7, 20000)
    iq[5000] = np.nan
    iq[9000] = np.inf + 0j
    iq[9001] = 1e-41 + 1e-42j
    iq[12000:12100] *= np.float32(1e-38)  # denormal region
    run_stream((48000, 9600, 5000, 1, 2000, False), iq, [4096] * 4, 4096)
    run_stream((48000, 9600, 5000, 1, 2000, True), iq[10000:], [4096, 4096], 4096)


def test_inf_sample_followed_by_many_clean_calls():
    """the loop state stays non-finite after an Inf sample: the device keeps the channel on its general path (sticky
    poison) and follows the oracle through the clean calls behind it"""
    iq = siggen.gmsk_channel(9, 10 * 4096)
    iq[3000] = np.inf + 0j
    run_stream((48000, 9600, 5000, 1, 2000, False), iq, [4096] * 10, 4096)
    run_stream((48000, 9600, 5000, 1, 2000, True), iq, [4096] * 10, 4096)


def test_oversize_input_gives_no_output(capfd):
    d = binding.FskDemod(48000, 9600, 5000, 1, 2000, True, 100)
    assert len(d.process(siggen.gmsk_channel(1, 101))) == 0
    d.close()
    assert "<3>requested buffer 101 is more than max: 100" in capfd.readouterr().err


# ---------------------------------------------------------------- batches (configs 3 and 5 shapes, scaled to seconds)

def test_batch_256_channels_bit_exact_and_independent():
    C_, N = 256, 16384
    cfg = (48000, 9600, 5000, 1, 2000, True)
    sig = siggen.gmsk_batch(C_, 2 * N)
    g = binding.Batch([cfg + (N,)] * C_, keep_soft=True)
    assert g.code == 0
    outs = [g.process([sig[c, k * N:(k + 1) * N] for c in range(C_)]) for k in range(2)]
    for c in list(range(0, C_, 37)) + [63, 64, 255]:
        o8a, _ = orc.demod_stream(cfg, sig[c], N)
        got = np.concatenate([outs[0][c], outs[1][c]])
        assert np.array_equal(got, o8a), c
    # size-independent property: every channel produced ~N*baud/fs symbols
    lens = np.array([len(outs[1][c]) for c in range(C_)])
    assert np.all(np.abs(lens - N / 5) < 40)
    g.close()


@pytest.mark.parametrize("channels", [512, 1100, 2100, 2700])
def test_many_channel_batches_bit_exact(channels):
    """BASELINE configs[3] gives every GPU 512 channels; 1100 is not a multiple of any workgroup size, with a second
    configuration mixed in (different filters, decimation and DC length inside one DC workgroup); 2100 takes the 32-channel
    clock-stage workgroups (k3_quantize behind them), 2700 the 64-channel ones with the plain ring.  Three calls (full,
    short, ragged) so that every stage's state crosses call boundaries; 24 spot channels against the oracle, float
    and int8 soft bits bit for bit; every other channel against its twin (same waveform: same bits)."""
    N = 8192
    cfg_a = (48000, 9600, 5000, 1, 2000, True)
    cfg_b = (48000, 4800, 5000, 2, 2000, True)
    distinct = 24
    cfgs = [(cfg_b if (c % distinct) % 5 == 3 else cfg_a) + (N,) for c in range(channels)]
    base = [siggen.gmsk_channel(100 + i, 2 * N + 700, fs=cfgs[i][0], baud=cfgs[i][1]) for i in range(distinct)]
    g = binding.Batch(cfgs, keep_soft=True)
    assert g.code == 0
    oracles = [orc.Fsk(*cfgs[i]) for i in range(distinct)]
    spots = list(range(channels - distinct, channels))  # the last 24 channels: partial workgroups of every stage
    for lo, n in ((0, N), (N, 700), (N + 700, N - 1)):
        lens = [n if c % 11 else max(0, n - 33) for c in range(channels)]
        parts = [base[c % distinct][lo:lo + lens[c]] for c in range(channels)]
        g8 = g.process(parts)
        want = {}
        for c in spots:
            o8, of = oracles[c % distinct].process(parts[c]) if (c % distinct, lens[c]) not in want else want[(c % distinct, lens[c])]
            want[(c % distinct, lens[c])] = (o8, of)
        for c in spots:
            o8, of = want[(c % distinct, lens[c])]
            assert_same(of, g.last_soft(c), o8, g8[c], where="%d channels, ch %d" % (channels, c))
        # twins: channels with the same waveform and the same lengths so far produce the same bits
        first = {}
        for c in range(channels):
            key = (c % distinct, c % 11 == 0)
            if key in first:
                assert np.array_equal(g8[c], g8[first[key]]), (channels, c, first[key])
            else:
                first[key] = c
    g.close()


_FANOUT_RANK = r"""
import os, sys
sys.path.insert(0, os.environ["SDRM_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SDRM_ROOT"], "tests"))
import numpy as np, torch, torch.distributed as dist
import sdrm_pkg; sdrm_pkg.load()
from sdr_modem_amd import binding, shard, siggen
import orc
rank = int(os.environ["RANK"]); backend = os.environ["SDRM_TEST_BACKEND"]
dev_index = rank % torch.cuda.device_count()     # gloo: the ranks share the box's devices
torch.cuda.set_device(dev_index)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index)); coll = torch.device("cuda", dev_index)
else:
    dist.init_process_group(backend); coll = "cpu"
# leg 1: equal-count shards of a 9-channel table (BASELINE configs[3] in miniature), every shard on the HIP path
table = [(48000, 9600, 5000, 1, 2000, True, 4096)] * 5 + [(48000, 4800, 5000, 2, 2000, False, 4096)] * 4 if rank == 0 else None
cfgs, lo, hi = shard.fanout_configs(table, 9, device=coll)
b = binding.Batch(cfgs, device=dev_index)
sigs = [siggen.gmsk_channel(lo + i, 4096, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
got = b.process(sigs)
ok = all(np.array_equal(orc.demod_stream(c[:6], s, 4096)[0], g) for c, s, g in zip(cfgs, sigs, got))
b.close()
# leg 2: BASELINE configs[4] in miniature -- cost-balanced shards of a mixed-rate table, the call's Doppler batches planned
# on rank 0 for the whole node and fanned out with the SAME partition; every rank corrects and demodulates its channels
n = 6000
mix = [(240000, 19200, 5000, 5, 2000, True, n)] * 4 + [(48000, 1200, 5000, 8, 2000, True, n)] * 12 if rank == 0 else None
part = shard.fanout_configs(mix, 16, device=coll, balance="cost")
segs0 = [(c, ln, f) for c in range(16) if c % 3 for ln, f in ((2500, 900 - 40 * c), (n - 2500, 905 - 40 * c))] if rank == 0 else None
mine = shard.fanout_nco_segments(segs0, part, device=coll)
b2 = binding.Batch(part.cfgs, device=dev_index)
sig2 = [siggen.gmsk_channel(50 + part.lo + i, n, fs=c[0], baud=c[1]) for i, c in enumerate(part.cfgs)]
got2 = b2.process_nco(sig2, mine)
ok2 = len(mine) == 2 * sum(1 for c in range(part.lo, part.hi) if c % 3)
for i, c in enumerate(part.cfgs):
    gc = part.lo + i
    x = sig2[i].view(np.float32)
    if gc % 3:
        osc = orc.Nco(1.0, c[0], n)
        x = np.concatenate([osc.multiply(900 - 40 * gc, x[:5000]), osc.multiply(905 - 40 * gc, x[5000:])])
    ok2 = ok2 and np.array_equal(orc.demod_stream(c[:6], x.view(np.complex64), n)[0], got2[i])
b2.close()
flags = [None, None]
dist.all_gather_object(flags, (lo, hi, bool(ok), part.lo, part.hi, bool(ok2)))
dist.barrier(); dist.destroy_process_group()
if rank == 0:
    assert flags[0][:2] == (0, 5) and flags[1][:2] == (5, 9) and flags[0][2] and flags[1][2], flags
    # the four 240 kHz channels weigh as much as the twelve 48 kHz ones: the cost cut is not the middle of the table
    assert flags[0][3] == 0 and flags[0][4] == flags[1][3] and flags[1][4] == 16 and flags[0][4] < 8, flags
    assert flags[0][5] and flags[1][5], flags
"""


def _two_ranks(code, extra_env):
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SDRM_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    return [p.wait(timeout=900) for p in procs]


def test_config_and_doppler_fanout_across_two_ranks_on_the_device():
    """the only collectives of the path are the broadcast of the channel table from rank 0 and, with Doppler
    correction, of the call's NCO batches.  Two ranks, each demodulating its shard on the HIP path against the oracle:
    equal-count shards (configs[3] shape) and cost-balanced shards of a mixed-rate table with the NCO fan-out
    (configs[4] shape).  Over RCCL (backend "nccl") when the node has two GPUs; on a one-GPU box the two ranks share
    cuda:0 and rendezvous over gloo -- the same rank code, spawn and collectives, a functional run."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    assert _two_ranks(_FANOUT_RANK, {"SDRM_TEST_BACKEND": backend}) == [0, 0]


def test_bench_with_two_ranks_runs_the_sharded_path(tmp_path):
    """`python bench.py --gpus 2` the way a user (or the driver, through torch.distributed.run) starts it: the parent
    spawns two fresh ranks before anything touches the GPU, rank 0 broadcasts the channel table, every rank runs its
    shard, the timing is the maximum over ranks, and config3_sharded / config5_sharded repeat that for BASELINE
    configs[3] and configs[4].  On a one-GPU box SDRM_BENCH_BACKEND=gloo lets the two ranks share cuda:0: functional,
    not a measurement."""
    import json
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    if torch.cuda.device_count() < 2:
        env["SDRM_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--no-cpu-baseline", "--sweep", "", "--channels-per-gpu", "64", "--chunks-resident", "2"],
                         env=env, stdout=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.returncode
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 4 and res["value"] > 0 and res["scaling"] == "weak"
    assert res["config3_sharded"]["channels_total"] == 1024 and res["config3_sharded"]["value"] > 0
    c5 = res["config5_sharded"]
    assert c5["channels_total"] == 512 and c5["value"] > 0 and c5["balance"] == "cost"
    assert sum(c5["channels_per_rank"]) == 512 and c5["channels_per_rank"][0] < 256  # the heavy channels come first
    (tmp_path / "bench_gloo2.json").write_text(lines[0])
    keep = os.environ.get("SDRM_KEEP_BENCH_LINE")  # tools/gpu_round3.sh keeps the line for profiles/
    if keep:
        with open(keep, "w") as f:
            f.write(lines[0] + "\n")


def test_mixed_batch_with_ragged_lengths():
    cfgs = [(48000, 9600, 5000, 1, 2000, True, 8192), (48000, 4800, 5000, 2, 2000, False, 8192),
            (240000, 19200, 5000, 5, 2000, True, 8192)] * 23
    sigs = [siggen.gmsk_channel(i, 9000, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    g = binding.Batch(cfgs, keep_soft=True)
    assert g.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs]
    for lo, hi in [(0, 5000), (5000, 5003), (5003, 9000)]:
        lens = [(hi - lo) if i % 7 else max(0, hi - lo - 17) for i in range(len(cfgs))]
        parts = [s[lo:lo + n] for s, n in zip(sigs, lens)]
        g8 = g.process(parts)
        for i, o in enumerate(oracles):
            o8, of = o.process(parts[i])
            assert_same(of, g.last_soft(i), o8, g8[i], where="ch %d" % i)
    g.close()


def test_default_build_of_the_clock_stage_on_ragged_mixed_streams():
    """Without KEEP_SOFT_F32 the clock stage runs its hand-scheduled symbol loop (int8 only): the same ragged,
    mixed-rate, multi-call streams as above, including sps >= 8 (tail quirk) and decimated channels, int8 bit-exact."""
    cfgs = [(48000, 9600, 5000, 1, 2000, True, 8192), (48000, 4800, 5000, 2, 2000, False, 8192),
            (240000, 19200, 5000, 5, 2000, True, 8192), (240000, 9600, 5000, 1, 2000, True, 8192),
            (192000, 40000, 5000, 1, 2500, True, 8192)] * 14
    sigs = [siggen.gmsk_channel(i, 20000, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    g = binding.Batch(cfgs)
    assert g.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs]
    for lo, hi in [(0, 8192), (8192, 8200), (8200, 8201), (8201, 16000), (16000, 16000), (16000, 20000)]:
        lens = [(hi - lo) if i % 5 else max(0, hi - lo - 33) for i in range(len(cfgs))]
        parts = [s[lo:lo + n] for s, n in zip(sigs, lens)]
        g8 = g.process(parts)
        for i, o in enumerate(oracles):
            o8, _ = o.process(parts[i])
            assert np.array_equal(g8[i], o8), (i, lo, hi)
    g.close()


def test_full_size_chunk_properties_131072():
    """BASELINE chunk size (config.conf:11 buffer_size 131072): spot-check channels against the oracle and check the
    size-independent property that splitting the stream differently gives the same symbols (sps < 8)."""
    N = 131072
    cfg = (48000, 9600, 5000, 1, 2000, True)
    sig = siggen.gmsk_batch(8, N)
    a = binding.Batch([cfg + (N,)] * 8)
    b = binding.Batch([cfg + (N,)] * 8)
    whole = a.process([sig[c] for c in range(8)])
    halves = [b.process([sig[c, :50000] for c in range(8)]), b.process([sig[c, 50000:] for c in range(8)])]
    for c in range(8):
        assert np.array_equal(whole[c], np.concatenate([halves[0][c], halves[1][c]]))
    o8, _ = orc.demod_stream(cfg, sig[5], N)
    assert np.array_equal(whole[5], o8)
    a.close()
    b.close()


def test_reference_perf_configuration_maximum_buffer():
    """test/perf_fsk_modem.c: fsk_demod_create(48000, 4800, 5000, 2, 2000, true, 2016000): one call with the whole
    2 016 000-sample buffer (the largest buffer the reference allocates), then the perf loop's own 4096-sample calls of
    `re = (uint8_t) i, im = 0` (perf_fsk_modem.c:81-83), all on one handle."""
    cfg = (48000, 4800, 5000, 2, 2000, True)
    big = 2016000
    d = binding.FskDemod(*cfg, big)
    o = orc.Fsk(*cfg, big)
    assert d.code == 0 and o.code == 0
    sig = siggen.gmsk_channel(11, big, fs=48000, baud=4800)
    g8 = d.process(sig)
    o8, _ = o.process(sig)
    assert len(g8) == len(o8) and abs(len(g8) - big // 10) < 50
    assert np.array_equal(g8, o8)
    ramp = np.zeros(4096, dtype=np.complex64)
    ramp.real = (np.arange(4096) % 256).astype(np.float32)
    for _ in range(20):
        assert np.array_equal(d.process(ramp), o.process(ramp)[0])
    d.close()


def test_one_channel_batch_reset_between_graph_replays():
    """a batch of one channel replays a graph for repeated call lengths; handing the channel to a new stream with other
    parameters (sdrm_batch_reset_channel) must drop the graph built for the old ones"""
    a = (48000, 4800, 5000, 2, 2000, True, 8192)
    b_ = (48000, 9600, 5000, 1, 2000, False, 4096)  # shorter filters, no DC blocker, smaller buffer: fits the batch
    g = binding.Batch([a])
    for cfg, seed in ((a, 1), (b_, 2), (a, 3)):
        if cfg is not a or seed == 3:
            assert g.reset_channel(0, cfg) == 0
        o = orc.Fsk(*cfg)
        sig = siggen.gmsk_channel(seed, 5 * 4096, fs=cfg[0], baud=cfg[1])
        for k in range(5):
            part = sig[k * 4096:(k + 1) * 4096]
            assert np.array_equal(g.process([part])[0], o.process(part)[0]), (cfg, k)
    g.close()


def test_call_longer_than_the_float_position_range():
    """The hand-scheduled symbol loop counts positions in a float's mantissa (1.5 * 2^23 + p, p < 2^22); a call with more
    samples than that takes the C++ form of the loop.  One 4.3 M-sample call (decimation 1, so the clock stage sees all of
    them), then ordinary calls on the same handle; a short-buffer handle on the same input must agree as well."""
    cfg = (48000, 9600, 5000, 1, 2000, True)
    big = (1 << 22) + 100000
    sig = siggen.gmsk_channel(5, big + 3 * 8192)
    d = binding.FskDemod(*cfg, big)
    o = orc.Fsk(*cfg, big)
    assert d.code == 0 and o.code == 0
    g8 = d.process(sig[:big])
    o8, _ = o.process(sig[:big])
    assert len(g8) == len(o8) and len(g8) > big // 5 - 50
    assert np.array_equal(g8, o8)
    for k in range(3):
        part = sig[big + k * 8192: big + (k + 1) * 8192]
        assert np.array_equal(d.process(part), o.process(part)[0])
    d.close()


def test_repeated_calls_of_one_handle_replay_a_graph():
    """A plain fsk_demod handle called again with the same buffer length replays a captured graph (staged input, control
    record, kernels, results: one launch, one wait).  Same-length runs, a length change (new capture), a ragged stretch
    (plain path), NaN input (the clock stage's general loop inside the graph), an empty and an oversize call in between:
    everything bit-identical to the oracle."""
    cfg = (48000, 4800, 5000, 2, 2000, True)
    sig = siggen.gmsk_channel(21, 200000, fs=48000, baud=4800)
    plan = [4096] * 6 + [1000] * 4 + [4096, 333, 4096, 4097, 4095] + [4096] * 3 + [0, 4096, 70000, 4096] + [8192] * 3
    o = orc.Fsk(*cfg, 65536)
    d = binding.FskDemod(*cfg, 65536)
    assert d.code == 0
    pos = 0
    for k, n in enumerate(plan):
        part = sig[pos:pos + n].copy()
        if n == 70000:  # more than the handle's maximum: "<3>requested buffer ..." and no output, stream untouched
            assert len(d.process(part)) == 0
            continue
        pos += n
        if k == 23:  # second call of its length: inside the graph
            part[100:110] = np.nan
        want, _ = o.process(part)
        got = d.process(part)
        assert np.array_equal(got, want), (k, n)
    d.close()


def test_absent_channel_keeps_its_state_on_the_device():
    """SDRM_LEN_ABSENT: a channel that takes no part in a call (a batcher round launched without that client's buffer)
    produces nothing and keeps its stream state, whereas an EMPTY call is answered like the reference answers it (at 16
    samples per symbol the clock stage re-emits a symbol from its carried samples).  Same checks as the CPU emulation's."""
    cfg = (96000, 1200, 5000, 5, 4000, False)
    other = (48000, 9600, 5000, 1, 2000, True)
    sig = siggen.gmsk_channel(9, 14000, fs=96000, baud=1200)
    sig2 = siggen.gmsk_channel(10, 14000)
    sizes = [3000, 3000, 4096, 500, 1, 3000]
    o = orc.Fsk(*cfg, 4096)
    g = binding.Batch([cfg + (4096,), other + (4096,)])
    p = q = 0
    for n in sizes:
        assert len(g.process([binding.ABSENT, sig2[q:q + 700]])[0]) == 0
        q += 700
        got = g.process([sig[p:p + n], binding.ABSENT])
        assert np.array_equal(got[0], o.process(sig[p:p + n])[0]) and len(got[1]) == 0
        p += n
        assert [len(r) for r in g.process([binding.ABSENT, binding.ABSENT])] == [0, 0]
    g.close()
    g = binding.Batch([cfg + (4096,)] * 2)
    o = orc.Fsk(*cfg, 4096)
    p = extra = 0
    for n in sizes:
        assert np.array_equal(g.process([sig[p:p + n]] * 2)[1], o.process(sig[p:p + n])[0])
        p += n
        a, b = g.process([None, None])[0], o.process(sig[0:0])[0]
        assert np.array_equal(a, b)
        extra += len(a)
    assert extra > 0
    g.close()


# ---------------------------------------------------------------- device-resident path + worker surface

def test_device_resident_call_matches_host_call():
    torch = pytest.importorskip("torch")
    C_, N = 64, 8192
    cfg = (48000, 9600, 5000, 1, 2000, True)
    sig = siggen.gmsk_batch(C_, N)
    g = binding.Batch([cfg + (N,)] * C_)
    h = binding.Batch([cfg + (N,)] * C_)
    t = torch.from_numpy(sig.view(np.float32).reshape(C_, 2 * N)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    g.process_device(t.data_ptr(), N, [N] * C_, stream)
    torch.cuda.synchronize()
    data, lens = g.fetch(N)
    ref = h.process([sig[c] for c in range(C_)])
    for c in range(C_):
        assert lens[c] == len(ref[c]) and np.array_equal(data[c, :lens[c]], ref[c])
    g.close()
    h.close()


def test_pipelined_host_path_calls_in_flight_match_oracle():
    """sdrm_batch_arena / _submit / _collect: inputs written into the pinned arena, up to three calls kept in flight, full-length
    and ragged calls (1-D and 2-D copy), results identical to the oracle's stream and to the synchronous host call."""
    C_, N = 48, 8192
    cfgs = [(48000, 9600, 5000, 1, 2000, True, N), (48000, 4800, 5000, 2, 2000, False, N)] * (C_ // 2)
    sigs = [siggen.gmsk_channel(i, 6 * N, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    g = binding.Batch(cfgs)
    assert g.code == 0
    arena = g.arena(4)
    assert arena.shape[0] == 4 and arena.shape[1] == C_
    plan = [[N] * C_, [N] * C_, [(N - 5 * (c % 9)) for c in range(C_)], [0 if c % 5 == 0 else 4000 + c for c in range(C_)],
            [N] * C_, [1000] * C_]
    pos = [0] * C_
    got = [[] for _ in range(C_)]
    pending = 0
    for k, lens in enumerate(plan):
        slot = k % 4
        for c in range(C_):
            part = sigs[c][pos[c]:pos[c] + lens[c]].view(np.float32)
            arena[slot, c, :len(part)] = part
            pos[c] += lens[c]
        code = g.submit(slot, lens)
        if code != 0:  # three already in flight
            for c, o in enumerate(g.collect()):
                got[c].append(o)
            pending -= 1
            assert g.submit(slot, lens) == 0
        pending += 1
    while pending:
        for c, o in enumerate(g.collect()):
            got[c].append(o)
        pending -= 1
    for c in range(0, C_, 5):
        o = orc.Fsk(*cfgs[c])
        exp, p0 = [], 0
        for lens in plan:
            o8, _ = o.process(sigs[c][p0:p0 + lens[c]])
            exp.append(o8)
            p0 += lens[c]
        assert np.array_equal(np.concatenate(got[c]), np.concatenate(exp)), c
    assert g.submit(0, [N] * C_) == 0 and g.submit(1, [N] * C_) == 0 and g.submit(2, [N] * C_) == 0
    assert g.submit(0, [N] * C_) != 0  # a fourth uncollected call is refused, not queued
    for _ in range(3):
        g.collect()
    g.close()


@pytest.mark.parametrize("channels,lanes", [(48, "16"), (4, "16"), (150, "32"), (130, "64x256p")])
def test_three_calls_in_flight_match_oracle_in_every_clock_stage_shape(channels, lanes, monkeypatch):
    """Three calls kept in flight through the pinned-arena path (the stages of consecutive calls overlap on the batch's
    streams), with the clock stage's workgroup shape forced: 14 calls (full, ragged, empty, one channel poisoned by a NaN),
    EVERY channel's stream against the oracle's.  (Until round 5 this also forced SDRM_K3_EARLY, the opt-in overlap of
    consecutive clock stages -- measured a loser outside one narrow range, profiles/r03_clock_early.txt, removed in round 6.)"""
    monkeypatch.setenv("SDRM_K3_LANES", lanes)
    N = 8192
    kinds = [(48000, 9600, 5000, 1, 2000, True, N), (48000, 4800, 5000, 2, 2000, False, N), (240000, 19200, 5000, 5, 2000, True, N)]
    cfgs = [kinds[c % 3] for c in range(channels)]
    distinct = min(channels, 12)
    base = [siggen.gmsk_channel(300 + i, 15 * N, fs=cfgs[i][0], baud=cfgs[i][1]) for i in range(distinct)]
    base[min(5, distinct - 1)][3 * N + 100] = np.nan
    g = binding.Batch(cfgs)
    assert g.code == 0
    arena = g.arena(4)
    plan = [[N] * channels] * 3 + [[(N - 7 * (c % 5)) for c in range(channels)], [0 if c % 4 == 0 else 3000 + c for c in range(channels)],
                                   [N] * channels, [1] * channels, [N] * channels, [0] * channels, [N - 1] * channels] + [[N] * channels] * 4
    pos = [0] * channels
    got = [[] for _ in range(channels)]
    pending = 0
    for k, lens in enumerate(plan):
        slot = k % 4
        for c in range(channels):
            part = base[c % distinct][pos[c]:pos[c] + lens[c]].view(np.float32)
            arena[slot, c, :len(part)] = part
            pos[c] += lens[c]
        if g.submit(slot, lens) != 0:  # three already in flight
            for c, o in enumerate(g.collect()):
                got[c].append(o)
            pending -= 1
            assert g.submit(slot, lens) == 0
        pending += 1
    while pending:
        for c, o in enumerate(g.collect()):
            got[c].append(o)
        pending -= 1
    want = {}
    for c in range(channels):
        key = (c % distinct, tuple(l[c] for l in plan))
        if key not in want:
            o = orc.Fsk(*cfgs[c])
            exp, p0 = [], 0
            for lens in plan:
                exp.append(o.process(base[c % distinct][p0:p0 + lens[c]])[0])
                p0 += lens[c]
            want[key] = np.concatenate(exp)
        assert np.array_equal(np.concatenate(got[c]), want[key]), (channels, lanes, c)
    g.close()


def test_dsp_worker_file_sink_matches_oracle():
    """dsp_worker push/pull surface (src/dsp_worker.c:44-106): put IQ buffers, get rx.demod2client.<id>.s8."""
    L = binding.load()
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    with tempfile.TemporaryDirectory() as tmp:
        cfg = binding.WorkerConfig(48000, 4800, 5000, 2, 2000, True, True, 0, 4096, 4, True, tmp.encode())
        w = C.c_void_p()
        assert L.dsp_worker_create(7, -1, C.byref(cfg), C.byref(w)) == 0
        wid = C.c_uint32(7)
        assert L.dsp_worker_find_by_id(C.byref(wid), w)
        for off in range(0, len(iq), 4096):
            part = np.ascontiguousarray(iq[off:off + 4096]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, len(part) // 2, w)
        L.dsp_worker_destroy(w)  # poison pill is delivered after the queued buffers (blocking queue)
        got = np.fromfile(os.path.join(tmp, "rx.demod2client.7.s8"), dtype=np.int8)
        dump = np.fromfile(os.path.join(tmp, "rx.sdr2demod.7.cf32"), dtype=np.complex64)
    want, _ = orc.demod_stream((48000, 4800, 5000, 2, 2000, True), iq, 4096)
    assert np.array_equal(dump, iq)
    assert np.array_equal(got, want)
    golden = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.s8"), dtype=np.int8)
    assert np.abs(got.astype(np.int32) - golden.astype(np.int32)).max() <= 2


# ---------------------------------------------------------------- next row f-1: Doppler planner + NCO (doppler.c, sig_source.c)

import json  # noqa: E402

DOPPLER = json.load(open(os.path.join(GOLDEN, "doppler_shifts_lucky7.json")))
NCO_TOL = 0.01  # the reference's own tolerance for the corrected IQ (test/utils.c:137 via test/test_doppler.c:60)


def test_doppler_corrected_iq_matches_reference_golden_file():
    """test/test_doppler.c:37-76 through the device NCO: lucky7.cf32 -> lucky7.expected.cf32 at the reference's 0.01, and
    bit for bit the oracle's corrected IQ: where the device's double cos/sin lies next to an fp32 rounding boundary the
    sample is re-evaluated in double-double and rounded once (csrc/sdrm_core.h, sdrm_nco_sample), which is what the host
    libm's (float) cos gives on all but ~1e-9 of the phases."""
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.cf32"), dtype=np.complex64)
    want = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.float32)
    chunk = 2000  # the reference harness reads 2000 samples at a time; the interpolated shift advances per call
    g = binding.Batch([(48000, 4800, 5000, 2, 2000, True, chunk)])
    planner = binding.DopplerPlanner(48000, lambda k: DOPPLER["shifts_hz"][min(k, len(DOPPLER["shifts_hz"]) - 1)])
    o = orc.Doppler(48000, DOPPLER["shifts_hz"], chunk)
    got, ref = [], []
    for off in range(0, len(iq), chunk):
        part = iq[off:off + chunk]
        segs = planner.plan(0, len(part))
        g.process_nco([part], segs)
        got.append(g.last_mixed(0))
        ref.append(o.process(part.view(np.float32)))
    got, ref = np.concatenate(got), np.concatenate(ref)
    assert len(got) == len(want)
    assert np.abs(got - want).max() < NCO_TOL
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    g.close()


def test_doppler_then_demod_soft_bits_match_oracle():
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.cf32"), dtype=np.complex64)
    cfg = (48000, 4800, 5000, 2, 2000, True)
    chunk = 20000
    g = binding.Batch([cfg + (chunk,)], keep_soft=True)
    planner = binding.DopplerPlanner(48000, lambda k: DOPPLER["shifts_hz"][min(k, len(DOPPLER["shifts_hz"]) - 1)])
    o = orc.Fsk(*cfg, chunk)
    d = orc.Doppler(48000, DOPPLER["shifts_hz"], chunk)
    for off in range(0, len(iq), chunk):
        part = iq[off:off + chunk]
        o8, of = o.process(d.process(part.view(np.float32)))
        g8 = g.process_nco([part], planner.plan(0, len(part)))[0]
        gf = g.last_soft(0)
        assert_same(of, gf, o8, g8, where="offset %d" % off)  # corrected IQ identical => soft bits identical
    g.close()


@pytest.mark.parametrize("decims", [(5, 8), (1, 1)], ids=["decimated", "undecimated_tail_quirk"])
def test_mixed_rate_batch_with_per_channel_doppler_ramp(decims):
    """BASELINE config 5 in miniature: 240 kHz / 19200 baud and 48 kHz / 1200 baud channels, each with its own linear
    Doppler ramp (+-10 kHz over the run, piecewise-constant per <= 1 s batch), some uncorrected.  Decimated to fewer than 8
    samples per symbol, and undecimated (12.5 and 40 samples per symbol, DC lengths 400 and 1280): there the reference's
    clock stage re-emits a symbol at chunk edges (clock_recovery_mm.c:127-133), which the device must reproduce call by
    call behind the oscillator."""
    cfgs = ([(240000, 19200, 5000, decims[0], 2000, True, 60000)] * 3 + [(48000, 1200, 5000, decims[1], 2000, True, 60000)] * 3) * 2
    n_calls, n = 3, 60000
    sigs = [siggen.gmsk_channel(i, n_calls * n, fs=c[0], baud=c[1], carrier_offset_hz=0.0) for i, c in enumerate(cfgs)]
    ramps = [(lambda k, i=i: -10000.0 + 2500.0 * k + 37.0 * i) if i % 3 else None for i in range(len(cfgs))]
    g = binding.Batch(cfgs, keep_soft=True)
    assert g.code == 0
    planners = [binding.DopplerPlanner(c[0], r) if r else None for c, r in zip(cfgs, ramps)]
    oracles = [orc.Fsk(*c) for c in cfgs]
    odops = [orc.Doppler(c[0], [r(k) for k in range(8)], n) if r else None for c, r in zip(cfgs, ramps)]
    for call in range(n_calls):
        parts = [s[call * n:(call + 1) * n] for s in sigs]
        segs = []
        for i, p in enumerate(planners):
            if p is not None:
                segs += p.plan(i, n)
        g8 = g.process_nco(parts, segs)
        for i, o in enumerate(oracles):
            x = parts[i].view(np.float32)
            if odops[i] is not None:
                x = odops[i].process(x)
            o8, of = o.process(x)
            assert_same(of, g.last_soft(i), o8, g8[i], where="call %d channel %d" % (call, i))
    g.close()


def test_nco_ragged_batches_and_large_steps_match_oracle():
    """The phase generator works in 64-sample blocks, 64 channels per workgroup: batches that end anywhere inside a
    block, empty batches, empty and odd-length inputs, channels that skip the correction for a call, a partly filled
    second workgroup, and shifts beyond the sampling rate (|step| > 2 pi: the reference's two-test wrap,
    sig_source.c:47-53) must all give the oscillator the oracle gives, sample for sample."""
    rng = np.random.default_rng(77)
    n_ch, maxlen, fs = 70, 5000, 48000
    g = binding.Batch([(fs, 9600, 5000, 1, 2000, True, maxlen)] * n_ch)
    assert g.code == 0
    ncos = [orc.Nco(1.0, fs, maxlen) for _ in range(n_ch)]
    total = same = 0
    for call in range(4):
        parts, segs, want = [], [], []
        for c in range(n_ch):
            n = int(rng.choice([0, 1, 3, 63, 64, 65, 1000, 4097, maxlen])) if rng.random() < 0.5 else int(rng.integers(0, maxlen + 1))
            x = (rng.standard_normal(2 * n) * 0.5).astype(np.float32)
            parts.append(x.view(np.complex64))
            if n == 0 or rng.random() < 0.1:
                want.append(None)
                continue
            k = int(rng.integers(1, 7))
            cuts = np.sort(rng.integers(0, n + 1, size=k - 1))
            lens = np.diff(np.concatenate([[0], cuts, [n]])).astype(int)
            ref, off = [], 0
            for ln in lens:
                f = int(rng.choice([60000, -70000, 123457])) if rng.random() < 0.08 else int(rng.integers(-20000, 20001))
                segs.append((c, int(ln), f))
                if ln:
                    ref.append(ncos[c].multiply(f, x[2 * off:2 * (off + ln)]))
                off += ln
            want.append(np.concatenate(ref))
        g.process_nco(parts, segs)
        for c in range(n_ch):
            got = g.last_mixed(c)
            if want[c] is None:
                assert len(got) == 0, (call, c)
                continue
            assert len(got) == len(want[c]), (call, c)
            assert np.array_equal(got.view(np.uint32), want[c].view(np.uint32)), (call, c)
    g.close()


def test_rx_client_on_a_socket_gets_the_response_and_then_the_soft_bits():
    """what an RX client of the reference sees on its socket -- the Response (src/api_utils.c:82-108, sent at
    src/tcp_server.c:677) and behind it the raw int8 soft bits the worker writes (src/dsp_worker.c:93-95) -- produced by
    the wire helper + a GPU worker configured from a hand-encoded RxRequest (api.proto:35-49), on a socketpair."""
    import socket
    import threading
    L = binding.load()
    L.sdrm_wire_write_response.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    L.sdrm_wire_decode_rx_request.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(binding.WorkerConfig), C.POINTER(C.c_int)]

    def varint(v):
        out = bytearray()
        v &= (1 << 64) - 1
        while True:
            b = v & 0x7f
            v >>= 7
            out.append(b | (0x80 if v else 0))
            if not v:
                return bytes(out)

    def field(num, value):
        return (varint(num << 3 | 2) + varint(len(value)) + value) if isinstance(value, bytes) else varint(num << 3) + varint(value)

    fsk = field(1, 5000) + field(2, 2000) + field(3, 1)
    body = (field(1, 437525000) + field(2, 48000) + field(3, 0) + field(4, 0) + field(5, 1) + field(6, 4800) + field(7, 2) +
            field(8, 1) + field(10, fsk))  # destination SOCKET
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    server, client = socket.socketpair()
    received = bytearray()

    def reader():
        while True:
            chunk = client.recv(65536)
            if not chunk:
                break
            received.extend(chunk)

    t = threading.Thread(target=reader)
    t.start()
    with tempfile.TemporaryDirectory() as tmp:
        cfg = binding.WorkerConfig()
        buf = (C.c_uint8 * len(body)).from_buffer_copy(body)
        assert L.sdrm_wire_decode_rx_request(buf, len(body), C.byref(cfg), None) == 0
        cfg.buffer_size, cfg.queue_size, cfg.rx_file_source, cfg.base_path = 4096, 4, True, tmp.encode()  # the server_config half
        w = C.c_void_p()
        assert L.dsp_worker_create(21, server.fileno(), C.byref(cfg), C.byref(w)) == 0
        assert L.sdrm_wire_write_response(server.fileno(), 0, 21) == 0
        for off in range(0, len(iq), 4096):
            part = np.ascontiguousarray(iq[off:off + 4096]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, len(part) // 2, w)
        L.dsp_worker_destroy(w)
    server.close()
    t.join(30)
    client.close()
    assert bytes(received[:10]) == bytes([0, 2, 0, 0, 0, 4, 0x08, 0x00, 0x10, 21])
    got = np.frombuffer(bytes(received[10:]), dtype=np.int8)
    want, _ = orc.demod_stream((48000, 4800, 5000, 2, 2000, True), iq, 4096)
    assert np.array_equal(got, want)
    golden = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.s8"), dtype=np.int8)
    assert len(got) == len(golden) and np.abs(got.astype(np.int32) - golden.astype(np.int32)).max() <= 2  # test_fsk_demod.c:47


def test_dsp_worker_with_doppler_callback():
    """the worker's Doppler leg (reference src/dsp_worker.c:65-71): shifts come from a per-second callback"""
    L = binding.load()
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.cf32"), dtype=np.complex64)
    shifts = DOPPLER["shifts_hz"]
    cb = binding.SHIFT_FN(lambda user, k: float(shifts[min(int(k), len(shifts) - 1)]))
    with tempfile.TemporaryDirectory() as tmp:
        cfg = binding.WorkerConfig(48000, 4800, 5000, 2, 2000, True, False, 0, 4096, 4, True, tmp.encode(),
                                   C.cast(cb, C.c_void_p), None)
        w = C.c_void_p()
        assert L.dsp_worker_create(9, -1, C.byref(cfg), C.byref(w)) == 0
        for off in range(0, len(iq), 4096):
            part = np.ascontiguousarray(iq[off:off + 4096]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, len(part) // 2, w)
        L.dsp_worker_destroy(w)
        got = np.fromfile(os.path.join(tmp, "rx.demod2client.9.s8"), dtype=np.int8)
    o = orc.Fsk(48000, 4800, 5000, 2, 2000, True, 4096)
    d = orc.Doppler(48000, shifts, 4096)
    want = np.concatenate([o.process(d.process(iq[off:off + 4096].view(np.float32)))[0] for off in range(0, len(iq), 4096)])
    assert np.array_equal(got, want)


# ---------------------------------------------------------------- next row f-3: per-GPU batcher behind many clients

import threading  # noqa: E402


def test_batcher_many_clients_on_the_device_match_oracle():
    """sdrm_batcher_*: 24 clients (mixed configurations, one with Doppler pre-correction), each with its own producer
    and consumer thread, share one batch; every client's soft bits equal the oracle's for its own stream."""
    cfgs = [(48000, 9600, 5000, 1, 2000, True, 8192), (48000, 4800, 5000, 2, 2000, False, 8192),
            (240000, 19200, 5000, 5, 2000, True, 8192)] * 8
    K, sizes = 5, [8192, 3000, 8192, 17, 8000]
    sigs = [siggen.gmsk_channel(i, sum(sizes), fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    chunks = [[s[sum(sizes[:k]):sum(sizes[:k + 1])] for k in range(K)] for s in sigs]
    bt = binding.Batcher(cfgs, slots=4, max_wait_us=50000, blocking=True)
    assert bt.code == 0
    shifts = [1500.0 - 400.0 * k for k in range(8)]
    planner = binding.DopplerPlanner(48000, lambda k: shifts[min(k, 7)])
    assert bt.set_doppler(1, planner) == 0
    got = [[] for _ in cfgs]

    def producer(c):
        for k in range(K):
            bt.put(c, chunks[c][k])

    def consumer(c):
        for k in range(K):
            got[c].append(bt.take(c))

    th = [threading.Thread(target=f, args=(c,)) for c in range(len(cfgs)) for f in (producer, consumer)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        assert not t.is_alive()
    for c, cfg in enumerate(cfgs):
        o = orc.Fsk(*cfg)
        d = orc.Doppler(48000, shifts, 8192) if c == 1 else None
        for k in range(K):
            x = chunks[c][k].view(np.float32)
            want = o.process(d.process(x) if d else chunks[c][k])[0]
            # the Doppler-corrected client too: the device's oscillator is the correctly rounded one
            assert np.array_equal(got[c][k], want), (c, k)
    # batched: about one device call per buffer index, not 24 * K (a round also goes 50 ms after its first buffer: a loaded box's
    # producer threads can miss that now and then -- 9 rounds seen once for K = 5)
    assert bt.rounds() <= 2 * K + 4
    for c in range(len(cfgs)):
        bt.interrupt(c)
    assert bt.take(0) is None
    bt.close()


def test_dsp_workers_sharing_one_batcher_write_the_reference_files():
    """dsp_worker surface on top of a shared batcher: 6 workers, file sinks byte-identical to the oracle's output."""
    L = binding.load()
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    n_w = 6
    cfg = (48000, 4800, 5000, 2, 2000, True, 4096)
    bt = binding.Batcher([cfg] * n_w, slots=4, max_wait_us=20000, blocking=True)
    assert bt.code == 0
    with tempfile.TemporaryDirectory() as tmp:
        ws = []
        for i in range(n_w):
            wc = binding.WorkerConfig(48000, 4800, 5000, 2, 2000, True, i == 0, 0, 4096, 4, True, tmp.encode())
            wc.batcher = bt.h
            wc.batcher_channel = i
            w = C.c_void_p()
            assert L.dsp_worker_create(20 + i, -1, C.byref(wc), C.byref(w)) == 0
            ws.append(w)
        shift = [37 * i for i in range(n_w)]  # every client sees the recording from a different offset

        def feed(i):
            x = iq[shift[i]:]
            for off in range(0, len(x), 4096):
                part = np.ascontiguousarray(x[off:off + 4096]).view(np.float32)
                L.dsp_worker_put(part.ctypes.data, len(part) // 2, ws[i])

        th = [threading.Thread(target=feed, args=(i,)) for i in range(n_w)]
        for t in th:
            t.start()
        for t in th:
            t.join(120)
        for w in ws:
            L.dsp_worker_destroy(w)
        for i in range(n_w):
            got = np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % (20 + i)), dtype=np.int8)
            want, _ = orc.demod_stream(cfg[:6], iq[shift[i]:], 4096)
            assert np.array_equal(got, want), i
        dump = np.fromfile(os.path.join(tmp, "rx.sdr2demod.20.cf32"), dtype=np.complex64)
        assert np.array_equal(dump, iq)
    bt.close()


# ---------------------------------------------------------------- next row f-2: file source -> workers -> file sinks

import subprocess  # noqa: E402


@pytest.mark.parametrize("workers", [1, 3])
def test_file_source_harness_reproduces_the_reference_fixture_files(workers):
    """tools/file_demod.c: a C program against the C-ABI alone reads lucky7.expected.cf32 in 4096-sample chunks like the
    reference's file source (file_source.c:101), feeds dsp_workers (one private demodulator, or three on a shared
    batcher) and leaves rx.demod2client.<id>.s8: identical to the oracle, within the reference's 2 LSB of its golden."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "file_demod")
    if not os.path.exists(exe):
        subprocess.check_call(["gcc", "-O2", "-pthread", os.path.join(root, "tools", "file_demod.c"), "-I" + os.path.join(root, "include"),
                               "-L" + os.path.join(root, "sdr-modem_amd", "csrc"), "-lsdrmodem_hip",
                               "-Wl,-rpath," + os.path.join(root, "sdr-modem_amd", "csrc"), "-o", exe])
    src = os.path.join(GOLDEN, "lucky7.expected.cf32")
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([exe, "-d", "-n", str(workers), src, tmp, "48000", "4800", "5000", "2", "2000", "1"], timeout=120)
        iq = np.fromfile(src, dtype=np.complex64)
        want, _ = orc.demod_stream((48000, 4800, 5000, 2, 2000, True), iq, 4096)
        golden = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.s8"), dtype=np.int8)
        for i in range(workers):
            got = np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % i), dtype=np.int8)
            assert np.array_equal(got, want), i
            assert len(got) == len(golden) and np.abs(got.astype(np.int32) - golden.astype(np.int32)).max() <= 2
        assert np.array_equal(np.fromfile(os.path.join(tmp, "rx.sdr2demod.0.cf32"), dtype=np.complex64), iq)


# ---------------------------------------------------------------- channels changing hands

def test_batch_channel_reassignment_on_the_device():
    """sdrm_batch_reset_channel: after streaming, channels are given other configurations (other rates, decimation, DC
    blocker off) or just a new stream; they then match freshly created oracles, their neighbours are undisturbed, a
    configuration with a longer buffer than the batch's is refused and leaves the channel alone."""
    big = (48000, 4800, 5000, 2, 2000, True, 8192)
    cfgs = [big, (48000, 9600, 5000, 1, 2000, True, 8192), big, (240000, 19200, 5000, 5, 2000, True, 8192)]
    g = binding.Batch(cfgs, keep_soft=True)
    assert g.code == 0
    sigs = [siggen.gmsk_channel(300 + i, 3 * 8192, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    orcs = [orc.Fsk(*c) for c in cfgs]
    g8 = g.process([s[:8192] for s in sigs])
    for i in range(4):
        assert np.array_equal(g8[i], orcs[i].process(sigs[i][:8192])[0])
    new1 = (48000, 9600, 5000, 1, 2000, False, 8192)
    new3 = (48000, 2400, 2400, 4, 1000, True, 4096)
    assert g.reset_channel(1, new1) == 0 and g.reset_channel(2) == 0 and g.reset_channel(3, new3) == 0
    assert g.reset_channel(0, (48000, 4800, 5000, 2, 2000, True, 16384)) != 0  # the buffer length is the batch's for life
    orcs[1], orcs[2], orcs[3] = orc.Fsk(*new1), orc.Fsk(*big), orc.Fsk(*new3)
    sigs[1] = siggen.gmsk_channel(311, 3 * 8192, fs=48000, baud=9600)
    sigs[3] = siggen.gmsk_channel(313, 3 * 8192, fs=48000, baud=2400)
    for k in (1, 2):
        parts = [s[k * 8192:(k + 1) * 8192] for s in sigs]
        parts[3] = parts[3][:4096]
        g8 = g.process(parts)
        for i in range(4):
            o8, of = orcs[i].process(parts[i])
            assert_same(of, g.last_soft(i), o8, g8[i], where="channel %d call %d" % (i, k))
    g.close()


def test_batch_grows_when_a_client_needs_longer_filters_or_a_longer_dc_boxcar():
    """Round 3: the batch's geometry is no longer frozen when it is created.  A batch of five 9600-baud channels WITHOUT DC
    blocker streams; then one slot gets the batch's first DC blocker, another a 240 kHz / 19200 baud client (LPF1 of 397
    taps instead of 117, LPF2 of 289 instead of 57, a 400-sample boxcar), later a third a 1200-baud client (1280-sample
    boxcar, 40 samples per symbol): every time the batch grows -- raw histories to a new stride, DC states to a new layout,
    private tap slots to a new size -- and the channels that keep their clients continue their streams bit for bit."""
    n = 6000
    base = (48000, 9600, 5000, 1, 2000, False, n)
    cfgs = [base] * 5
    g = binding.Batch(cfgs, keep_soft=True)
    assert g.code == 0
    sigs = [siggen.gmsk_channel(500 + i, 8 * n, fs=48000, baud=9600) for i in range(5)]
    orcs = [orc.Fsk(*c) for c in cfgs]
    pos = [0] * 5

    def calls(k):
        for _ in range(k):
            lens = [n, n - 17, n, 1000, n]
            parts = [s[p:p + ln] for s, p, ln in zip(sigs, pos, lens)]
            for i, ln in enumerate(lens):
                pos[i] += ln
            g8 = g.process(parts)
            for i in range(5):
                o8, of = orcs[i].process(parts[i])
                assert_same(of, g.last_soft(i), o8, g8[i], where="channel %d at %d" % (i, pos[i]))

    def hand_over(ch, cfg, seed):
        assert g.reset_channel(ch, cfg) == 0, cfg
        orcs[ch] = orc.Fsk(*cfg)
        sigs[ch] = siggen.gmsk_channel(seed, 8 * n, fs=cfg[0], baud=cfg[1])
        pos[ch] = 0

    calls(2)
    hand_over(1, (48000, 9600, 5000, 1, 2000, True, n), 601)      # the batch's first DC blocker
    calls(1)
    hand_over(3, (240000, 19200, 5000, 1, 2000, True, n), 603)    # longer filters, longer history, longer boxcar
    calls(2)
    hand_over(0, (48000, 1200, 5000, 1, 2000, True, n), 600)      # 1280-sample boxcar
    hand_over(3, base, 613)                                        # and back to a small configuration in a grown batch
    calls(2)
    info = g.info(0)
    assert (info.dc_length, g.info(1).dc_length, g.info(3).dc_length) == (1280, 160, 0)
    assert g.reset_channel(2, (48000, 9600, 5000, 1, 2000, True, 2 * n)) != 0  # only the buffer length cannot grow
    calls(1)
    g.close()


def test_worker_slots_are_reused_by_clients_with_other_parameters():
    """The server flow on a shared batcher: two workers run, one leaves, a third client with different demodulator
    parameters takes over its slot; every file sink equals the oracle's output for that client's stream."""
    L = binding.load()
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    slot_cfg = (48000, 4800, 5000, 2, 2000, True, 4096)
    bt = binding.Batcher([slot_cfg] * 2, slots=4, max_wait_us=5000, blocking=True)
    assert bt.code == 0

    def make(wid, slot, baud, decim, dc):
        wc = binding.WorkerConfig(48000, baud, 5000, decim, 2000, dc, False, 0, 4096, 4, True, tmp.encode())
        wc.batcher = bt.h
        wc.batcher_channel = slot
        w = C.c_void_p()
        assert L.dsp_worker_create(wid, -1, C.byref(wc), C.byref(w)) == 0
        return w

    def feed(w, x):
        for off in range(0, len(x), 4096):
            part = np.ascontiguousarray(x[off:off + 4096]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, len(part) // 2, w)

    with tempfile.TemporaryDirectory() as tmp:
        w0, w1 = make(40, 0, 4800, 2, True), make(41, 1, 4800, 2, True)
        half = (len(iq) // 2) // 4096 * 4096
        t = [threading.Thread(target=feed, args=(w, iq[:half])) for w in (w0, w1)]
        for x in t:
            x.start()
        for x in t:
            x.join(120)
        L.dsp_worker_destroy(w1)                       # client 41 leaves
        w2 = make(42, 1, 9600, 1, False)               # client 42 takes slot 1 with other parameters
        t = [threading.Thread(target=feed, args=(w0, iq[half:])), threading.Thread(target=feed, args=(w2, iq[:half]))]
        for x in t:
            x.start()
        for x in t:
            x.join(120)
        L.dsp_worker_destroy(w0)
        L.dsp_worker_destroy(w2)
        got = {i: np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % i), dtype=np.int8) for i in (40, 41, 42)}
    assert np.array_equal(got[40], orc.demod_stream((48000, 4800, 5000, 2, 2000, True), iq, 4096)[0])
    assert np.array_equal(got[41], orc.demod_stream((48000, 4800, 5000, 2, 2000, True), iq[:half], 4096)[0])
    assert np.array_equal(got[42], orc.demod_stream((48000, 9600, 5000, 1, 2000, False), iq[:half], 4096)[0])
    bt.close()


def test_plain_fsk_demod_handles_share_one_batcher_when_asked_to(monkeypatch):
    """SDRM_SHARED_SLOTS: the reference's own usage -- one fsk_demod handle and one DSP thread per client, one blocking
    fsk_demod_process per buffer -- with the handles of the process served by one batched device call per round.
    Twelve threads with different parameters (all within the first handle's geometry) and one handle whose filters
    are longer (falls back to a private batch): every stream bit-exact, far fewer device calls than buffers."""
    monkeypatch.setenv("SDRM_SHARED_SLOTS", "16")
    monkeypatch.setenv("SDRM_SHARED_WAIT_US", "20000")
    first = (48000, 4800, 5000, 2, 2000, True, 4096)           # 157 / 57 taps, DC on: the shared geometry
    others = [(48000, 9600, 5000, 1, 2000, True, 4096), (48000, 9600, 5000, 1, 2000, False, 4096),
              (48000, 4800, 5000, 2, 2000, False, 2048)]
    cfgs = [first] + [others[i % 3] for i in range(11)] + [(48000, 1200, 5000, 8, 2000, True, 4096)]  # last: 207 taps
    K = 6
    sigs = [siggen.gmsk_channel(500 + i, K * c[6], fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    handles = [binding.FskDemod(*c) for c in cfgs]  # created in order: the first fixes the geometry
    assert all(h.code == 0 for h in handles)
    got = [[] for _ in cfgs]
    start = threading.Barrier(len(cfgs))

    def client(i):
        n = cfgs[i][6]
        start.wait()
        for k in range(K):
            got[i].append(handles[i].process(sigs[i][k * n:(k + 1) * n]))

    th = [threading.Thread(target=client, args=(i,)) for i in range(len(cfgs))]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
        assert not t.is_alive()
    for i, c in enumerate(cfgs):
        o = orc.Fsk(*c)
        n = c[6]
        for k in range(K):
            assert np.array_equal(got[i][k], o.process(sigs[i][k * n:(k + 1) * n])[0]), (i, k)
    for h in handles:
        h.close()


# ---------------------------------------------------------------- any samples per symbol (generic DC / clock stages)

def test_any_samples_per_symbol_on_the_device():
    """fsk_demod_create accepts what the reference accepts (src/dsp/fsk_demod.c:53-63): 240 kHz / 600 baud without decimation is
    400 samples per symbol -- a 1091-tap LPF1, a 12800-sample DC boxcar, up to 412 samples carried by the clock stage -- beyond
    what the LDS-resident DC and clock stages hold; such a channel keeps the fast front-end and runs k2_dc_generic /
    k3_clock_generic (state in global memory).  Plain handle over ragged calls (the one-channel graph replay must stay off),
    and a batch that mixes generic and ordinary channels, absent calls included; int8 and float soft bits bit for bit."""
    iq = siggen.gmsk_channel(6, 130000, fs=240000, baud=600)
    chunks = [16384, 1000, 0, 16384, 7, 20000, 1, 1, 300, 20000, 20000, 20000, 15923]
    assert run_stream((240000, 600, 5000, 1, 2000, True), iq, chunks, 20000) > 250
    d = binding.FskDemod(240000, 600, 5000, 1, 2000, True, 20000)
    o = orc.Fsk(240000, 600, 5000, 1, 2000, True, 20000)
    assert d.code == 0
    for k in range(5):  # equal lengths: a one-channel batch would replay a graph from the second call on
        part = iq[k * 20000:(k + 1) * 20000]
        assert np.array_equal(d.process(part), o.process(part)[0]), k
    d.close()
    iq2 = siggen.gmsk_channel(7, 60000, fs=240000, baud=900)
    run_stream((240000, 900, 5000, 1, 2000, False), iq2, [9000, 300, 20700, 5, 20000, 9995], 20700)
    # the longest filters a tile's LDS holds: 2.4 MHz / 9600 baud, 5899 + 2891 taps (118 of the CU's 160 KiB; the halo is longer than
    # the tile and the history longer than most calls); one size up (10909 taps) is refused when the batch is planned
    iq3 = siggen.gmsk_channel(9, 30000, fs=2400000, baud=9600)
    run_stream((2400000, 9600, 5000, 1, 2000, True), iq3, [4096, 4096, 100, 4096, 7, 8192, 3000], 8192)
    assert binding.Batch([(2400000, 600, 5000, 1, 2000, True, 4096)]).code == -95  # -ENOTSUP, no launch attempted
    run_stream((240000, 900, 5000, 1, 2000, True), iq2, [20700, 20700, 18600], 20700)
    cfgs = [(240000, 600, 5000, 1, 2000, True, 8192), (48000, 9600, 5000, 1, 2000, True, 8192), (240000, 900, 5000, 1, 2000, False, 8192),
            (48000, 4800, 5000, 2, 2000, False, 8192)] * 10  # 40 channels: the pipelined stages, 20 generic channels
    sigs = [siggen.gmsk_channel(20 + i, 3 * 8192, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs[:4])]
    g = binding.Batch(cfgs, keep_soft=True)
    assert g.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs[:4]]
    pos = [0] * 4
    for lens in ([8192] * 4, [100, 8192, 0, 17], [8092, 3000, 8192, 8192]):
        parts = [sigs[i % 4][pos[i % 4]:pos[i % 4] + lens[i % 4]] for i in range(len(cfgs))]
        g8 = g.process(parts)
        want = [o.process(parts[i]) for i, o in enumerate(oracles)]
        for i in range(len(cfgs)):
            o8, of = want[i % 4]
            assert_same(of, g.last_soft(i), o8, g8[i], where="channel %d lens %s" % (i, lens))
        pos = [pos[i] + lens[i] for i in range(4)]
    # a slot handed from an ordinary client to a generic one and back
    assert g.reset_channel(1, (240000, 600, 5000, 1, 2000, True, 8192)) == 0
    o = orc.Fsk(240000, 600, 5000, 1, 2000, True, 8192)
    got = g.process([binding.ABSENT, sigs[0][:8192]] + [binding.ABSENT] * (len(cfgs) - 2))
    assert np.array_equal(got[1], o.process(sigs[0][:8192])[0]) and len(got[0]) == 0
    assert g.reset_channel(1, (48000, 9600, 5000, 1, 2000, True, 8192)) == 0
    o = orc.Fsk(48000, 9600, 5000, 1, 2000, True, 8192)
    got = g.process([binding.ABSENT, sigs[1][:8192]] + [binding.ABSENT] * (len(cfgs) - 2))
    assert np.array_equal(got[1], o.process(sigs[1][:8192])[0])
    g.close()


# ---------------------------------------------------------------- stages of ONE call overlap (in-call hand-off, round 5)

def test_hand_off_in_a_batch_whose_stage_buffers_exceed_four_gibibytes():
    """2048 channels x 524288 samples: the front-end's output array is 4 GiB, the DC blocker's likewise.  The hand-off's device-
    scope accesses go through raw buffer resources with 32-bit byte offsets; until round 6 one resource spanned the whole array
    and the offset of channel 1024's row wrapped to channel 0's (advisor's finding: the DC stage then read and wrote other
    channels' rows, silently).  Now a resource starts at the row of the DC workgroup's first channel.  One blocking call with
    the hand-off (64 clock-stage + 128 DC workgroups waiting: the limit exactly), every channel reading the same row
    (input stride 0: no 8 GiB of input needed); channels on both sides of the 4 GiB line and the last one against the oracle."""
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 48 * (1 << 30):
        pytest.skip("needs ~40 GB of device memory")
    n, C_ = 524288, 2048
    cfg = (48000, 9600, 5000, 1, 2000, True)
    row = siggen.gmsk_channel(77, n)
    want8, _ = orc.demod_stream(cfg, row, n)
    x = torch.from_numpy(row.view(np.float32).copy()).cuda()
    g = binding.Batch([cfg + (n,)] * C_, calibrate=False)
    assert g.code == 0
    assert g.schedule()["clock_stage"] == "32x512"
    g.process_device(x.data_ptr(), 0, [n] * C_, torch.cuda.current_stream().cuda_stream)
    g.sync()
    assert g.handoff_calls() == 1
    data, got = g.fetch(len(want8) + 64)
    for c in (0, 1, 1023, 1024, 1025, 1535, 2047):
        assert got[c] == len(want8) and np.array_equal(data[c, :got[c]], want8), c
    # a second call behind it (stream order, no hand-off: the batch is not idle when it is enqueued) continues every stream
    g.process_device(x.data_ptr(), 0, [n] * C_, torch.cuda.current_stream().cuda_stream)
    g.process_device(x.data_ptr(), 0, [n] * C_, torch.cuda.current_stream().cuda_stream)
    g.sync()
    o = orc.Fsk(*cfg, n)
    for _ in range(3):
        last, _ = o.process(row)
    data, got = g.fetch(len(last) + 64)
    for c in (0, 1024, 2047):
        assert got[c] == len(last) and np.array_equal(data[c, :got[c]], last), c
    g.close()
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("handoff", ["1", "0"])
def test_blocking_calls_with_the_stages_of_one_call_resident_together(handoff, monkeypatch):
    """A blocking call meets an idle batch (the reference's caller waits for fsk_demod_process, src/dsp_worker.c:75): its
    front-end, DC blocker and clock recovery are then resident TOGETHER, each starting on the first finished pieces of the one
    in front (tile stamps / output counts in device memory, device-scope accesses) instead of on its end.  Same bits as
    the oracle for everything that travels through that hand-over: channels with and without DC blocker (the clock stage then
    reads the front-end's tiles directly), decimation (tiles shorter than a staging block), ragged, empty and absent inputs,
    NaN / Inf / beyond-the-tame-amplitude samples LATE in a call (their flags reach the clock stage block by block, after it
    has produced symbols: the NaN-aware form from there on, or the call run again from its start by sdrm_k3_rescue), and a
    channel with fewer than one sample per symbol.  SDRM_HANDOFF=0: the same calls with the stages one after the other."""
    monkeypatch.setenv("SDRM_HANDOFF", handoff)
    maxlen = 40000
    cfgs = [(48000, 9600, 5000, 1, 2000, True), (48000, 9600, 5000, 1, 2000, False), (48000, 4800, 5000, 2, 2000, True),
            (240000, 19200, 5000, 5, 2000, True), (48000, 1200, 5000, 8, 2000, False), (192000, 40000, 5000, 1, 2000, True),
            (48000, 9600, 3, 1, 2000, True), (48000, 9600, 3, 1, 2000, False), (48000, 9600, 5000, 7, 2000, True),
            (48000, 9600, 5000, 1, 2000, True), (48000, 9600, 5000, 1, 2000, False), (240000, 9600, 5000, 1, 2000, True)]
    full = [c + (maxlen,) for c in cfgs]
    g = binding.Batch(full, keep_soft=True)
    assert g.code == 0
    oracles = [orc.Fsk(*c) for c in full]
    sigs = [siggen.gmsk_channel(700 + i, 5 * maxlen, fs=c[0], baud=c[1]).copy() for i, c in enumerate(cfgs)]
    rng = np.random.default_rng(5)
    # channels 9 and 10: a NaN, an Inf and a burst of huge samples three quarters into the second and third calls
    for i in (9, 10):
        sigs[i][maxlen + 30000] = np.nan
        sigs[i][2 * maxlen + 31000] = np.inf
        sigs[i][3 * maxlen + 29000:3 * maxlen + 29040] *= np.float32(1e30)
    # channels 6 and 7 (discriminator gain 2546): quiet until three quarters into each call, then full-scale noise
    for i in (6, 7):
        quiet = np.exp(2j * np.pi * 1e-6 * np.arange(5 * maxlen)).astype(np.complex64)
        loud = (rng.normal(0, 0.7, 5 * maxlen) + 1j * rng.normal(0, 0.7, 5 * maxlen)).astype(np.complex64)
        mask = (np.arange(5 * maxlen) % maxlen) > 30000
        sigs[i] = np.where(mask, loud, quiet).astype(np.complex64)
    pos = [0] * len(cfgs)
    plans = [[maxlen] * len(cfgs), [maxlen] * len(cfgs), [maxlen] * len(cfgs), [maxlen] * len(cfgs),
             [int(rng.choice([0, 1, 7, 100, 3839, 3840, 3841, 20000, maxlen])) for _ in cfgs]]
    for call, lens in enumerate(plans):
        parts = [s[p:p + n] for s, p, n in zip(sigs, pos, lens)]
        if call == 4:
            parts[2] = binding.ABSENT if hasattr(binding, "ABSENT") else parts[2]
        g8 = g.process(parts)
        for i, o in enumerate(oracles):
            if call == 4 and i == 2 and hasattr(binding, "ABSENT"):
                assert len(g8[i]) == 0
                continue
            pos[i] += lens[i]
            o8, of = o.process(parts[i])
            gf = g.last_soft(i)
            assert len(o8) == len(g8[i]), (cfgs[i], call, len(o8), len(g8[i]))
            same = (of.view(np.uint32) == gf.view(np.uint32)) | (np.isnan(of) & np.isnan(gf))
            assert same.all() and np.array_equal(o8, g8[i]), (cfgs[i], call)
    assert (g.handoff_calls() > 0) == (handoff == "1")
    assert g.wild_calls() >= 8  # channels 6, 7 (every call), 8 (every call), 9 / 10 (the burst)
    g.close()


def test_hand_off_after_the_creation_time_calibration_starts_from_fresh_stamps(monkeypatch):
    """A batch of 32 channels or more times its own pipeline when it is created and then puts every stream back to its initial
    state -- the call count too.  The hand-off's stamp value of a call used to be that count + 1, so the caller's calls found
    the calibration's stamps of the same value in place: the DC and clock stages took tiles for finished that the front-end had
    not written yet.  Found by the batcher soak of round 5 at seed 90173 (38 clients, filters of up to 932 taps, short buffers,
    rounds launched 300 us after their first buffer): that round again, a dozen times, behind a batcher that calibrates
    (SDRM_BATCHER_CALIBRATE=1; by default a batcher no longer does) -- before the fix one round in three had wrong clients."""
    import threading
    from test_gpu_fuzz import _cases
    monkeypatch.setenv("SDRM_BATCHER_CALIBRATE", "1")
    monkeypatch.setenv("SDRM_AUTOTUNE", "2")  # every question asked, also where the rules decide (round 6): the batch IS timed
    seed = 90173
    for rep in range(12):
        rng = np.random.default_rng(seed)
        maxlen = int(rng.choice([4096, 8192]))
        cfgs = [c + (maxlen,) for c in _cases(seed, int(rng.integers(2, 40)))]
        cfgs = [c for c in cfgs if orc.Fsk(*c).code == 0]
        K = int(rng.integers(2, 8))
        sizes = [[int(rng.choice([1, 17, 500, 3000, maxlen])) for _ in range(K)] for _ in cfgs]
        sigs = [siggen.gmsk_channel(int(rng.integers(0, 1 << 30)), sum(sz), fs=c[0], baud=c[1]) for c, sz in zip(cfgs, sizes)]
        chunks = [[s[sum(sz[:k]):sum(sz[:k + 1])] for k in range(K)] for s, sz in zip(sigs, sizes)]
        bt = binding.Batcher(cfgs, slots=int(rng.integers(3, 7)), max_wait_us=int(rng.choice([300, 2000, 50000])), blocking=True)
        assert bt.code == 0 and len(cfgs) >= 32
        got = [[] for _ in cfgs]

        def producer(c):
            for k in range(K):
                bt.put(c, chunks[c][k])

        def consumer(c):
            for k in range(K):
                got[c].append(bt.take(c))

        th = [threading.Thread(target=f, args=(c,)) for c in range(len(cfgs)) for f in (producer, consumer)]
        for t in th:
            t.start()
        for t in th:
            t.join(120)
            assert not t.is_alive()
        for c, cfg in enumerate(cfgs):
            o = orc.Fsk(*cfg)
            for k in range(K):
                want = o.process(chunks[c][k])[0]
                assert got[c][k] is not None and np.array_equal(got[c][k], want), (rep, c, k, cfg, sizes[c])
        for c in range(len(cfgs)):
            bt.interrupt(c)
        bt.close()


def test_hand_off_stamp_values_start_over_cleanly(monkeypatch):
    """the stamp value of a hand-off call is a 32-bit count (0xfffffff0 values, then over again -- weeks of calls): when it
    starts over the tables are cleared, so that no stamp of the last time round can look like a new call's.  A batch whose count
    starts ten calls before the end (SDRM_HAND_EPOCH0) makes twenty blocking calls across it, ragged, against the oracle."""
    monkeypatch.setenv("SDRM_HAND_EPOCH0", str(0xfffffff0 - 10))
    maxlen = 20000
    cfgs = [(48000, 9600, 5000, 1, 2000, True, maxlen), (48000, 9600, 5000, 1, 2000, False, maxlen),
            (240000, 9600, 5000, 1, 2000, True, maxlen), (48000, 4800, 5000, 2, 2000, True, maxlen)] * 3
    g = binding.Batch(cfgs)
    assert g.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs]
    rng = np.random.default_rng(11)
    sigs = [siggen.gmsk_channel(4000 + i, 20 * maxlen, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    pos = [0] * len(cfgs)
    for call in range(20):
        # long calls early (every tile stamped with the old values), short ones around the turn, long ones after it
        lens = [maxlen if call < 6 or call > 12 else int(rng.choice([300, 5000])) for _ in cfgs]
        parts = [s[p:p + n] for s, p, n in zip(sigs, pos, lens)]
        g8 = g.process(parts)
        for i, o in enumerate(oracles):
            pos[i] += lens[i]
            assert np.array_equal(o.process(parts[i])[0], g8[i]), (call, cfgs[i], lens[i])
    assert g.handoff_calls() == 20
    g.close()


def test_a_plain_handle_takes_the_hand_off_for_long_calls_and_the_graph_replay_for_short_ones():
    """fsk_demod_process on one handle: repeated short calls are a replayed graph of the three stages, calls long enough for the
    overlap to pay run the in-call hand-off on the handle's stream plus two side streams -- every mix of the two, and ragged
    lengths in between, must continue the same stream bit for bit"""
    cfg = (48000, 9600, 5000, 1, 2000, True, 131072)
    d = binding.FskDemod(*cfg)
    o = orc.Fsk(*cfg)
    sig = siggen.gmsk_channel(77, 600000)
    pos = 0
    for n in [4096, 4096, 4096, 131072, 131072, 4096, 4096, 50000, 50000, 50000, 7, 131072, 0, 4096, 4096, 100000]:
        part = sig[pos:pos + n]
        pos += n
        want, _ = o.process(part)
        got = d.process(part)
        assert np.array_equal(got, want), (n, pos, len(got), len(want))
    d.close()


@pytest.mark.parametrize("keep_soft", [False, True])
def test_few_live_channels_in_a_large_batch_keep_their_streams_on_the_device(keep_soft):
    """A 96-slot batch of four kinds of channels with few of them live per call, presence and lengths changing from call to call --
    what a server's batcher sees off-peak.  Since round 5 absent channels no longer send their workgroup's live neighbours to the
    predicated code (DC stage: empty slots become replicas of a live one; clock stage: absent rows and rows that have ended are
    staged along, a lane writes its state back when ITS samples end, before its ring slots are overwritten by the longer rows'
    blocks): profiles/r05_node_schedule.txt.  Every live channel's stream equals the oracle's, across calls it sat out."""
    kinds = [(48000, 9600, 5000, 1, 2000, True), (240000, 19200, 5000, 5, 2000, True), (48000, 1200, 5000, 8, 2000, True),
             (48000, 9600, 5000, 1, 2000, False), (240000, 9600, 5000, 1, 2000, True)]
    maxlen = 20000
    n_ch = 96
    cfgs = [kinds[i % 5] + (maxlen,) for i in range(n_ch)]
    g = binding.Batch(cfgs, keep_soft=keep_soft)
    assert g.code == 0
    oracles = [orc.Fsk(*c) for c in cfgs]
    sigs = {}
    pos = [0] * n_ch
    rng = np.random.default_rng(23)
    patterns = [[0], [0, 1, 2], [95], [3, 4, 19, 40, 41], list(range(n_ch)), [16, 18, 80], [5], [0, 95], list(range(10, 58)), [1, 17, 33, 49, 65, 81]]
    for call, live in enumerate(patterns):
        parts = []
        for i in range(n_ch):
            if i in live:
                if i not in sigs:
                    sigs[i] = siggen.gmsk_channel(1000 + i, 6 * maxlen, fs=cfgs[i][0], baud=cfgs[i][1])
                n = int(rng.choice([maxlen, maxlen, 16384, 4097, 300, 0]))
                parts.append(sigs[i][pos[i]:pos[i] + n])
                pos[i] += n
            else:
                parts.append(binding.ABSENT)
        g8 = g.process(parts)
        for i in range(n_ch):
            if i in live:
                o8, of = oracles[i].process(parts[i])
                assert np.array_equal(o8, g8[i]), (call, i, len(o8), len(g8[i]))
                if keep_soft:
                    assert np.array_equal(of.view(np.uint32), g.last_soft(i).view(np.uint32)), (call, i)
            else:
                assert len(g8[i]) == 0
    g.close()
