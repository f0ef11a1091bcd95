"""Pin the CPU oracle (oracle/sdrm_oracle.c) against everything the reference's own tests hold for the
fsk_demod path: the four end-to-end golden files (test/test_fsk_demod.c), the inline known answers of the
stage unit tests (tests/golden/ref_unit_vectors.json, extracted by make_golden.py) and -- where it builds
without libvolk -- the reference's own code (oracle/_ref).  CPU only.
"""
import json
import os

import numpy as np
import pytest

import orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
VEC = json.load(open(os.path.join(GOLDEN, "ref_unit_vectors.json")))["vectors"]


def ramp(n, off=0):
    """test/utils.c:104-112 setup_input_data"""
    return np.arange(off, off + n, dtype=np.float64).astype(np.float32)


def cramp(n, off=0):
    """test/utils.c:124-132 setup_input_complex_data, interleaved: re = 2(off+i), im = 2(off+i)+1"""
    return np.arange(2 * off, 2 * (off + n), dtype=np.float64).astype(np.float32)


# ---------------------------------------------------------------- end-to-end goldens (test/test_fsk_demod.c)

E2E = [
    # name, create args (fs, baud, dev, decim, tw, dc), input, expected          (test_fsk_demod.c:52-79)
    ("lucky7", (48000, 4800, 5000, 2, 2000, True), "lucky7.expected.cf32", "lucky7.expected.s8"),
    ("lucky7_nodc", (48000, 4800, 5000, 2, 2000, False), "lucky7.expected.cf32", "lucky7.expected.nodc.s8"),
    ("nusat", (192000, 40000, 5000, 1, 2000, True), "nusat.cf32", "processed.s8"),
    ("nan", (240000, 9600, 5000, 1, 2000, True), "inputnan.cf32", "nan.s8"),
]


# Off-by-one soft bits against the committed golden files: exactly the counts SURVEY.md 8(c) measured for the UNMODIFIED
# reference sources with generic VOLK kernels in this container (the golden files were generated on the author's
# machine; the reference compares at +-2, test_fsk_demod.c:47).  Any change of the oracle's arithmetic moves them.
E2E_MISMATCHES = {"lucky7": 22, "lucky7_nodc": 35, "nusat": 1, "nan": 0}


@pytest.mark.parametrize("name,cfg,inp,exp", E2E, ids=[e[0] for e in E2E])
def test_e2e_golden_files(name, cfg, inp, exp):
    iq = np.fromfile(os.path.join(GOLDEN, inp), dtype=np.float32)
    want = np.fromfile(os.path.join(GOLDEN, exp), dtype=np.int8)
    got, _ = orc.demod_stream(cfg, iq, 4096)  # 4096 = the harness's buffer, test_fsk_demod.c:20
    assert len(got) == len(want)
    diff = np.abs(got.astype(np.int32) - want.astype(np.int32))
    # the reference's own tolerance is 2 LSB (test_fsk_demod.c:47); the oracle is within 1, on exactly the symbols the
    # reference itself misses
    assert diff.max() <= 1, (name, int(diff.max()))
    assert int((diff != 0).sum()) == E2E_MISMATCHES[name], (name, int((diff != 0).sum()))


# ---------------------------------------------------------------- second, independent restatement (numpy fp32)

def _synthetic(fs, baud, n, seed):
    import sdrm_pkg
    sdrm_pkg.load()
    from sdr_modem_amd import siggen
    return siggen.gmsk_channel(seed, n, fs=fs, baud=baud).view(np.float32)


NP_CASES = [
    # name, config, input file or (fs, baud, samples, seed) of a synthetic GMSK stream, chunking of the C oracle
    ("lucky7_9600_d1", (48000, 9600, 5000, 1, 2000, True), "lucky7.expected.cf32", 4096),
    ("lucky7_4800_d2", (48000, 4800, 5000, 2, 2000, True), "lucky7.expected.cf32", 4096),
    ("lucky7_4800_d2_nodc", (48000, 4800, 5000, 2, 2000, False), "lucky7.expected.cf32", 1000),
    ("nusat", (192000, 40000, 5000, 1, 2000, True), "nusat.cf32", 4096),
    ("config5_240k_19200_d5", (240000, 19200, 5000, 5, 2000, True), (240000, 19200, 60000, 11), 7777),
    ("config5_48k_1200_d8", (48000, 1200, 5000, 8, 2000, True), (48000, 1200, 50000, 12), 4096),
    ("bench_48k_9600", (48000, 9600, 5000, 1, 2000, True), (48000, 9600, 40000, 13), 131072),
]


@pytest.mark.parametrize("name,cfg,src,chunk", NP_CASES, ids=[c[0] for c in NP_CASES])
def test_numpy_restatement_agrees_bit_for_bit(name, cfg, src, chunk):
    """tests/np_oracle.py is written from SURVEY.md Appendix B and processes the whole stream at once; the C oracle is
    written from the reference's stage files and works chunk by chunk with carried state.  Their FLOAT soft bits must be
    the same bits (this also re-proves chunk invariance for samples per symbol < 8)."""
    import np_oracle
    iq = np.fromfile(os.path.join(GOLDEN, src), dtype=np.float32) if isinstance(src, str) else _synthetic(*src)
    a8, af = np_oracle.demod_stream(cfg, iq)
    b8, bf = orc.demod_stream(cfg, iq, chunk)
    assert len(af) == len(bf) and len(af) > 100
    assert np.array_equal(af.view(np.uint32), bf.view(np.uint32)), name
    assert np.array_equal(a8, b8), name


def test_two_golden_files_are_reproduced_byte_for_byte_with_a_simd_summation_order():
    """Where do the golden files' +-1 LSB against the generic summation order (22 / 35 / 1 bytes above) come from?  From the
    order alone: the same restatement with its two FIRs summing the way a 4-lane SIMD dot product does (four partial sums, then
    left to right; libvolk's tuned kernels -- the reference's CI forces the generic ones, its author's machine did not)
    reproduces `lucky7.expected.nodc.s8` (9603 soft bits) and `processed.s8` (1064) BYTE FOR BYTE.  Every other stage of the
    restatement -- discriminator, arctangent, DC blocker (nusat runs with it), interpolator, timing loop, int8 conversion --
    is thereby pinned exactly by two of the reference's own files, not within a tolerance.  (The third file, lucky7 with the
    DC blocker, stays within 3 bytes over 49 order combinations: four cascaded running sums keep every rounding difference
    of their input for ever, so it needs the exact kernel of the machine that wrote it.)"""
    import np_oracle
    for cfg, inp, exp in (((48000, 4800, 5000, 2, 2000, False), "lucky7.expected.cf32", "lucky7.expected.nodc.s8"),
                          ((192000, 40000, 5000, 1, 2000, True), "nusat.cf32", "processed.s8")):
        iq = np.fromfile(os.path.join(GOLDEN, inp), dtype=np.float32)
        want = np.fromfile(os.path.join(GOLDEN, exp), dtype=np.int8)
        got, _ = np_oracle.demod_stream(cfg, iq, simd_order=(4, 1))
        assert np.array_equal(got, want), (exp, int(np.sum(got[:len(want)] != want[:len(got)])))


def test_numpy_restatement_tables_and_taps():
    import np_oracle
    # the arctan table is atan(i / 255) through "%.6e", its last two entries equal (fast_atan2f.c:23-67)
    formula = np.array([float("%.6e" % np.arctan(min(i, 255) / 255.0)) for i in range(257)], dtype=np.float32)
    assert np.array_equal(formula, np_oracle.ATAN)
    # its tap design equals the reference's known answer (test/test_lpf_taps.c:28-41) and the C oracle's, bit for bit
    v = VEC["test_lpf_taps.c:expected_taps"]
    taps = np_oracle.design_taps(8000, 1750, 500)
    assert np.abs(taps - np.array(v["values"], dtype=np.float32)).max() < 1e-4
    for fs, fc, tw in ((48000, 9800, 980), (48000, 4800, 2000), (240000, 14600, 1460), (192000, 20000, 2000)):
        assert np.array_equal(np_oracle.design_taps(fs, fc, tw).view(np.uint32), orc.lowpass_taps(fs, fc, tw)[1].view(np.uint32))


@pytest.mark.parametrize("chunk", [1000, 2000, 4096, 96000])
def test_e2e_chunk_invariance_sps_lt_8(chunk):
    """sps < 8 => any chunking gives the same stream (SURVEY finding 3)."""
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.float32)
    cfg = (48000, 4800, 5000, 2, 2000, True)
    a8, af = orc.demod_stream(cfg, iq, 4096)
    b8, bf = orc.demod_stream(cfg, iq, chunk)
    assert np.array_equal(a8, b8)
    assert np.array_equal(af.view(np.uint32), bf.view(np.uint32))


def test_e2e_oversize_input_gives_no_output(capfd):
    d = orc.Fsk(48000, 4800, 5000, 2, 2000, True, 100)
    assert d.code == 0
    out, soft = d.process(cramp(101))
    assert len(out) == 0
    assert "<3>requested buffer 101 is more than max: 100" in capfd.readouterr().err


def test_create_errors():
    # cutoff above fs/2 => -1, as test/test_dsp_worker.c baud==fs case (lpf_taps.c:20-23)
    assert orc.Fsk(48000, 48000, 5000, 1, 2000, True, 4096).code == -1
    assert orc.Fsk(0, 4800, 5000, 1, 2000, True, 4096).code == -1
    assert orc.Fsk(48000, 4800, 5000, 1, 0, True, 4096).code == -1


# ---------------------------------------------------------------- stage known answers

def test_lpf_taps_known_answer():
    v = VEC["test_lpf_taps.c:expected_taps"]
    code, taps = orc.lowpass_taps(8000, 1750, 500)
    assert code == 0 and len(taps) == 39
    want = np.array(v["values"], dtype=np.float32)
    assert np.array_equal((want * 10000).astype(np.int32), (taps * 10000).astype(np.int32))


def test_lpf_taps_bounds():
    # test/test_lpf_taps.c:7-26
    assert orc.lowpass_taps(0, 1750, 500)[0] == -1
    assert orc.lowpass_taps(8000, 0, 500)[0] == -1
    assert orc.lowpass_taps(8000, 4001, 500)[0] == -1
    assert orc.lowpass_taps(8000, 1750, 0)[0] == -1


def test_lpf_complex_two_calls():
    f = orc.Fir(1, 48000, 4800, 2000, 2000, 2)
    x = cramp(500)
    for lab, sl in (("complex_call1", slice(0, 500)), ("complex_call2", slice(500, 1000))):
        v = VEC["test_lpf.c:" + lab]
        got = f.process(x[sl])
        want = np.array(v["values"], dtype=np.float32)
        assert len(got) == len(want)
        assert np.abs(got.astype(np.float64) - want).max() < v["tolerance"]


def test_lpf_float_decim2_two_calls():
    f = orc.Fir(2, 48000, 4800, 2000, 2000, 1)
    x = ramp(1000)
    for lab, sl in (("float_call1", slice(0, 500)), ("float_call2", slice(500, 1000))):
        v = VEC["test_lpf.c:" + lab]
        got = f.process(x[sl])
        want = np.array(v["values"], dtype=np.float32)
        assert len(got) == len(want)
        assert np.abs(got.astype(np.float64) - want).max() < v["tolerance"]


def test_lpf_small_buffer_counts():
    v = VEC["test_lpf.c:small_buffer"]
    f = orc.Fir(2, 48000, 4800, 2000, 2000, 2)
    x = cramp(500)
    lens = [len(f.process(x[0:0])) // 2, len(f.process(x[0:2])) // 2, len(f.process(x[2:4])) // 2]
    last = f.process(x[4:6])
    lens.append(len(last) // 2)
    assert lens == v["output_lens"]
    assert abs(last[0] - v["last_value"][0]) < 1e-3 and abs(last[1] - v["last_value"][1]) < 1e-3


def test_lpf_big_buffer(capfd):
    f = orc.Fir(2, 48000, 4800, 2000, 100, 2)
    assert len(f.process(cramp(101))) == 0
    assert "more than max" in capfd.readouterr().err


def test_quadrature_demod_known_answer():
    q = orc.Quad(25.4, 2000)
    x = cramp(200)
    got1 = q.process(x[:4])
    got2 = q.process(x[4:])
    w1 = np.array(VEC["test_quadrature_demod.c:expected"]["values"])
    w2 = np.array(VEC["test_quadrature_demod.c:expected2"]["values"])
    assert len(got1) == 2 and len(got2) == 198
    assert np.abs(got1 - w1).max() < 1e-3
    assert np.abs(got2 - w2).max() < 1e-3


def test_dc_blocker_known_answer():
    d = orc.Dc(32)
    got = d.process(ramp(200))
    want = np.array(VEC["test_dc_blocker.c:expected"]["values"])
    assert np.abs(got - want).max() < 1e-3


def test_mmse_known_answer():
    got = orc.mmse_interp(ramp(8), 0, 0.14)
    assert abs(got - VEC["test_mmse_fir_interpolator.c:normal"]["values"][0]) < 1e-3


def test_mmse_alignment_lead_is_neutral_for_finite_data_and_nan_otherwise():
    x = np.random.default_rng(1).standard_normal(32).astype(np.float32)
    base = orc.mmse_interp(x[5:], 0, 0.3)  # window x[5:13], no leading samples
    assert orc.mmse_interp(x, 5, 0.3) == base  # idx 5 -> one leading sample times 0
    y = x.copy()
    y[4] = np.inf
    assert np.isnan(orc.mmse_interp(y, 5, 0.3))  # reference fir_filter.c:116-121 multiplies it by a zero tap


def test_clock_recovery_known_answer():
    c = orc.Clock(2.0, 0.25 * 0.175 * 0.175, 0.005, 0.175, 0.005, 100)
    x = ramp(100)
    got1 = c.process(x[:42])
    got2 = c.process(x[42:78])
    w1 = np.array(VEC["test_clock_recovery_mm.c:expected"]["values"])
    w2 = np.array(VEC["test_clock_recovery_mm.c:expected2"]["values"])
    assert len(got1) == len(w1) and len(got2) == len(w2)
    assert np.abs(got1 - w1).max() < 1e-3
    assert np.abs(got2 - w2).max() < 1e-3


def test_clock_recovery_small_buffers():
    c = orc.Clock(2.0, 0.25 * 0.175 * 0.175, 0.005, 0.175, 0.005, 100)
    x = ramp(100)
    lens = [len(c.process(x[0:0])), len(c.process(x[0:4])), len(c.process(x[4:7])), len(c.process(x[7:8]))]
    assert lens == VEC["test_clock_recovery_mm.c:small_buffers"]["output_lens"]


def test_clock_recovery_big_buffer(capfd):
    c = orc.Clock(2.0, 0.25 * 0.175 * 0.175, 0.005, 0.175, 0.005, 100)
    assert len(c.process(ramp(101))) == 0
    assert "more than max" in capfd.readouterr().err


def test_nco_known_answer():
    s = orc.Nco(1.0, 4, 4)
    got = s.process(1, 4)
    want = np.array(VEC["test_sig_source.c:success"]["values"], dtype=np.float32)
    assert np.abs(got - want).max() < 1e-2


# ---------------------------------------------------------------- next row f-1: Doppler batching + NCO (test/test_doppler.c)

DOPPLER = json.load(open(os.path.join(GOLDEN, "doppler_shifts_lucky7.json")))


@pytest.mark.parametrize("buflen", [2000, 47000, 95000])
def test_doppler_rx_golden_file(buflen):
    """test/test_doppler.c:37-76: lucky7.cf32 -> lucky7.expected.cf32 (the .47000/.95000 expected files are
    byte-identical to it), fed in 2000-sample reads, tolerance 0.01 (test/utils.c:137)."""
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.cf32"), dtype=np.float32)
    want = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.float32)
    d = orc.Doppler(DOPPLER["sampling_freq"], DOPPLER["shifts_hz"], buflen)
    assert d.code == 0
    got = np.concatenate([d.process(iq[2 * o: 2 * (o + 2000)]) for o in range(0, len(iq) // 2, 2000)])
    assert len(got) == len(want)
    assert np.abs(got - want).max() < 0.01
    assert np.abs(got - want).max() < 2e-6  # in fact equal to float rounding of the stored file


def test_doppler_batches_split_at_second_boundaries():
    d = orc.Doppler(48000, DOPPLER["shifts_hz"], 131072)
    plan = d.plan(131072)
    assert [p[0] for p in plan] == [48000, 48000, 35072]
    assert plan[0][1] == int(DOPPLER["shifts_hz"][0]) and plan[1][1] == int(DOPPLER["shifts_hz"][1])
    # a call that ends inside a second: the next call first finishes that second with the interpolated shift
    nxt = d.plan(20000)
    assert [p[0] for p in nxt] == [12928, 7072]


def test_doppler_invalid_arguments(capfd):
    d = orc.Doppler(48000, DOPPLER["shifts_hz"], 2000)
    assert len(d.process(np.zeros(0, np.float32))) == 0
    assert len(d.process(np.zeros(2 * 2001, np.float32))) == 0
    assert "more than max: 2000" in capfd.readouterr().err


# ---------------------------------------------------------------- against the reference's own code (oracle/_ref)

# (decided on the file's presence, not by loading it: collection must not map anything built from /root/reference into a process
# that is about to run the GPU suite -- oracle/_ref stays in the container where the reference tree is, .gpurunignore)
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(orc.ORC_DIR, "_ref", "libsdrm_ref.so")),
                               reason="oracle/_ref not built (reference tree absent)")

CONFIG_FILTERS = [
    (48000, 9800, 980), (48000, 4800, 2000), (48000, 7400, 740), (48000, 2400, 2000),
    (240000, 14600, 1460), (240000, 9600, 2000), (48000, 5600, 560), (48000, 600, 2000),
    (192000, 25000, 2500), (192000, 20000, 2000), (240000, 9800, 980), (240000, 4800, 2000), (8000, 1750, 500),
]


@needs_ref
@pytest.mark.parametrize("fs,fc,tw", CONFIG_FILTERS)
def test_taps_bit_identical_to_reference_code(fs, fc, tw):
    c1, a = orc.lowpass_taps(fs, fc, tw)
    c2, b = orc.ref_lowpass_taps(fs, fc, tw)
    assert c1 == c2 == 0 and len(a) == len(b)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("length", [2, 3, 32, 80, 154, 160])
def test_dc_blocker_bit_identical_to_reference_code(length):
    rng = np.random.default_rng(length)
    x = (rng.standard_normal(3000) * 3 + 0.7).astype(np.float32)
    x[100] = 1e-41  # denormal
    a, b = orc.Dc(length), orc.RefDc(length)
    for sl in (slice(0, 1), slice(1, 700), slice(700, 3000)):
        ga, gb = a.process(x[sl]), b.process(x[sl])
        assert np.array_equal(ga.view(np.uint32), gb.view(np.uint32))


@needs_ref
def test_dc_blocker_nan_poisons_like_reference_code():
    x = np.ones(500, dtype=np.float32)
    x[50] = np.nan
    ga, gb = orc.Dc(32).process(x), orc.RefDc(32).process(x)
    assert np.array_equal(np.isnan(ga), np.isnan(gb))
    assert np.array_equal(ga[~np.isnan(ga)].view(np.uint32), gb[~np.isnan(gb)].view(np.uint32))


@needs_ref
def test_fast_atan2f_bit_identical_to_reference_code():
    rng = np.random.default_rng(7)
    ys = np.concatenate([rng.standard_normal(20000), [0, 0, 1, -1, 1e-3, -1e-9, 0.0, -0.0, np.inf, np.nan, 1e-42, 3e38]])
    xs = np.concatenate([rng.standard_normal(20000), [0, -1, 1, -1, 1, -1, -0.0, 0.0, 1.0, 1.0, 1e-40, 3e38]])
    # ratios straddling the small-angle threshold and the table knots
    k = np.arange(1, 256, dtype=np.float64) / 255.0
    ys = np.concatenate([ys, k, k * (1 + 1e-7), k * (1 - 1e-7), [0.003921569, 0.0039215689, 0.00392157]])
    xs = np.concatenate([xs, np.ones(3 * 255 + 3)])
    L, R = orc.lib(), orc.ref_lib()
    for y, x in zip(ys.astype(np.float32), xs.astype(np.float32)):
        a, b = np.float32(L.orc_fast_atan2f(y, x)), np.float32(R.fast_atan2f(y, x))
        assert a.view(np.uint32) == b.view(np.uint32) or (np.isnan(a) and np.isnan(b)), (y, x, a, b)


def test_simd_stand_in_build_is_a_timing_double_not_a_checker():
    """oracle/libsdrm_oracle_tuned.so (bench.py's second CPU figure) is the oracle's source with vectorised dot products:
    its soft bits stay within the reference's own +-2 LSB tolerance of the pinned build on the lucky7 recording (so its
    timing is of the same computation); it is never used as the checker (different summation order)."""
    T = orc.tuned_lib()
    if T is None:
        pytest.skip("no AVX2/FMA here or the tuned build is missing")
    import ctypes as C
    T.orc_fsk_create.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_uint8, C.c_uint32, C.c_bool, C.c_uint32,
                                 C.POINTER(C.c_void_p)]
    T.orc_fsk_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_int8)), C.POINTER(C.c_size_t)]
    T.orc_fsk_process.restype = None
    T.orc_fsk_destroy.argtypes = [C.c_void_p]
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.float32)
    h = C.c_void_p()
    assert T.orc_fsk_create(48000, 4800, 5000, 2, 2000, True, len(iq) // 2, C.byref(h)) == 0
    p, n = C.POINTER(C.c_int8)(), C.c_size_t()
    T.orc_fsk_process(h, iq.ctypes.data, len(iq) // 2, C.byref(p), C.byref(n))
    got = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    T.orc_fsk_destroy(h)
    want, _ = orc.Fsk(48000, 4800, 5000, 2, 2000, True, len(iq) // 2).process(iq)
    assert len(got) == len(want)
    assert np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 2
