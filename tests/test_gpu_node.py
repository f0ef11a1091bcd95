"""`-m gpu`: the in-process node front door (sdrm_node_*, sdr-modem_amd/host/node.cpp) on the device.  The pool's boxes have
one GPU, so the node is given TWO batchers on device 0 -- the same code path as two devices (placement, one batcher thread
and one device batch per batcher, per-batcher error), with the two batches sharing the chip.  Reference process model:
src/tcp_server.c:659, src/sdr_worker.c:25-55, src/dsp_worker.c:188."""
import ctypes as C
import os
import subprocess
import tempfile
import threading

import numpy as np
import pytest

import orc
import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd import binding, siggen  # noqa: E402

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert binding.load().sdrm_device_count() > 0, "these tests need an MI355X; the library has no CPU path"


def test_twenty_four_clients_behind_one_node_handle_on_two_batchers():
    """a node of two batchers x 20 slots on device 0: 24 RX clients of three kinds (one kind over ten times as costly) created
    with sdrm_worker_config.node, fed from per-client threads, files against the oracle; twice the slots of one batcher run
    through one handle, and the placement balances cost, not count."""
    L = binding.load()
    geom = (48000, 9600, 5000, 1, 2000, True, 8192)
    node = binding.Node(geom, 20, n_batchers=2, devices=[0, 0], batcher=(4, 50000, True))
    assert node.code == 0 and node.batchers() == 2
    kinds = [(48000, 9600, 5000, 1, 2000, True, 8192), (48000, 4800, 5000, 2, 2000, False, 8192),
             (240000, 19200, 5000, 5, 2000, True, 8192)]
    n_w, sizes = 24, [8192, 3000, 8192, 17, 8000]
    cfgs = [kinds[(i * 7) % 3] for i in range(n_w)]
    sigs = [siggen.gmsk_channel(500 + i, sum(sizes), fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    with tempfile.TemporaryDirectory() as tmp:
        ws = []
        for i in range(n_w):
            c = cfgs[i]
            wc = binding.WorkerConfig(c[0], c[1], c[2], c[3], c[4], c[5], False, 0, c[6], 8, True, tmp.encode(),
                                      None, None, None, 0, node.h, 1 + i % 4, 0)
            w = C.c_void_p()
            assert L.dsp_worker_create(700 + i, -1, C.byref(wc), C.byref(w)) == 0
            ws.append(w)
        st = [node.stat(0), node.stat(1)]
        assert st[0].clients + st[1].clients == n_w and st[0].device == 0 and st[1].device == 0
        assert max(st[0].load, st[1].load) - min(st[0].load, st[1].load) <= binding.channel_cost(kinds[2]) * (1 + 1e-9)
        # every slot of both batchers can be used through the one handle; with all 40 taken the next client is refused, not queued
        wc = binding.WorkerConfig(48000, 9600, 5000, 1, 2000, True, False, 0, 8192, 8, True, tmp.encode(),
                                  None, None, None, 0, node.h, 0, 0)
        idle = []
        for k in range(16):
            w = C.c_void_p()
            assert L.dsp_worker_create(900 + k, -1, C.byref(wc), C.byref(w)) == 0
            idle.append(w)
        w = C.c_void_p()
        assert L.dsp_worker_create(999, -1, C.byref(wc), C.byref(w)) == -16  # -EBUSY
        assert node.stat(0).clients == 20 and node.stat(1).clients == 20
        for w in idle:
            L.dsp_worker_destroy(w)  # sixteen clients that never sent a buffer leave: their slots are free again

        def feed(i):
            pos = 0
            for n in sizes:
                part = np.ascontiguousarray(sigs[i][pos:pos + n]).view(np.float32)
                L.dsp_worker_put(part.ctypes.data, n, ws[i])
                pos += n
        th = [threading.Thread(target=feed, args=(i,), daemon=True) for i in range(n_w)]
        for t in th:
            t.start()
        for t in th:
            t.join(180)
            assert not t.is_alive()
        for w in ws:
            L.dsp_worker_destroy(w)
        assert node.stat(0).clients == 0 and node.stat(1).clients == 0 and node.stat(0).error == 0 and node.stat(1).error == 0
        for i in range(n_w):
            got = np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % (700 + i)), dtype=np.int8)
            o = orc.Fsk(*cfgs[i])
            pos, want = 0, []
            for n in sizes:
                want.append(o.process(sigs[i][pos:pos + n])[0])
                pos += n
            assert np.array_equal(got, np.concatenate(want)), i
    node.close()


def _file_demod():
    exe = os.path.join(ROOT, "tools", "file_demod")
    src = os.path.join(ROOT, "tools", "file_demod.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-pthread", src, "-I" + os.path.join(ROOT, "include"),
                               "-L" + os.path.join(ROOT, "sdr-modem_amd", "csrc"), "-lsdrmodem_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "sdr-modem_amd", "csrc"), "-o", exe])
    return exe


def test_file_source_harness_through_a_node_of_two_batchers():
    """tools/file_demod -g 2 -n 5: a plain C program, one node handle, five workers placed over two batchers; every
    worker's rx.demod2client.<id>.s8 is the oracle's stream, within the reference's 2 LSB of its golden file"""
    src = os.path.join(GOLDEN, "lucky7.expected.cf32")
    with tempfile.TemporaryDirectory() as tmp:
        out = subprocess.run([_file_demod(), "-n", "5", "-g", "2", src, tmp, "48000", "4800", "5000", "2", "2000", "1"],
                             timeout=180, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-1500:]
        served = sorted(int(ln.split("served")[1].split()[0]) for ln in out.stderr.splitlines() if "served" in ln)
        assert served == [2, 3], out.stderr[-600:]
        iq = np.fromfile(src, dtype=np.complex64)
        want, _ = orc.demod_stream((48000, 4800, 5000, 2, 2000, True), iq, 4096)
        golden = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.s8"), dtype=np.int8)
        for i in range(5):
            got = np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % i), dtype=np.int8)
            assert np.array_equal(got, want), i
            assert len(got) == len(golden) and np.abs(got.astype(np.int32) - golden.astype(np.int32)).max() <= 2


@pytest.mark.parametrize("layout", [["-n", "1"], ["-n", "3"], ["-n", "3", "-g", "2"]], ids=["private", "batcher", "node"])
def test_file_source_frequency_offset_on_the_device(layout):
    """RxRequest.rx_offset as src/sdr/file_source.c:120-128 applies it -- sig_source_multiply(offset) on every buffer read,
    phase carried -- done by the device's NCO in front of the demodulator (tools/file_demod -o): a recording moved up by
    1200 Hz and read with -o -1200 gives, bit for bit, orc.Nco(-1200) + orc.Fsk on the same buffers."""
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.expected.cf32"), dtype=np.complex64)
    n = np.arange(len(iq))
    moved = (iq * np.exp(2j * np.pi * 1200.0 * n / 48000.0)).astype(np.complex64)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "moved.cf32")
        moved.tofile(path)
        subprocess.check_call([_file_demod(), "-o", "-1200"] + layout + [path, tmp, "48000", "4800", "5000", "2", "2000", "1"], timeout=180)
        osc = orc.Nco(1.0, 48000, 4096)
        o = orc.Fsk(48000, 4800, 5000, 2, 2000, True, 4096)
        want = np.concatenate([o.process(osc.multiply(-1200, moved[k:k + 4096].view(np.float32)))[0] for k in range(0, len(moved), 4096)])
        for i in range(int(layout[1])):
            got = np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % i), dtype=np.int8)
            assert np.array_equal(got, want), i
        # and the recording is demodulated: same symbols as the un-moved file up to the oscillator's rounding (hard bits)
        plain, _ = orc.demod_stream((48000, 4800, 5000, 2, 2000, True), iq, 4096)
        assert len(plain) == len(want) and np.mean((plain >= 0) == (want >= 0)) > 0.995


def test_reference_signature_adapter_on_the_device(tmp_path):
    """integration/dsp_worker_ref.c -- dsp_worker_create with the REFERENCE'S parameter list (src/dsp_worker.h:22: id, socket,
    struct server_config *, struct RxRequest *, &worker) -- built here and driven on the device (until round 5 only over the
    kernel emulation): (1) workers that own a private demodulator, as the reference lays them out, one of them dumping its IQ;
    (2) workers placed by a node of two batchers on this device; (3) a worker with RxRequest.doppler set, the predictor built
    by a factory installed with sdrm_ref_set_doppler_factory (a synthetic one here: the shipped SGP4 factory is host-only and is
    pinned in the CPU suite).  The files the reference's worker writes
    (rx.demod2client.<id>.s8, rx.sdr2demod.<id>.cf32; src/dsp_worker.c:154,165) hold the oracle's bytes."""
    import test_abi_cpu as T
    A = T._ref_adapter(tmp_path)
    A.dsp_worker_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(T._ServerConfig), C.POINTER(T._RxRequest), C.POINTER(C.c_void_p)]
    A.sdrm_ref_attach_node.argtypes = [C.c_void_p]
    L = binding.load()
    L.dsp_worker_put.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.dsp_worker_destroy.argtypes = [C.c_void_p]
    cfg = (48000, 4800, 5000, 2, 2000, True, 4096)
    sc = T._ServerConfig(buffer_size=4096, queue_size=4, rx_sdr_type=2, base_path=str(tmp_path).encode())

    def run(ids, make_request, sigs):
        ws = []
        for k, i in enumerate(ids):
            req = make_request(k)
            w = C.c_void_p()
            assert A.dsp_worker_create(i, -1, C.byref(sc), C.byref(req), C.byref(w)) == 0, i
            ws.append(w)

        def feed(k):
            for off in range(0, len(sigs[k]), 4096):
                part = np.ascontiguousarray(sigs[k][off:off + 4096]).view(np.float32)
                L.dsp_worker_put(part.ctypes.data, len(part) // 2, ws[k])
        th = [threading.Thread(target=feed, args=(k,)) for k in range(len(ids))]
        for t in th:
            t.start()
        for t in th:
            t.join(120)
            assert not t.is_alive()
        for w in ws:
            L.dsp_worker_destroy(w)
        return [np.fromfile(os.path.join(str(tmp_path), "rx.demod2client.%d.s8" % i), dtype=np.int8) for i in ids]

    # (1) private demodulators
    sigs = [siggen.gmsk_channel(600 + k, 5 * 4096 + 77, fs=48000, baud=4800) for k in range(3)]
    got = run([801, 802, 803], lambda k: T._request(48000, 4800, 5000, 2, 2000, True, k == 1, 0), sigs)
    for k in range(3):
        assert np.array_equal(got[k], orc.demod_stream(cfg[:6], sigs[k], 4096)[0]), k
    assert np.array_equal(np.fromfile(os.path.join(str(tmp_path), "rx.sdr2demod.802.cf32"), dtype=np.complex64), sigs[1])
    # (2) placed by a node
    node = binding.Node(cfg, 4, n_batchers=2, devices=[0, 0], batcher=(4, 50000, True))
    assert node.code == 0
    A.sdrm_ref_attach_node(node.h)

    def node_request(k):
        req = T._request(48000, 4800, 5000, 2, 2000, True, False, 0)
        req.rx_center_freq = 437525000 if k % 2 == 0 else 145800000
        return req
    sigs = [siggen.gmsk_channel(610 + k, 3 * 4096 + 5, fs=48000, baud=4800) for k in range(4)]
    got = run([811, 812, 813, 814], node_request, sigs)
    for k in range(4):
        assert np.array_equal(got[k], orc.demod_stream(cfg[:6], sigs[k], 4096)[0]), k
    assert [node.stat(d).clients for d in range(2)] == [0, 0]
    A.sdrm_ref_attach_node(None)
    node.close()
    # (3) Doppler from the request, through a factory (sdrm_ref_set_doppler_factory).  The orbit model is host-only and the shipped
    # factory (integration/doppler_factory_ref.c on the reference's src/sgpsdp) is pinned against the reference's shifts in the
    # CPU suite (test_abi_cpu.py::test_shipped_doppler_factory_returns_the_references_shifts) -- nothing built from the
    # reference's sources travels to the GPU box.  Here a synthetic factory hands out the same station's recorded shifts
    # (tests/golden/doppler_shifts_lucky7.json): what is under test is the request -> factory -> predictor -> device NCO path.
    import json
    recorded = json.load(open(os.path.join(GOLDEN, "doppler_shifts_lucky7.json")))["shifts_hz"]
    shift_cb = binding.SHIFT_FN(lambda user, k: float(recorded[min(int(k), len(recorded) - 1)]))
    FACTORY = C.CFUNCTYPE(C.c_int, C.POINTER(T._RxRequest), C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p))
    asked = []

    def factory(req, cfg_, fn_out, user_out):
        asked.append((bool(req.contents.doppler), bool(req.contents.file_settings)))
        fn_out[0] = C.cast(shift_cb, C.c_void_p)
        user_out[0] = None
        return 0
    factory_cb = FACTORY(factory)
    A.sdrm_ref_set_doppler_factory.argtypes = [C.c_void_p]
    A.sdrm_ref_set_doppler_release.argtypes = [C.c_void_p]
    A.sdrm_ref_set_doppler_factory(C.cast(factory_cb, C.c_void_p))
    A.sdrm_ref_set_doppler_release(None)
    lines = (C.c_char_p * 3)(*T.LUCKY7_TLE)
    ds = T._DopplerSettings(n_tle=3, tle=lines, latitude=537200000, longitude=475700000, altitude=0)
    fs_ = T._FileSettings(filename=b"x.cf32", start_time_seconds=1583840449)

    def doppler_request(k):
        req = T._request(48000, 4800, 5000, 2, 2000, True, False, 0)
        req.doppler = C.cast(C.pointer(ds), C.c_void_p)
        req.file_settings = C.cast(C.pointer(fs_), C.c_void_p)
        return req
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.cf32"), dtype=np.complex64)
    got = run([821], doppler_request, [iq])
    assert asked == [(True, True)]
    # the oracle's Doppler block with the same shifts, then its demodulator
    o = orc.Fsk(*cfg)
    d = orc.Doppler(48000, recorded, 4096)
    want = np.concatenate([o.process(d.process(iq[off:off + 4096].view(np.float32)))[0] for off in range(0, len(iq), 4096)])
    assert np.array_equal(got[0], want)
    A.sdrm_ref_set_doppler_factory(None)


@pytest.mark.parametrize("placed", ["private", "node"])
def test_file_source_offset_and_doppler_correction_in_series_on_the_device(placed):
    """RxRequest.rx_offset AND RxRequest.doppler: the reference runs two oscillators in series, each with a phase of its own and
    every sample rounded to fp32 in between -- the file source's (src/sdr/file_source.c:120-128), then the Doppler correction's
    (src/dsp_worker.c:65-71, src/dsp/doppler.c:116-190).  -ENOTSUP until round 5; now the batch's pre-offset oscillator in front
    of the NCO batches (sdrm_batch_set_pre_offset).  Bit for bit orc.Nco -> orc.Doppler -> orc.Fsk over the reference's
    recording, across one-second boundaries, as a private worker and on a node slot beside a plain client."""
    L = binding.load()
    L.dsp_worker_put.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.dsp_worker_destroy.argtypes = [C.c_void_p]
    import json
    shifts = json.load(open(os.path.join(GOLDEN, "doppler_shifts_lucky7.json")))["shifts_hz"]
    cb = binding.SHIFT_FN(lambda user, k: float(shifts[min(int(k), len(shifts) - 1)]))
    iq = np.fromfile(os.path.join(GOLDEN, "lucky7.cf32"), dtype=np.complex64)
    cfg = (48000, 4800, 5000, 2, 2000, True, 4096)
    node = binding.Node(cfg, 4, n_batchers=2, devices=[0, 0], batcher=(4, 50000, True)) if placed == "node" else None
    with tempfile.TemporaryDirectory() as tmp:
        def worker(i, offset, doppler):
            wc = binding.WorkerConfig(48000, 4800, 5000, 2, 2000, True, False, 0, 4096, 4, True, tmp.encode(),
                                      C.cast(cb, C.c_void_p) if doppler else None, None, None, 0, node.h if node else None, 1, offset)
            w = C.c_void_p()
            assert L.dsp_worker_create(i, -1, C.byref(wc), C.byref(w)) == 0
            return w
        ws = [worker(41, -1200, True), worker(42, 0, False), worker(43, 700, False)]
        for off in range(0, len(iq), 4096):
            part = np.ascontiguousarray(iq[off:off + 4096]).view(np.float32)
            for w in ws:
                L.dsp_worker_put(part.ctypes.data, len(part) // 2, w)
        for w in ws:
            L.dsp_worker_destroy(w)
        got = [np.fromfile(os.path.join(tmp, "rx.demod2client.%d.s8" % i), dtype=np.int8) for i in (41, 42, 43)]
    parts = [iq[off:off + 4096].view(np.float32) for off in range(0, len(iq), 4096)]
    osc, dop, o = orc.Nco(1.0, 48000, 4096), orc.Doppler(48000, shifts, 4096), orc.Fsk(*cfg)
    assert np.array_equal(got[0], np.concatenate([o.process(dop.process(osc.multiply(-1200, p)))[0] for p in parts]))
    assert np.array_equal(got[1], orc.demod_stream(cfg[:6], iq, 4096)[0])
    osc, o = orc.Nco(1.0, 48000, 4096), orc.Fsk(*cfg)
    assert np.array_equal(got[2], np.concatenate([o.process(osc.multiply(700, p))[0] for p in parts]))
    if node:
        node.close()
