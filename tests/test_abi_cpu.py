"""CPU-only: the C-ABI library builds for gfx950, loads without a GPU, exports every symbol the public header
declares, reports parameter errors like the reference and refuses to compute without a device (no CPU fallback).
Also the host-only queue, mirrored from the reference's test/test_queue.c."""
import ctypes as C
import errno
import os
import re
import subprocess
import threading
import time

import numpy as np
import pytest

import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd import binding  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "sdrmodem_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef[^;{]*\(\s*\*\s*\w+\s*\)[^;]*;", "", text)  # function-pointer typedefs are not exports
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_exports_every_declared_symbol():
    L = binding.load()
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), "declared in include/sdrmodem_hip.h but not exported: " + name
    assert sorted(declared) == sorted(binding.EXPORTS)


def test_version_and_device_count():
    L = binding.load()
    assert b"gfx950" in L.sdrm_version()
    assert L.sdrm_device_count() >= 0


def test_bad_parameters_are_reported_before_touching_the_device():
    # reference test/test_dsp_worker.c:58-66: baud == fs => cutoff above fs/2 => -1
    assert binding.Batch([(48000, 48000, 5000, 1, 2000, True, 4096)]).code == -1
    assert binding.FskDemod(48000, 48000, 5000, 1, 2000, True, 4096).code == -1
    assert binding.Batch([(48000, 4800, 5000, 1, 0, True, 4096)]).code == -1
    # any samples per symbol the reference accepts (fsk_demod.c:53-63) is accepted since round 4 (400 here: generic DC and clock
    # stages); what is still refused is a filter that cannot fit a tile (48 MHz / 1200 baud: 206 k taps)
    assert binding.Batch([(48000, 1200, 5000, 1, 2000, True, 4096), (480000, 1200, 5000, 1, 2000, True, 4096)]).code in (0, -errno.ENODEV)
    assert binding.Batch([(240000, 1200, 5000, 1, 2000, True, 4096)]).code in (0, -errno.ENODEV)
    assert binding.Batch([(48000000, 1200, 5000, 1, 2000, True, 4096)]).code == -errno.ENOTSUP
    # ... and one whose tile + halo + taps exceed a CU's 160 KiB of LDS (2.4 MHz / 600 baud: 10909 + 2891 taps) is refused
    # when the batch is planned, not when its first kernel is launched; the same channel at 9600 baud fits
    assert binding.Batch([(2400000, 600, 5000, 1, 2000, True, 4096)]).code == -errno.ENOTSUP
    assert binding.Batch([(2400000, 9600, 5000, 1, 2000, True, 4096)]).code in (0, -errno.ENODEV)


@pytest.mark.skipif(binding.load().sdrm_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_enodev_not_a_cpu_fallback(capfd):
    b = binding.Batch([(48000, 9600, 5000, 1, 2000, True, 4096)])
    assert b.code == -errno.ENODEV
    assert "no CPU fallback" in capfd.readouterr().err
    out = np.zeros(4, np.float32)
    assert binding.load().sdrm_probe_atan2(out.ctypes.data, out.ctypes.data, out.ctypes.data, 4) == -errno.ENODEV


# ---------------------------------------------------------------- queue (reference test/test_queue.c)

def cbuf(vals):
    return np.array(vals, dtype=np.float32)


def test_queue_invalid_arguments(capfd):
    assert binding.Queue(4, 0, False).code == -1
    assert binding.Queue(0, 10, False).code == -1
    q = binding.Queue(4, 10, False)
    assert q.code == 0
    assert q.put(None) == -1
    assert q.put(np.zeros(0, np.float32)) == -1
    assert q.put_raw(cbuf([1, 2]).ctypes.data, 0) == -1
    assert q.put(cbuf(range(1, 11))) == -1  # 5 samples > buffer_size 4
    q.close()
    err = capfd.readouterr().err
    assert "<3>invalid queue size: 0" in err and "<3>invalid buffer size: 0" in err
    assert "<3>requested buffer 5 is more than max: 4" in err


def test_queue_put_take():
    q = binding.Queue(262144, 10, False)
    a, b = cbuf(range(1, 11)), cbuf([1, 2])
    assert q.put(a) == 0 and q.put(b) == 0
    assert np.array_equal(q.take(), a)
    q.complete()
    assert np.array_equal(q.take(), b)
    q.complete()
    q.close()


def test_queue_overflow_overwrites_newest(capfd):
    q = binding.Queue(262144, 1, False)
    a, b = cbuf(range(1, 11)), cbuf(range(11, 21))
    assert q.put(a) == 0 and q.put(b) == 0
    assert np.array_equal(q.take(), b)
    q.complete()
    q.close()
    assert "<3>queue is full" in capfd.readouterr().err


def test_queue_terminated_only_after_fully_processed():
    q = binding.Queue(262144, 10, False)
    a = cbuf(range(1, 11))
    q.put(a)
    q.interrupt()
    assert np.array_equal(q.take(), a)
    q.complete()
    assert q.take() is None
    binding.load().interrupt_waiting_the_data(None)  # no-op
    q.close()


def test_queue_put_skipped_after_termination():
    q = binding.Queue(262144, 1, True)
    a = cbuf(range(1, 11))
    assert q.put(a) == 0
    q.interrupt()
    assert q.put(a) == -1
    q.close()


def test_queue_blocking_put_waits_for_a_free_slot():
    q = binding.Queue(16, 1, True)
    a, b = cbuf([1, 2]), cbuf([3, 4])
    assert q.put(a) == 0
    done = []
    t = threading.Thread(target=lambda: done.append(q.put(b)))
    t.start()
    time.sleep(0.1)
    assert not done  # still blocked: the only slot is filled
    assert np.array_equal(q.take(), a)
    time.sleep(0.05)
    assert not done  # slot detached, not yet recycled
    q.complete()
    t.join(2)
    assert done == [0]
    assert np.array_equal(q.take(), b)
    q.complete()
    q.close()


# ---------------------------------------------------------------- wire framing

def _varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _field(num, value):
    """proto2 field: int -> varint, bytes -> length-delimited"""
    if isinstance(value, bytes):
        return _varint(num << 3 | 2) + _varint(len(value)) + value
    return _varint(num << 3) + _varint(value)


def test_wire_response_bytes_and_header_round_trip():
    """reference src/api.h:23-27 (packed header: version, type, u32 length in network order) and src/api_utils.c:82-108
    (Response{status, details}, both required and therefore always serialised): the bytes a reference client reads"""
    import socket
    L = binding.load()
    L.sdrm_wire_write_response.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
    L.sdrm_wire_read_header.argtypes = [C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)]
    a, b = socket.socketpair()
    assert L.sdrm_wire_write_response(a.fileno(), 0, 7) == 0        # SUCCESS, details = client id 7 (tcp_server.c:677)
    assert b.recv(64) == bytes([0, 2, 0, 0, 0, 4, 0x08, 0x00, 0x10, 0x07])
    assert L.sdrm_wire_write_response(a.fileno(), 1, 300) == 0      # FAILURE, a two-byte varint
    assert b.recv(64) == bytes([0, 2, 0, 0, 0, 5, 0x08, 0x01, 0x10, 0xac, 0x02])
    b.sendall(bytes([0, 0, 0, 0, 1, 44]) + bytes(300))               # an RxRequest header announcing 300 bytes
    t, n = C.c_uint8(), C.c_uint32()
    assert L.sdrm_wire_read_header(a.fileno(), C.byref(t), C.byref(n)) == 0 and (t.value, n.value) == (0, 300)
    a.recv(300)
    b.sendall(bytes([9, 0, 0, 0, 0, 0]))                             # wrong protocol version
    assert L.sdrm_wire_read_header(a.fileno(), C.byref(t), C.byref(n)) == -2
    b.sendall(bytes([0, 0, 0xff, 0xff, 0xff, 0xff]))                 # a peer announcing 4 GiB does not size our buffer
    assert L.sdrm_wire_read_header(a.fileno(), C.byref(t), C.byref(n)) == -3
    b.sendall(bytes([0, 0, 0, 1, 0, 0]))                             # 65536: the largest body accepted
    assert L.sdrm_wire_read_header(a.fileno(), C.byref(t), C.byref(n)) == 0 and n.value == 65536
    b.close()
    assert L.sdrm_wire_read_header(a.fileno(), C.byref(t), C.byref(n)) == -1  # peer gone
    a.close()


def test_wire_rx_request_fields_reach_the_worker_configuration():
    """api.proto:35-49: the fields src/dsp_worker.c:120-163 reads, decoded without protobuf-c; unknown fields are skipped,
    a missing required field or another modem type is an error"""
    L = binding.load()
    L.sdrm_wire_decode_rx_request.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(binding.WorkerConfig), C.POINTER(C.c_int)]
    fsk = _field(1, -5000) + _field(2, 2000) + _field(3, 1)            # deviation (int64, negative), transition width, dc
    doppler = _field(1, b"LUCKY-7") + _field(2, 1) + _field(3, 2) + _field(4, 3)
    body = (_field(1, 437525000) + _field(2, 48000) + _field(3, 1) + _field(4, -12000) + _field(5, 1) + _field(6, 4800) +
            _field(7, 2) + _field(8, 1) + _field(9, doppler) + _field(10, fsk) + _field(99, 12345) + _field(98, b"future"))
    cfg, dop = binding.WorkerConfig(), C.c_int(-1)
    buf = (C.c_uint8 * len(body)).from_buffer_copy(body)
    assert L.sdrm_wire_decode_rx_request(buf, len(body), C.byref(cfg), C.byref(dop)) == 0
    assert (cfg.rx_sampling_freq, cfg.demod_baud_rate, cfg.demod_decimation, cfg.demod_destination) == (48000, 4800, 2, 1)
    assert (cfg.demod_fsk_deviation, cfg.demod_fsk_transition_width, cfg.demod_fsk_use_dc_block, cfg.rx_dump_file) == (-5000, 2000, True, True)
    assert dop.value == 1
    short = _field(1, 1) + _field(2, 48000) + _field(3, 0) + _field(4, 0) + _field(5, 1) + _field(6, 4800) + _field(7, 2)  # no destination
    buf = (C.c_uint8 * len(short)).from_buffer_copy(short)
    assert L.sdrm_wire_decode_rx_request(buf, len(short), C.byref(cfg), C.byref(dop)) == -1
    bad = body[:len(body) - 3]                                        # truncated in the middle of a field
    buf = (C.c_uint8 * len(bad)).from_buffer_copy(bad)
    assert L.sdrm_wire_decode_rx_request(buf, len(bad), C.byref(cfg), C.byref(dop)) == -1
    # demod_destination outside FILE / SOCKET / BOTH (api.proto:29-33) is refused, not turned into a worker that
    # demodulates and discards
    odd = body.replace(_field(8, 1), _field(8, 3))
    buf = (C.c_uint8 * len(odd)).from_buffer_copy(odd)
    assert L.sdrm_wire_decode_rx_request(buf, len(odd), C.byref(cfg), C.byref(dop)) == -1
    # a key whose high bits would alias onto field 8 when truncated to 32 bits (field 2^29 + 8), and field number 0
    for key in ((((1 << 29) + 8) << 3), 0):
        v, enc = key, b""
        while True:
            enc += bytes([(v & 0x7f) | (0x80 if v >> 7 else 0)])
            v >>= 7
            if not v:
                break
        alias = short + enc + bytes([1])
        buf = (C.c_uint8 * len(alias)).from_buffer_copy(alias)
        assert L.sdrm_wire_decode_rx_request(buf, len(alias), C.byref(cfg), C.byref(dop)) == -1, key


def _ref_adapter(tmp_path):
    """integration/dsp_worker_ref.c (dsp_worker_create with the reference's own parameter list, src/dsp_worker.h:22)
    built against integration/ref_fields.h -- the two reference types restated, api.pb-c.h:104-121, server_config.h:16-40 --
    and linked against the library"""
    import subprocess
    so = os.path.join(str(tmp_path), "libref_adapter.so")
    libdir = os.path.dirname(binding.LIB_PATH)
    subprocess.check_call(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "integration"), os.path.join(ROOT, "integration", "dsp_worker_ref.c"),
                           "-o", so, "-L", libdir, "-lsdrmodem_hip", "-Wl,-rpath," + libdir])
    binding.load()  # the library first: the adapter's DT_NEEDED then resolves to the copy already mapped
    return C.CDLL(so)


class _PbBase(C.Structure):
    _fields_ = [("descriptor", C.c_void_p), ("n_unknown_fields", C.c_uint), ("unknown_fields", C.c_void_p)]


class _FskSettings(C.Structure):
    _fields_ = [("base", _PbBase), ("demod_fsk_deviation", C.c_int64), ("demod_fsk_transition_width", C.c_uint32),
                ("demod_fsk_use_dc_block", C.c_int)]


class _RxRequest(C.Structure):
    _fields_ = [("base", _PbBase), ("rx_center_freq", C.c_uint64), ("rx_sampling_freq", C.c_uint64), ("rx_dump_file", C.c_int),
                ("rx_offset", C.c_int64), ("demod_type", C.c_int), ("demod_baud_rate", C.c_uint32), ("demod_decimation", C.c_uint32),
                ("demod_destination", C.c_int), ("doppler", C.c_void_p), ("fsk_settings", C.POINTER(_FskSettings)),
                ("file_settings", C.c_void_p)]


class _ServerConfig(C.Structure):
    _fields_ = [("bind_address", C.c_char_p), ("port", C.c_uint16), ("read_timeout_seconds", C.c_int), ("buffer_size", C.c_uint32),
                ("queue_size", C.c_uint16), ("rx_sdr_type", C.c_uint8), ("rx_sdr_server_address", C.c_char_p),
                ("rx_sdr_server_port", C.c_int), ("base_path", C.c_char_p), ("rx_file_base_path", C.c_char_p),
                ("tx_file_base_path", C.c_char_p), ("tx_sdr_type", C.c_uint8), ("tx_plutosdr_gain", C.c_double),
                ("rx_plutosdr_gain", C.c_double), ("tx_plutosdr_timeout_millis", C.c_uint), ("iio", C.c_void_p)]


def _request(fs, baud, dev, decim, tw, dc, dump, dest):
    fsk = _FskSettings(demod_fsk_deviation=dev, demod_fsk_transition_width=tw, demod_fsk_use_dc_block=1 if dc else 0)
    req = _RxRequest(rx_center_freq=437525000, rx_sampling_freq=fs, rx_dump_file=1 if dump else 0, demod_type=1,
                     demod_baud_rate=baud, demod_decimation=decim, demod_destination=dest, fsk_settings=C.pointer(fsk))
    req._keep = fsk
    return req


def test_reference_signature_adapter_compiles_and_maps_the_request(tmp_path, capfd):
    """the adapter a maintainer drops into sdr-modem in place of src/dsp_worker.c: compiled -Wall -Wextra -Werror, called
    with the reference's parameter list.  Without a GPU the demodulator cannot be created, but the request is mapped and
    judged first, as in the reference: its own unit test's bad request (test/test_dsp_worker.c:58-66, baud == sampling
    rate => LPF cutoff above fs/2 => -1) fails with -1 and the reference's message, a good one with -ENODEV."""
    A = _ref_adapter(tmp_path)
    A.dsp_worker_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(_ServerConfig), C.POINTER(_RxRequest), C.POINTER(C.c_void_p)]
    sc = _ServerConfig(buffer_size=4096, queue_size=4, rx_sdr_type=2, base_path=str(tmp_path).encode())
    w = C.c_void_p()
    bad = _request(48000, 48000, 5000, 1, 2000, True, False, 0)
    assert A.dsp_worker_create(3, -1, C.byref(sc), C.byref(bad), C.byref(w)) == -1
    assert "<3>[3] unable to create demodulator" in capfd.readouterr().err
    if binding.load().sdrm_device_count() == 0:
        good = _request(48000, 4800, 5000, 2, 2000, True, False, 0)
        assert A.dsp_worker_create(4, -1, C.byref(sc), C.byref(good), C.byref(w)) == -errno.ENODEV
    other = _request(48000, 4800, 5000, 2, 2000, True, False, 0)
    other.demod_type = 7  # not GMSK: rejected instead of dereferencing a demodulator that was never made
    assert A.dsp_worker_create(5, -1, C.byref(sc), C.byref(other), C.byref(w)) == -1


def test_reference_signature_adapter_drives_workers_on_a_shared_batcher(tmp_path):
    """the same adapter with a per-GPU batcher attached (here: the product's batcher code over the kernel emulation): three
    RX clients created through dsp_worker_create(id, socket, server_config *, RxRequest *, &worker), fed through
    dsp_worker_put like sdr_worker.c:25-29 does, shut down through dsp_worker_destroy; the files the reference's worker
    writes (rx.demod2client.<id>.s8, rx.sdr2demod.<id>.cf32, src/dsp_worker.c:154,165) hold the oracle's bytes."""
    import emu_api
    import orc
    from sdr_modem_amd import siggen
    A = _ref_adapter(tmp_path)
    A.dsp_worker_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(_ServerConfig), C.POINTER(_RxRequest), C.POINTER(C.c_void_p)]
    NEXT = C.CFUNCTYPE(C.c_size_t, C.c_void_p)
    A.sdrm_ref_attach_batcher.argtypes = [C.c_void_p, NEXT, C.c_void_p]
    L = binding.load()
    cfg = (48000, 4800, 5000, 2, 2000, True, 4096)
    n_w = 3
    bt = emu_api.emu_batcher([cfg] * n_w, slots=4, max_wait_us=20000, blocking=True)
    counter = [0]

    def next_channel(_user):
        counter[0] += 1
        return counter[0] - 1
    cb = NEXT(next_channel)
    A.sdrm_ref_attach_batcher(bt.h, cb, None)
    sc = _ServerConfig(buffer_size=4096, queue_size=4, rx_sdr_type=2, base_path=str(tmp_path).encode())
    sigs = [siggen.gmsk_channel(70 + i, 3 * 4096 + 100, fs=48000, baud=4800) for i in range(n_w)]
    ws = []
    for i in range(n_w):
        req = _request(48000, 4800, 5000, 2, 2000, True, i == 1, 0)
        w = C.c_void_p()
        assert A.dsp_worker_create(30 + i, -1, C.byref(sc), C.byref(req), C.byref(w)) == 0
        ws.append(w)
    L.dsp_worker_put.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.dsp_worker_destroy.argtypes = [C.c_void_p]

    def feed(i):
        for off in range(0, len(sigs[i]), 4096):
            part = np.ascontiguousarray(sigs[i][off:off + 4096]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, len(part) // 2, ws[i])
    import threading
    th = [threading.Thread(target=feed, args=(i,)) for i in range(n_w)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
        assert not t.is_alive()
    for w in ws:
        L.dsp_worker_destroy(w)
    for i in range(n_w):
        got = np.fromfile(os.path.join(str(tmp_path), "rx.demod2client.%d.s8" % (30 + i)), dtype=np.int8)
        want, _ = orc.demod_stream(cfg[:6], sigs[i], 4096)
        assert np.array_equal(got, want), i
    dump = np.fromfile(os.path.join(str(tmp_path), "rx.sdr2demod.31.cf32"), dtype=np.complex64)
    assert np.array_equal(dump, sigs[1])
    A.sdrm_ref_attach_batcher(None, NEXT(0), None)
    bt.close()


def test_reference_signature_adapter_places_workers_through_a_node(tmp_path):
    """the adapter with a NODE attached (sdrm_ref_attach_node; here the product's node code over two virtual devices of the
    kernel emulation): four RX clients created with the reference's parameter list are placed two per device -- clients of one
    SDR source (same centre frequency, sdr_worker.c:83-95) together --, demodulate to the oracle's bytes, and give their slots
    back in dsp_worker_destroy."""
    import emu_api
    import orc
    from sdr_modem_amd import siggen
    A = _ref_adapter(tmp_path)
    A.dsp_worker_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(_ServerConfig), C.POINTER(_RxRequest), C.POINTER(C.c_void_p)]
    A.sdrm_ref_attach_node.argtypes = [C.c_void_p]
    L = binding.load()
    cfg = (48000, 4800, 5000, 2, 2000, True, 4096)
    node = emu_api.emu_node(cfg, 4, 2, batcher=(4, 20000, True))
    A.sdrm_ref_attach_node(node.h)
    sc = _ServerConfig(buffer_size=4096, queue_size=4, rx_sdr_type=2, base_path=str(tmp_path).encode())
    sigs = [siggen.gmsk_channel(90 + i, 2 * 4096 + 50, fs=48000, baud=4800) for i in range(4)]
    ws = []
    for i in range(4):
        req = _request(48000, 4800, 5000, 2, 2000, True, False, 0)
        req.rx_center_freq = 437525000 if i % 2 == 0 else 145800000  # two SDR sources
        w = C.c_void_p()
        assert A.dsp_worker_create(60 + i, -1, C.byref(sc), C.byref(req), C.byref(w)) == 0
        ws.append(w)
    assert [node.stat(d).clients for d in range(2)] == [2, 2]
    L.dsp_worker_put.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.dsp_worker_destroy.argtypes = [C.c_void_p]
    import threading

    def feed(i):
        for off in range(0, len(sigs[i]), 4096):
            part = np.ascontiguousarray(sigs[i][off:off + 4096]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, len(part) // 2, ws[i])
    th = [threading.Thread(target=feed, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
        assert not t.is_alive()
    for w in ws:
        L.dsp_worker_destroy(w)
    assert [node.stat(d).clients for d in range(2)] == [0, 0]
    for i in range(4):
        got = np.fromfile(os.path.join(str(tmp_path), "rx.demod2client.%d.s8" % (60 + i)), dtype=np.int8)
        want, _ = orc.demod_stream(cfg[:6], sigs[i], 4096)
        assert np.array_equal(got, want), i
    A.sdrm_ref_attach_node(None)
    node.close()


def test_binding_structures_have_the_headers_sizes_and_offsets(tmp_path):
    """The ctypes mirrors in sdr-modem_amd/binding.py (test and bench infrastructure) against include/sdrmodem_hip.h as a C
    compiler lays it out: size of every structure, offset of its last field.  A field added on one side only shows here."""
    import ctypes as C
    import subprocess
    pairs = [("sdrm_fsk_config", binding.FskConfig), ("sdrm_fsk_info", binding.FskInfo), ("sdrm_nco_segment", binding.NcoSegment),
             ("sdrm_worker_config", binding.WorkerConfig), ("sdrm_batcher_config", binding.BatcherConfig),
             ("sdrm_batch_schedule_info", binding.ScheduleInfo), ("sdrm_node_config", binding.NodeConfig),
             ("sdrm_node_slot", binding.NodeSlot), ("sdrm_node_stat", binding.NodeStat)]
    src = tmp_path / "sizes.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sdrmodem_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        last = cls._fields_[-1][0]
        lines.append('    printf("%s %%zu %%zu\\n", sizeof(%s), offsetof(%s, %s));' % (cname, cname, cname, last))
    lines += ['    return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict((ln.split()[0], (int(ln.split()[1]), int(ln.split()[2]))) for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in pairs:
        last = cls._fields_[-1][0]
        assert out[cname] == (C.sizeof(cls), getattr(cls, last).offset), (cname, out[cname], C.sizeof(cls), getattr(cls, last).offset)


def test_public_header_is_plain_c99_and_cxx11(tmp_path):
    """include/sdrmodem_hip.h is what a C host (sdr-modem is C99) and a C++ host include: no torch, no HIP types"""
    import subprocess
    hdr = os.path.join(ROOT, "include", "sdrmodem_hip.h")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", hdr])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++", hdr])
    includes = [ln for ln in open(hdr) if ln.lstrip().startswith("#include")]
    assert all(any(std in ln for std in ("<stdbool.h>", "<stddef.h>", "<stdint.h>", "<complex.h>")) for ln in includes), includes


# ---------------------------------------------------------------- the shipped Doppler factory (build container: reference tree present)

FACTORY_SO = os.path.join(ROOT, "oracle", "_ref", "libsdrm_doppler_factory.so")
LUCKY7_TLE = [b"LUCKY-7", b"1 44406U 19038W   20069.88080907  .00000505  00000-0  32890-4 0  9992",
              b"2 44406  97.5270  32.5584 0026284 107.4758 252.9348 15.12089395 37524"]


class _DopplerSettings(C.Structure):
    _fields_ = [("base", _PbBase), ("n_tle", C.c_size_t), ("tle", C.POINTER(C.c_char_p)), ("latitude", C.c_uint32),
                ("longitude", C.c_uint32), ("altitude", C.c_uint32)]


class _FileSettings(C.Structure):
    _fields_ = [("base", _PbBase), ("filename", C.c_char_p), ("start_time_seconds", C.c_uint64)]


@pytest.mark.skipif(not os.path.exists(FACTORY_SO), reason="oracle/_ref is built where the reference tree is (make -C oracle)")
def test_shipped_doppler_factory_returns_the_references_shifts():
    """integration/doppler_factory_ref.c: the factory the reference-signature adapter wants (sdrm_ref_set_doppler_factory), on
    the reference's own vendored SGP4 (src/sgpsdp, compiled where it lies into oracle/_ref).  With doppler_create's arguments
    of the reference's Doppler test (test/test_doppler.c:14,38: LUCKY-7, 53.72F N 47.57F E, 437.525 MHz, 1583840449) the
    per-second shifts are the fixture's doubles, bit for bit -- the fixture comes from oracle/ref_doppler_shifts.c, a second
    statement of src/dsp/doppler.c:31-42,151-172 --; asked out of order they are the same; and through the request
    (RxRequest.doppler as src/dsp_worker.c:120-136 reads it: degrees x 10E6 in uint32 fields) they are the shifts of that
    station."""
    import json
    import struct
    F = C.CDLL(FACTORY_SO)
    SHIFT = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_uint64)
    F.sdrm_ref_doppler_open.argtypes = [C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_uint64, C.c_int64, C.c_int64,
                                        C.c_char * 80 * 3, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    F.sdrm_ref_doppler_close.argtypes = [C.c_void_p]
    F.sdrm_ref_doppler_factory.argtypes = [C.POINTER(_RxRequest), C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "doppler_shifts_lucky7.json")))["shifts_hz"]
    tle = (C.c_char * 80 * 3)()
    for i, line in enumerate(LUCKY7_TLE):
        tle[i].value = line
    f32 = lambda v: struct.unpack("<f", struct.pack("<f", v))[0]  # noqa: E731  (the test passes float literals)
    fn, user = C.c_void_p(), C.c_void_p()
    assert F.sdrm_ref_doppler_open(f32(53.72), f32(47.57), 0.0, 48000, 437525000, 0, 1583840449, tle, C.byref(fn), C.byref(user)) == 0
    shift = SHIFT(fn.value)
    got = [shift(user, k) for k in range(len(want))]
    assert got == want
    assert shift(user, 5) == want[5] and shift(user, 2) == want[2] and shift(user, 3) == want[3]  # any order: the same sums
    F.sdrm_ref_doppler_close(user)
    bad = (C.c_char * 80 * 3)()
    assert F.sdrm_ref_doppler_open(53.72, 47.57, 0.0, 48000, 437525000, 0, 1583840449, bad, C.byref(fn), C.byref(user)) == -1
    # through the request
    lines = (C.c_char_p * 3)(*LUCKY7_TLE)
    ds = _DopplerSettings(n_tle=3, tle=lines, latitude=537200000, longitude=475700000, altitude=0)
    fs_ = _FileSettings(filename=b"x.cf32", start_time_seconds=1583840449)
    req = _request(48000, 4800, 5000, 2, 2000, True, False, 0)
    req.doppler = C.cast(C.pointer(ds), C.c_void_p)
    req.file_settings = C.cast(C.pointer(fs_), C.c_void_p)
    assert F.sdrm_ref_doppler_factory(C.byref(req), None, C.byref(fn), C.byref(user)) == 0
    via_request = [SHIFT(fn.value)(user, k) for k in range(len(want))]
    F.sdrm_ref_doppler_close(user)
    assert F.sdrm_ref_doppler_open(537200000 / 10E6, 475700000 / 10E6, 0 / 10E3, 48000, 437525000, 0, 1583840449, tle, C.byref(fn),
                                   C.byref(user)) == 0
    assert via_request == [SHIFT(fn.value)(user, k) for k in range(len(want))]
    F.sdrm_ref_doppler_close(user)
    assert max(abs(a - b) for a, b in zip(via_request, want)) < 0.01  # 53.72F is not 53.72: the same pass to a hundredth of a hertz
    req.doppler = None
    assert F.sdrm_ref_doppler_factory(C.byref(req), None, C.byref(fn), C.byref(user)) == -1


def test_a_batcher_client_whose_iq_dump_cannot_be_written_is_ended(tmp_path, capfd):
    """rx_dump_file with a full disk: the reference's DSP thread prints "<3>[id] unable to write sdr data" and leaves its loop
    (src/dsp_worker.c:56-64) -- the client gets no more soft bits.  On a shared batcher the dump is written by the source thread
    in dsp_worker_put; until round 5 that path printed and carried on.  Now: message, the client's channel closed, later
    buffers dropped, its neighbour on the same batcher untouched."""
    import emu_api
    import orc
    from sdr_modem_amd import siggen
    L = binding.load()
    cfg = (48000, 4800, 5000, 2, 2000, True, 4096)
    bt = emu_api.emu_batcher([cfg] * 2, slots=4, max_wait_us=20000, blocking=True)
    os.symlink("/dev/full", os.path.join(str(tmp_path), "rx.sdr2demod.71.cf32"))
    L.dsp_worker_put.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.dsp_worker_destroy.argtypes = [C.c_void_p]
    L.sdrm_dsp_worker_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(binding.WorkerConfig), C.POINTER(C.c_void_p)]
    ws = []
    for i in range(2):
        wc = binding.WorkerConfig(rx_sampling_freq=48000, demod_baud_rate=4800, demod_fsk_deviation=5000, demod_decimation=2,
                                  demod_fsk_transition_width=2000, demod_fsk_use_dc_block=True, rx_dump_file=(i == 1), demod_destination=0,
                                  buffer_size=4096, queue_size=4, rx_file_source=True, base_path=str(tmp_path).encode(),
                                  batcher=bt.h, batcher_channel=i)
        w = C.c_void_p()
        assert L.sdrm_dsp_worker_create(70 + i, -1, C.byref(wc), C.byref(w)) == 0
        ws.append(w)
    sigs = [siggen.gmsk_channel(95 + i, 3 * 4096, fs=48000, baud=4800) for i in range(2)]
    for off in range(0, 3 * 4096, 4096):
        for i in (1, 0):
            part = np.ascontiguousarray(sigs[i][off:off + 4096]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, len(part) // 2, ws[i])
    for w in ws:
        L.dsp_worker_destroy(w)
    err = capfd.readouterr().err
    assert err.count("<3>[71] unable to write sdr data") == 1
    got0 = np.fromfile(os.path.join(str(tmp_path), "rx.demod2client.70.s8"), dtype=np.int8)
    assert np.array_equal(got0, orc.demod_stream(cfg[:6], sigs[0], 4096)[0])
    assert os.path.getsize(os.path.join(str(tmp_path), "rx.demod2client.71.s8")) == 0  # ended before its first buffer was demodulated
    bt.close()


def test_device_wide_ledger_of_waiting_hand_off_workgroups(tmp_path):
    """sdr-modem_amd/host/ledger.cpp (what decides, per device and across every batch and handle of the process, whether a call
    may take the in-call hand-off): the budget of waiting workgroups with each caller's own limit, entries that are reaped once
    their call's event has fired -- but never before the event was recorded --, an owner's new call replacing its old entry, and
    the count of plain handles' calls in flight.  Deterministic scenario (tests/ledger_check.cpp); the threaded stress runs under
    the sanitizers (tests/san/host_stress.cpp, profiles/r06_sanitizers.txt)."""
    exe = os.path.join(str(tmp_path), "ledger_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", os.path.join(ROOT, "tests", "ledger_check.cpp"),
                           os.path.join(ROOT, "sdr-modem_amd", "host", "ledger.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "ledger ok" in out.stdout, out.stderr
