"""CPU-only: the C-ABI library builds for gfx950, loads without a GPU, exports every symbol the public header
declares, reports parameter errors like the reference and refuses to compute without a device (no CPU fallback).
Also the host-only queue, mirrored from the reference's test/test_queue.c."""
import ctypes as C
import errno
import os
import re
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
    assert binding.Batch([(48000, 1200, 5000, 1, 2000, True, 4096), (480000, 1200, 5000, 1, 2000, True, 4096)]).code == -errno.ENOTSUP


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
