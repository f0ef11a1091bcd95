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
