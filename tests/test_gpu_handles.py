"""`-m gpu`: the reference's process model at the reference's scale -- one demodulator and one blocking call per client and buffer
(src/dsp_worker.c:188 thread per client, :75 the call, src/tcp_server.c:659), buffers of 131072 samples
(src/resources/config.conf:11) -- through tools/handles_bench, a plain C program on the public header: many PRIVATE fsk_demod
handles (and private dsp_workers) calling at once, each long enough to take the in-call hand-off.  What must hold: no handle ends
in the sticky error state, every client's soft-bit stream is the oracle's for its input, and the device-wide ledger of waiting
workgroups and of plain handles' calls in flight (sdrm_handoff_stats) admits and refuses hand-offs as designed -- admission per
batch (round 5) let the waiting workgroups of different handles add up CU by CU."""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import orc
import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd import binding, siggen  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = (48000, 9600, 5000, 1, 2000, True)
BUF = 131072


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert binding.load().sdrm_device_count() > 0, "these tests need an MI355X; the library has no CPU path"


def _exe():
    exe = os.path.join(ROOT, "tools", "handles_bench")
    src = os.path.join(ROOT, "tools", "handles_bench.c")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-pthread", src, "-I" + os.path.join(ROOT, "include"),
                               "-L" + os.path.join(ROOT, "sdr-modem_amd", "csrc"), "-lsdrmodem_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "sdr-modem_amd", "csrc"), "-lm", "-o", exe])
    return exe


def fnv1a(data):
    h = 0xcbf29ce484222325
    for b in bytes(data):
        h = ((h ^ b) * 0x100000001b3) & 0xffffffffffffffff
    return h


@pytest.fixture(scope="module")
def streams():
    """eight client recordings of 6 buffers each, and what the oracle makes of each (symbol count, FNV-1a of the int8 stream)"""
    tmp = tempfile.TemporaryDirectory()
    files, want = [], []
    for k in range(8):
        iq = siggen.gmsk_channel(900 + k, 6 * BUF, carrier_offset_hz=37.0 * k - 120.0)
        path = os.path.join(tmp.name, "client%d.cf32" % k)
        iq.tofile(path)
        files.append(path)
        prefix = {}
        for calls in (3, 6):
            soft, _ = orc.demod_stream(CFG, iq[:calls * BUF], BUF)
            prefix[calls] = (len(soft), fnv1a(soft.tobytes()))
        want.append(prefix)
    yield files, want
    tmp.cleanup()


def _run(args, files, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([_exe()] + [str(a) for a in args] + files, env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    classes = {int(m.group(1)): (int(m.group(2)), int(m.group(3)), int(m.group(4), 16), m.group(5))
               for m in re.finditer(r"class (\d+): handles (\d+) symbols (\d+) fnv ([0-9a-f]+) agree (\w+)", out.stdout)}
    tail = re.search(r"errors (\d+), hand-off taken (\d+) refused (\d+) peak waiting (\d+)", out.stdout)
    assert tail, out.stdout
    ms = float(re.search(r"\): ([0-9.]+) ms,", out.stdout).group(1))
    return classes, tuple(int(x) for x in tail.groups()), ms, out.stderr


def _check(classes, want, calls, handles):
    assert len(classes) == min(len(want), handles)
    for k, (members, symbols, digest, agree) in classes.items():
        assert agree == "yes", "handles fed the same recording disagree (class %d)" % k
        assert (symbols, digest) == want[k][calls], "class %d: %d symbols, the oracle has %d (or other bits)" % (k, symbols, want[k][calls][0])


@pytest.mark.parametrize("handles,calls", [(3, 6), (8, 6), (64, 6), (256, 3)])
def test_many_private_handles_call_at_once_with_the_references_buffer_size(streams, handles, calls):
    files, want = streams
    classes, (errors, taken, refused, peak), ms, err = _run(["-W", 0, handles, BUF, calls], files)
    assert errors == 0 and "<3>" not in err, err[-1500:]
    _check(classes, want, calls, handles)
    # every call is long enough for the hand-off (>= 12288 samples) and meets an idle batch: it asked the device's ledger, which
    # admits a plain handle's call only while no other is in flight -- with threads calling at once nearly every call is refused
    # and runs its stages in stream order (the hand-off loses with concurrent handles, profiles/r06_handles.txt); whatever was
    # admitted stayed within the device's budget of waiting workgroups
    # (3 and 8 handles: admitted and refused calls alternate on the same handle -- its side streams and its own stream in turn)
    assert taken + refused == handles * calls and peak <= 192
    if handles >= 64:
        assert refused >= handles * (calls - 1)


def test_a_lone_private_handle_takes_the_hand_off_and_two_calling_together_do_not(streams):
    """one client: every call is admitted (two workgroups waiting).  Two clients calling at once: a plain handle's call takes the
    hand-off only while no other plain handle's call is in flight, so calls that overlap are refused -- whatever the mix, the
    streams are the oracle's"""
    files, want = streams
    classes, (errors, taken, refused, peak), ms, err = _run(["-W", 0, 1, BUF, 6], files)
    assert errors == 0 and "<3>" not in err, err[-1500:]
    _check(classes, want, 6, 1)
    assert taken == 6 and refused == 0 and peak == 2
    classes, (errors, taken, refused, peak), ms, err = _run(["-W", 0, 2, BUF, 6], files)
    assert errors == 0 and "<3>" not in err, err[-1500:]
    _check(classes, want, 6, 2)
    assert taken + refused == 12 and peak <= 2


def test_the_same_streams_without_the_hand_off(streams):
    files, want = streams
    classes, (errors, taken, refused, _), _, err = _run(["-W", 0, 64, BUF, 3], files, env={"SDRM_HANDOFF": "0"})
    assert errors == 0 and taken == 0 and refused == 0
    _check(classes, want, 3, 64)


def test_thirty_two_private_workers_with_the_references_buffer_size(streams):
    """dsp_worker_create x 32 (private handle each, file sink), buffer_size 131072, fed from 32 source threads through
    dsp_worker_put: the files the workers wrote are the oracle's streams"""
    files, want = streams
    classes, (errors, taken, refused, peak), _, err = _run(["-w", "-W", 0, 32, BUF, 6], files)
    assert errors == 0 and "<3>" not in err, err[-1500:]
    _check(classes, want, 6, 32)
    assert taken + refused == 32 * 6 and peak <= 192


def test_the_same_clients_sharing_one_batcher(streams):
    """sdrm_fsk_demod_share(64, 1000) -- the programmatic form of SDRM_SHARED_SLOTS -- before the handles are created: the 64 client
    threads' blocking calls become one device call per round of buffers; the streams stay the oracle's.  (What a server with more
    than a handful of clients wants: 1.7 instead of 0.09 Gsamples/s, profiles/r06_handles.txt.)"""
    files, want = streams
    classes, (errors, taken, refused, peak), ms, err = _run(["-s", "-W", 0, 64, BUF, 6], files)
    assert errors == 0 and "<3>" not in err, err[-1500:]
    _check(classes, want, 6, 64)


def test_three_batches_call_at_once_and_share_the_devices_budget():
    """Three threads, each with its own 96-channel batch, blocking calls of 32768 samples at once: every call meets an idle batch and
    asks for the hand-off with 12 waiting workgroups (6 clock-stage + 6 DC); the ledger adds them up per device (36 <= 192: all
    admitted while they overlap) and every channel's stream stays the oracle's."""
    import threading
    n, calls, C_ = 32768, 5, 96
    cfg = CFG + (n,)
    sig = siggen.gmsk_batch(8, calls * n, first_channel=950)
    want = [orc.demod_stream(CFG, sig[c], n)[0] for c in range(8)]
    before = binding.handoff_stats()
    bad = []

    def run(t):
        g = binding.Batch([cfg] * C_)
        if g.code != 0:
            bad.append((t, "create", g.code))
            return
        got = [[] for _ in range(C_)]
        for k in range(calls):
            out = g.process([sig[(c + t) % 8, k * n:(k + 1) * n] for c in range(C_)])
            for c in range(C_):
                got[c].append(out[c])
        for c in range(C_):
            if not np.array_equal(np.concatenate(got[c]), want[(c + t) % 8]):
                bad.append((t, c))
        if g.handoff_calls() == 0:
            bad.append((t, "no hand-off"))
        g.close()

    th = [threading.Thread(target=run, args=(t,)) for t in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
        assert not t.is_alive()
    assert not bad, bad[:5]
    after = binding.handoff_stats()
    assert after[0] > before[0] and after[2] <= 192
