"""`-m gpu`: the shapes bench.py REPORTS, checked against the oracle at the sizes it reports them at.

The other parity tests feed 8 192- or 16 384-sample calls; the library picks its schedule from the call length and the
channel count (clock-stage workgroup shape, companion grid, placement holds, int8 conversion in the staging wave or in
k3_quantize), so the 131 072-sample calls of the bench line are a different schedule.  These tests drive exactly the
bench path -- bench.Rig: device-resident input, sdrm_batch_process_device with rolling chunk offsets, no host
synchronisation between steps -- and compare spot channels with the CPU restatement of the reference, int8 and float
soft bits bit for bit, after the last step and at two points in between (reference behaviour:
src/dsp/fsk_demod.c:80-110 on buffer_size 131072, src/resources/config.conf:11)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import sdrm_pkg  # noqa: E402

sdrm_pkg.load()
from sdr_modem_amd import binding, siggen  # noqa: E402

import bench  # noqa: E402  (repo root: the Rig / SpotChecker the bench line itself uses)

pytestmark = pytest.mark.gpu
N = 131072
CFG = (bench.FS, bench.BAUD, bench.DEV, bench.DECIM, bench.TW, bench.DC, N)


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    assert binding.load().sdrm_device_count() > 0, "these tests need an MI355X; the library has no CPU path"
    torch.cuda.set_device(0)
    return torch, torch.device("cuda", 0)


def _run(torch, dev, channels, steps, check_at, spots, resident=2):
    rig = bench.Rig(torch, binding, siggen, dev, 0, [CFG] * channels, 0, N, resident)
    checker = bench.SpotChecker(rig.cfgs, rig.row, N, spots)
    checks = 0
    try:
        for i in range(steps):
            rig.step(i)  # no synchronisation between steps: stages of consecutive calls overlap as in the bench
            if (i + 1) in check_at or i + 1 == steps:
                ok, detail = checker.check(rig.batch, rig.fed)
                assert ok, ("%d channels x %d samples, after step %d" % (channels, N, i + 1), detail)
                checks += 1
        # size-independent property at this size: every channel produced ~N * baud / fs symbols in the last call
        _, olen = rig.batch.fetch(1)
        assert np.all(np.abs(np.asarray(olen) - N * bench.BAUD / bench.FS) < 40)
    finally:
        rig.close()
    return checks


def test_headline_shape_256_channels_pipelined_twelve_steps(torch_dev):
    """BASELINE configs[2] as benchmarked: 256 x 131072, 12 pipelined steps -- the companion grid live, the clock stage
    in its 16-channel x 1024-sample shape with the int8 conversion in the staging wave.  8 spot channels incl. both sides
    of a clock-stage workgroup boundary and the last channel; checked after steps 4, 8 and 12."""
    torch, dev = torch_dev
    assert _run(torch, dev, 256, 12, (4, 8), [0, 15, 16, 17, 100, 129, 240, 255], resident=4) == 3


@pytest.mark.parametrize("channels", [1024, 4096])
def test_sweep_shapes_at_full_call_length(torch_dev, channels):
    """bench.py's channel_sweep points: 1024 channels (front-end placement hold; 16 x 1024 clock stage) and 4096 (64
    channels x 256-sample plain ring + k3_quantize), 131072-sample calls, 3 pipelined steps; 8 spot channels each incl.
    the last workgroup of every stage."""
    torch, dev = torch_dev
    spots = [0, 15, 16, 63, 64, channels // 2 + 3, channels - 2, channels - 1]
    assert _run(torch, dev, channels, 3, (1, 2), spots) == 3


def test_config5_workload_as_benchmarked(torch_dev):
    """bench.py's `config5` block: 256 channels, 240 kHz / 19200 baud / decimation 5 interleaved with 48 kHz / 1200 baud /
    decimation 8, 131072-sample calls, three NCO batches per channel and call; 192 warm-up steps (the batch refines its schedule
    online over the first calls with NCO batches: front hold and companion grid change between calls) + 3 timed steps; 6 spot channels
    against orc.Nco + orc.Fsk after every timed step."""
    torch, dev = torch_dev
    res = bench.config5_single(torch, binding, siggen, dev, 256, N, steps=3, verify=True, check_at=(2,))
    assert res["verified_vs_oracle"] is True, res["verify_mismatches"]
    assert res["schedule"]["online"]["state"] == 2 and res["schedule"]["online"]["choice"] in (0, 1, 2, 3), res["schedule"]
    assert all(ms > 0 for ms in res["schedule"]["online"]["ms_per_call"][:4]), res["schedule"]


def test_config5_at_the_reference_default_decimation_as_benchmarked(torch_dev):
    """bench.py's `config5_d1` block (SURVEY 8d Config 5, "run both"): the same mix at decimation 1 -- 12.5 and 40 samples per
    symbol, LPF2 of 289 taps at the full rate, DC boxcars of 400 and 1280 samples, the clock stage's tail quirk active
    (clock_recovery_mm.c:127-133: chunking must equal the oracle's, and does: one oracle call per device call) -- on per-kind
    seeded waveforms with circular shifts; spot channels of both kinds against orc.Nco + orc.Fsk."""
    torch, dev = torch_dev
    res = bench.config5_single(torch, binding, siggen, dev, 256, N, steps=3, verify=True, check_at=(2,), decimated=False)
    assert res["verified_vs_oracle"] is True, res["verify_mismatches"]
    assert res["wild_channel_calls"] == 0


def test_blocking_call_block_as_benchmarked(torch_dev):
    """bench.py's `blocking_call` block: one call at a time (src/dsp_worker.c:75) on 256 and 1024 channels and on one plain handle,
    with the in-call hand-off and with SDRM_HANDOFF=0, every last call against the oracle; the hand-off is taken by every call
    of the batches when it is on and by none when it is off."""
    torch, dev = torch_dev
    res = bench.blocking_calls(torch, binding, siggen, dev, N, verify=True, calls=3)
    for key in ("256x131072", "1024x131072"):
        r = res[key]
        assert r["handoff_verified"] is True and r["handoff_off_verified"] is True, r
        assert r["handoff_calls_taken"] >= 6 and r["handoff_off_calls_taken"] == 0, r
    for key in ("one_handle_48000_9600", "one_handle_48000_4800_d2"):
        assert res[key]["handoff_verified"] is True and res[key]["handoff_off_verified"] is True, res[key]


def test_bench_line_carries_the_spot_check(torch_dev, capsys):
    """`python bench.py` (short) prints verified_vs_oracle: true on the headline, the sweep and config5"""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--sweep", "1024",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["verified_vs_oracle"] is True, line.get("verify")
    assert line["channel_sweep"]["1024"]["verified_vs_oracle"] is True
    assert line["config5"]["verified_vs_oracle"] is True, line["config5"]
    assert line["config5_d1"]["verified_vs_oracle"] is True, line["config5_d1"]
    assert line["blocking_call"]["256x131072"]["handoff_verified"] is True, line["blocking_call"]
    assert line["blocking_call"]["one_handle_48000_9600"]["handoff_verified"] is True, line["blocking_call"]


def test_self_calibration_leaves_every_stream_as_new(torch_dev, monkeypatch):
    """A batch of 32 channels or more may time its own pipeline when it is created (sdrm_batch_schedule) and then puts every
    stream back to its initial state: the first real calls give the oracle's bits, with the calibration and without it
    (SDRM_AUTOTUNE=0), and the schedule says which of the two happened.  Small batches and forced settings are left alone."""
    import orc
    cfg = (48000, 9600, 5000, 1, 2000, True, 32768)
    C_ = 96
    sig = siggen.gmsk_batch(8, 2 * 32768, first_channel=40)
    outs = {}
    for mode in ("on", "off"):
        # (since round 6 a dimension is timed only where its rule is in doubt, which for 96 channels is nowhere: SDRM_AUTOTUNE=2
        # asks every question regardless)
        monkeypatch.setenv("SDRM_AUTOTUNE", "2" if mode == "on" else "0")
        g = binding.Batch([cfg] * C_, keep_soft=True)
        assert g.code == 0
        sch = g.schedule()
        assert sch["calibrated"] is (mode == "on"), sch
        if mode == "on":
            assert sch["ms_per_call_after"] <= sch["ms_per_call_before"] * 1.0001 and sch["calibration_ms"] > 0
        res = []
        for k in range(2):
            res.append(g.process([sig[c % 8, k * 32768:(k + 1) * 32768] for c in range(C_)]))
        for c in (0, 7, 95):
            o8, of = orc.demod_stream(cfg[:6], sig[c % 8], 32768)
            assert np.array_equal(np.concatenate([res[0][c], res[1][c]]), o8), (mode, c)
        assert np.array_equal(g.last_soft(95).view(np.uint32), orc.demod_stream(cfg[:6], sig[95 % 8], 32768)[1][-len(g.last_soft(95)):].view(np.uint32))
        outs[mode] = res
        g.close()
    for k in range(2):
        for c in range(C_):
            assert np.array_equal(outs["on"][k][c], outs["off"][k][c])
    monkeypatch.delenv("SDRM_AUTOTUNE")
    small = binding.Batch([cfg] * 8)
    assert small.schedule()["calibrated"] is False
    small.close()
    # left to itself: 96 channels are far from every rule's boundary (shape changes at 1280 / 2560 channels, the front hold is in
    # doubt from 640 to 1280, the companion grid is clearly worth it here) -- nothing is timed; 700 channels have open questions
    decided = binding.Batch([cfg] * C_)
    assert decided.schedule()["calibrated"] is False and decided.schedule()["calibration_ms"] == 0
    decided.close()
    asked = binding.Batch([cfg] * 700)
    assert asked.schedule()["calibrated"] is True and asked.schedule()["calibration_ms"] > 0
    asked.close()
    monkeypatch.setenv("SDRM_K3_LANES", "32")
    forced = binding.Batch([cfg] * 600)
    assert forced.schedule()["clock_stage"] == "32x512"
    forced.close()


def test_short_calls_on_a_long_buffer_batch_refine_the_schedule_online(torch_dev):
    """The creation-time calibration times full-length calls without NCO batches.  Calls of less than half that length are a
    class it did not cover: the batch re-decides front hold and companion grid over ~130 such calls from the 17th on (the starting
    point's steady state, four settings for eight calls each, the starting point and the winner again, the winner's probation;
    device-side timing) -- and every call's soft bits are the oracle's whatever setting it ran under; a
    full-length call afterwards runs the calibrated setting again."""
    import orc
    cfg = (48000, 9600, 5000, 1, 2000, True, 65536)
    C_ = 64
    n = 8192
    sig = siggen.gmsk_batch(4, 192 * n + 65536, first_channel=70)
    g = binding.Batch([cfg] * C_, keep_soft=True)
    assert g.code == 0 and g.schedule()["online"]["state"] == 0
    oracles = {c: orc.Fsk(*cfg) for c in (0, 17, 63)}
    for k in range(192):
        out = g.process([sig[c % 4, k * n:(k + 1) * n] for c in range(C_)])
        for c, o in oracles.items():
            assert np.array_equal(out[c], o.process(sig[c % 4, k * n:(k + 1) * n])[0]), (k, c)
    sch = g.schedule()["online"]
    assert sch["state"] == 2 and sch["choice"] in (0, 1, 2, 3) and all(ms > 0 for ms in sch["ms_per_call"][:4]), sch
    out = g.process([sig[c % 4, 192 * n:192 * n + 65536] for c in range(C_)])
    for c, o in oracles.items():
        assert np.array_equal(out[c], o.process(sig[c % 4, 192 * n:192 * n + 65536])[0]), c
    g.close()
