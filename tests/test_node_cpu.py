"""The in-process node front door (sdr-modem_amd/host/node.cpp, include/sdrmodem_hip.h sdrm_node_*) without a GPU: the same
C++ placement code over 2 / 4 / 8 VIRTUAL devices, each an emulation-backed batcher (tests/emu), driven the way the
reference's single process drives its clients (src/tcp_server.c:659 creates a worker per RX client, src/sdr_worker.c:25-55
feeds them, src/dsp_worker.c:188 runs the client's thread)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import emu_api
import orc
import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd import binding, shard, siggen  # noqa: E402

GEOM = (48000, 9600, 5000, 1, 2000, True, 4096)
HEAVY = (240000, 19200, 5000, 5, 2000, True, 4096)
LIGHT = (48000, 1200, 5000, 8, 2000, True, 4096)


def test_cost_function_is_the_one_the_rank_sharding_uses():
    """one placement cost for both front doors: shard.channel_cost (cuts a known table across ranks) and the C node
    (places clients as they arrive) -- fs x (4 T1 + 2 T2 / d) with the reference's tap-count rule"""
    for cfg in (GEOM, HEAVY, LIGHT, (48000, 4800, 5000, 2, 2000, False, 4096), (192000, 40000, 5000, 1, 2000, True, 4096),
                (240000, 9600, 5000, 1, 2000, True, 4096), (240000, 1200, 5000, 1, 2000, True, 4096)):
        assert binding.channel_cost(cfg) == shard.channel_cost(cfg), cfg


@pytest.mark.parametrize("devices", [2, 4, 8])
def test_placement_balances_a_mixed_fleet(devices):
    """BASELINE configs[4]'s mix arriving client by client in random order: every device ends within 1.15 of the lightest
    (what shard_by_cost achieves on the known table; online greedy placement leaves the devices at most ONE client's cost
    apart, here a 240 kHz channel = 7 % of a device's share), no device over its slots, counts differ where costs do"""
    slots = 96
    node = emu_api.emu_node(GEOM, slots, devices)
    assert node.batchers() == devices
    rng = np.random.default_rng(7)
    fleet = [HEAVY] * (8 * devices) + [LIGHT] * (56 * devices)
    order = rng.permutation(len(fleet))
    placed = []
    for i in order:
        code, slot = node.attach(fleet[i])
        assert code == 0
        placed.append((fleet[i], slot))
    stats = [node.stat(i) for i in range(devices)]
    loads = [st.load for st in stats]
    assert max(loads) / min(loads) <= 1.15, loads
    assert max(loads) - min(loads) <= binding.channel_cost(HEAVY) * (1 + 1e-9), loads
    assert sum(st.clients for st in stats) == len(fleet) and all(st.clients <= slots for st in stats)
    # every (batcher, channel) pair handed out once
    assert len({(s.batcher_index, s.channel) for _, s in placed}) == len(fleet)
    # slots are recycled across devices: the heavy clients of device 0 leave, the next clients land where the room is
    gone = [(c, s) for c, s in placed if s.batcher_index == 0 and c == HEAVY]
    for _, s in gone:
        assert node.detach(s) == 0
    assert node.detach(gone[0][1]) == -1  # a slot is given back once
    back = []
    for _ in range(len(gone)):
        code, slot = node.attach(HEAVY)
        assert code == 0
        back.append(slot)
    assert all(s.batcher_index == 0 for s in back), [s.batcher_index for s in back]
    assert {s.channel for s in back} == {s.channel for _, s in gone}
    node.close()


def test_clients_of_one_source_stay_together_while_that_balances():
    node = emu_api.emu_node(GEOM, 32, 4)
    slots = []
    # four sources with four clients each: each source ends on one device (a source's channels together)
    for k in range(4):
        for src in (11, 22, 33, 44):
            code, s = node.attach(GEOM, source_id=src)
            assert code == 0
            slots.append((src, s))
    by_src = {}
    for src, s in slots:
        by_src.setdefault(src, set()).add(s.batcher_index)
    assert all(len(v) == 1 for v in by_src.values()), by_src
    assert len({next(iter(v)) for v in by_src.values()}) == 4
    # ... but not at any price: a source that grows far beyond the others spills onto the least-loaded device
    for _ in range(12):
        code, s = node.attach(GEOM, source_id=11)
        assert code == 0
        slots.append((11, s))
    loads = [node.stat(i).load for i in range(4)]
    assert max(loads) / min(loads) <= 1.15 + 1e-9, loads
    node.close()


def test_a_full_node_says_so_and_takes_clients_again_after_a_detach():
    node = emu_api.emu_node(GEOM, 3, 2)
    held = []
    for _ in range(6):
        code, s = node.attach(GEOM)
        assert code == 0
        held.append(s)
    code, _ = node.attach(GEOM)
    assert code == -16  # -EBUSY
    assert node.detach(held[4]) == 0
    code, s = node.attach(GEOM)
    assert code == 0 and (s.batcher_index, s.channel) == (held[4].batcher_index, held[4].channel)
    node.close()


def test_clients_of_the_generic_stages_are_kept_together_and_away_from_the_others():
    """A client beyond the fast stages' range (48 kHz / 150 baud: 320 samples per symbol) runs the generic DC and clock stages,
    milliseconds per call that every client of the same batch waits for: such clients share a batcher, the others avoid it
    while another batcher has room (and use it when there is none)"""
    slow = (48000, 150, 5000, 1, 2000, True, 4096)
    node = emu_api.emu_node(GEOM, 8, 3)
    where = []
    for cfg in (GEOM, slow, GEOM, GEOM, slow, GEOM, slow, GEOM):
        code, s = node.attach(cfg)
        assert code == 0
        where.append((cfg, s))
    slow_on = {s.batcher_index for cfg, s in where if cfg == slow}
    fast_on = {s.batcher_index for cfg, s in where if cfg == GEOM}
    assert len(slow_on) == 1, where
    # the first slow client went to the least-loaded batcher (one that had a fast client by then: three batchers, one client
    # before it); fast clients that came later stay away from it
    later_fast = [s.batcher_index for cfg, s in where[2:] if cfg == GEOM]
    assert not (set(later_fast) & slow_on), where
    assert len(fast_on - slow_on) == 2
    # no room elsewhere: the fast clients take the slow batcher's free slots
    extra = []
    while True:
        code, s = node.attach(GEOM)
        if code != 0:
            assert code == -16
            break
        extra.append(s)
    assert sum(node.stat(i).clients for i in range(3)) == 24
    assert any(s.batcher_index in slow_on for s in extra)
    # the slow clients leave: the batcher is an ordinary one again
    for cfg, s in where:
        if cfg == slow:
            assert node.detach(s) == 0
    code, s = node.attach(slow)
    assert code == 0 and s.batcher_index in slow_on  # (the only batcher with free slots)
    node.close()


def _worker_cfg(node, tmp, cfg, source_id=0, offset=0):
    return binding.WorkerConfig(cfg[0], cfg[1], cfg[2], cfg[3], cfg[4], cfg[5], False, 0, cfg[6], 8, True, str(tmp).encode(),
                                None, None, None, 0, node.h, source_id, offset)


def test_workers_placed_by_the_node_demodulate_and_one_failed_device_ends_only_its_own_clients(tmp_path, capfd):
    """12 RX clients created with sdrm_worker_config.node on a node of 4 virtual devices: the workers attach themselves,
    3 per device, every client's file holds the oracle's bytes.  Then device 2 fails a call: its three clients end with the
    device's code, the other nine keep streaming, new clients are placed on the three healthy devices only, and the failed
    device's slots are not handed out again."""
    L = binding.load()
    node = emu_api.emu_node(GEOM, 6, 4, batcher=(4, 20000, True))
    n_w, chunk = 12, 4096
    cfgs = [GEOM if i % 3 else (48000, 4800, 5000, 2, 2000, True, 4096) for i in range(n_w)]
    sigs = [siggen.gmsk_channel(300 + i, 4 * chunk, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    ws = []
    for i in range(n_w):
        w = C.c_void_p()
        wc = _worker_cfg(node, tmp_path, cfgs[i])
        assert L.dsp_worker_create(100 + i, -1, C.byref(wc), C.byref(w)) == 0
        ws.append(w)
    assert [node.stat(d).clients for d in range(4)] == [3, 3, 3, 3]

    def feed(i, lo, hi):
        for k in range(lo, hi):
            part = np.ascontiguousarray(sigs[i][k * chunk:(k + 1) * chunk]).view(np.float32)
            L.dsp_worker_put(part.ctypes.data, chunk, ws[i])

    def feed_all(lo, hi):
        th = [threading.Thread(target=feed, args=(i, lo, hi)) for i in range(n_w)]
        for t in th:
            t.start()
        for t in th:
            t.join(60)
            assert not t.is_alive()
    feed_all(0, 2)
    # device 2 fails its next call
    emu_api.lib().emu_node_fail_device(2, 1)
    feed_all(2, 4)
    import time
    t0 = time.time()
    while node.stat(2).error == 0 and time.time() - t0 < 20:
        time.sleep(0.01)
    assert node.stat(2).error == -5 and [node.stat(d).error for d in (0, 1, 3)] == [0, 0, 0]
    # new clients go to the healthy devices only
    extra = []
    for k in range(6):
        w = C.c_void_p()
        wc = _worker_cfg(node, tmp_path, GEOM)
        assert L.dsp_worker_create(200 + k, -1, C.byref(wc), C.byref(w)) == 0
        extra.append(w)
    assert node.stat(2).attached == 3 and sum(node.stat(d).clients for d in (0, 1, 3)) == 9 + 6
    for w in ws + extra:
        L.dsp_worker_destroy(w)
    assert [node.stat(d).clients for d in range(4)] == [0, 0, 0, 0]
    err = capfd.readouterr().err
    assert err.count("the demodulator's device call failed (-5); the client is ended") == 3
    survivors, victims = 0, 0
    for i in range(n_w):
        got = np.fromfile(os.path.join(str(tmp_path), "rx.demod2client.%d.s8" % (100 + i)), dtype=np.int8)
        want, _ = orc.demod_stream(cfgs[i][:6], sigs[i], chunk)
        assert np.array_equal(got, want[:len(got)]), i  # whatever a client got, before the failure or all of it, is right
        if len(got) == len(want):
            survivors += 1
        else:
            victims += 1
    assert survivors == 9 and victims == 3
    node.close()


def test_worker_with_the_file_sources_rx_offset_matches_the_reference_vector(tmp_path):
    """RxRequest.rx_offset as the reference's file source applies it (src/sdr/file_source.c:120-128: sig_source_multiply with
    the offset on every buffer read): the reference's own vector (test/test_file_source.c:47-61, offset 1000 Hz at 48 kHz) and
    a long stream across buffers and one-second boundaries, through a worker on a node slot, against orc.Nco + orc.Fsk."""
    import json
    vec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_unit_vectors.json")))["vectors"]
    v = vec["test_file_source.c:rx_offset"]
    osc = orc.Nco(1.0, 48000, 2000)
    got = osc.multiply(v["offset_hz"], np.array(v["input"], dtype=np.float32))
    assert np.allclose(got, np.array(v["values"], dtype=np.float32), atol=v["tolerance"])  # the oracle's oscillator = the reference's
    L = binding.load()
    node = emu_api.emu_node(GEOM, 2, 1, batcher=(4, 20000, True))
    chunk = 4096
    sig = siggen.gmsk_channel(9, 14 * chunk, carrier_offset_hz=-1000.0)  # 57344 samples: crosses a one-second boundary
    w = C.c_void_p()
    wc = _worker_cfg(node, tmp_path, GEOM, offset=1000)
    assert L.dsp_worker_create(7, -1, C.byref(wc), C.byref(w)) == 0
    for k in range(14):
        part = np.ascontiguousarray(sig[k * chunk:(k + 1) * chunk]).view(np.float32)
        L.dsp_worker_put(part.ctypes.data, chunk, w)
    L.dsp_worker_destroy(w)
    osc = orc.Nco(1.0, 48000, chunk)
    o = orc.Fsk(*GEOM)
    want = np.concatenate([o.process(osc.multiply(1000, sig[k * chunk:(k + 1) * chunk].view(np.float32)))[0] for k in range(14)])
    got = np.fromfile(os.path.join(str(tmp_path), "rx.demod2client.7.s8"), dtype=np.int8)
    assert np.array_equal(got, want)
    # offset and Doppler together (-ENOTSUP until round 5): two oscillators in series with separate phases, every sample rounded
    # to fp32 in between -- the file source's (file_source.c:122), then the Doppler correction's (src/dsp_worker.c:65-71)
    shifts = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "doppler_shifts_lucky7.json")))["shifts_hz"]
    cb = binding.SHIFT_FN(lambda user, k: float(shifts[min(int(k), len(shifts) - 1)]))
    wc2 = _worker_cfg(node, tmp_path, GEOM, offset=1000)
    wc2.doppler_shift = C.cast(cb, C.c_void_p)
    assert L.dsp_worker_create(8, -1, C.byref(wc2), C.byref(w)) == 0
    for k in range(14):
        part = np.ascontiguousarray(sig[k * chunk:(k + 1) * chunk]).view(np.float32)
        L.dsp_worker_put(part.ctypes.data, chunk, w)
    L.dsp_worker_destroy(w)
    osc = orc.Nco(1.0, 48000, chunk)
    dop = orc.Doppler(48000, shifts, chunk)
    o = orc.Fsk(*GEOM)
    want = np.concatenate([o.process(dop.process(osc.multiply(1000, sig[k * chunk:(k + 1) * chunk].view(np.float32))))[0] for k in range(14)])
    got = np.fromfile(os.path.join(str(tmp_path), "rx.demod2client.8.s8"), dtype=np.int8)
    assert np.array_equal(got, want)
    node.close()
