"""Host logic of the per-GPU batcher (sdr-modem_amd/host/batcher.cpp) without a GPU: the same C++ code,
built over the kernel emulation (tests/emu), is driven by producer and consumer threads and compared with the oracle.
Checks the reference queue's behaviours (src/queue.c) as they appear through the batcher: per-channel order, blocking
put when all rounds are taken, overwrite-newest for live sources, buffers put before the poison pill still delivered."""
import threading
import time

import numpy as np

import emu_api
import orc
import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd import siggen  # noqa: E402

CFG_A = (48000, 9600, 5000, 1, 2000, True, 4096)
CFG_B = (48000, 4800, 5000, 2, 2000, False, 4096)


def oracle_stream(cfg, chunks):
    o = orc.Fsk(*cfg)
    return [o.process(c)[0] for c in chunks]


def test_many_clients_one_batch_ordered_and_bit_exact():
    cfgs = [CFG_A, CFG_B, CFG_A, CFG_A, CFG_B, CFG_A]
    K = 5
    sizes = [4096, 1000, 4096, 37, 2500]
    sigs = [siggen.gmsk_channel(i, sum(sizes), fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    chunks = [[s[sum(sizes[:k]):sum(sizes[:k + 1])] for k in range(K)] for s in sigs]
    bt = emu_api.emu_batcher(cfgs, slots=4, max_wait_us=200000, blocking=True)
    got = [[] for _ in cfgs]
    start = threading.Barrier(len(cfgs))

    def producer(c):
        start.wait()
        for k in range(K):
            bt.put(c, chunks[c][k])

    def consumer(c):
        for k in range(K):
            got[c].append(bt.take(c))

    th = [threading.Thread(target=f, args=(c,)) for c in range(len(cfgs)) for f in (producer, consumer)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
        assert not t.is_alive()
    for c, cfg in enumerate(cfgs):
        exp = oracle_stream(cfg, chunks[c])
        for k in range(K):
            assert np.array_equal(got[c][k], exp[k]), (c, k)
    # one batched call per buffer index, not one per client buffer
    assert bt.rounds() <= 2 * K + 2, bt.rounds()  # (a round also goes 200 ms after its first buffer: slack for a loaded machine)
    bt.close()


def test_round_without_a_clients_buffer_leaves_that_clients_stream_alone():
    """A round that is launched because its wait ran out goes without the buffers of the clients that were late.  Those
    channels must be ABSENT from the device call, not given an empty one: at 16 samples per symbol the clock stage answers
    an empty call with a re-emitted symbol (clock_recovery_mm.c:127-135) and the late client's stream would change.
    Two brisk clients keep rounds going while a slow one (16 samples per symbol) delivers a buffer now and then."""
    slow = (96000, 1200, 5000, 5, 4000, False, 4096)
    cfgs = [slow, CFG_A, CFG_A]
    sizes = [[3000, 3000, 4096, 500, 1, 3000], [600] * 30, [600] * 30]
    sigs = [siggen.gmsk_channel(20 + i, sum(sz), fs=c[0], baud=c[1]) for i, (c, sz) in enumerate(zip(cfgs, sizes))]
    chunks = [[s[sum(sz[:k]):sum(sz[:k + 1])] for k in range(len(sz))] for s, sz in zip(sigs, sizes)]
    bt = emu_api.emu_batcher(cfgs, slots=4, max_wait_us=300, blocking=True)
    got = [[] for _ in cfgs]

    def producer(c):
        for k in range(len(sizes[c])):
            time.sleep(0.02 if c == 0 else 0.004)  # both far beyond the round's wait
            bt.put(c, chunks[c][k])

    def consumer(c):
        for k in range(len(sizes[c])):
            got[c].append(bt.take(c))

    th = [threading.Thread(target=f, args=(c,)) for c in range(len(cfgs)) for f in (producer, consumer)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
        assert not t.is_alive()
    assert bt.rounds() > 12  # most rounds went without the slow client
    for c, cfg in enumerate(cfgs):
        exp = oracle_stream(cfg, chunks[c])
        for k in range(len(sizes[c])):
            assert np.array_equal(got[c][k], exp[k]), (c, k)
    bt.close()


def test_blocking_producer_waits_for_consumers():
    bt = emu_api.emu_batcher([CFG_A], slots=4, max_wait_us=100, blocking=True)
    sig = siggen.gmsk_channel(1, 6 * 2048)
    done = []

    def producer():
        for k in range(6):
            bt.put(0, sig[k * 2048:(k + 1) * 2048])
            done.append(k)

    t = threading.Thread(target=producer)
    t.start()
    time.sleep(0.5)
    assert t.is_alive() and len(done) == 4, done  # four rounds taken, nothing consumed: the fifth put blocks
    outs = [bt.take(0) for _ in range(6)]
    t.join(10)
    assert not t.is_alive()
    exp = oracle_stream(CFG_A, [sig[k * 2048:(k + 1) * 2048] for k in range(6)])
    for k in range(6):
        assert np.array_equal(outs[k], exp[k]), k
    bt.close()


def test_live_producer_overwrites_its_newest_pending_buffer(capfd):
    # channel 1 never delivers, so with a long deadline channel 0's rounds stay un-launched ("filled, not yet taken")
    bt = emu_api.emu_batcher([CFG_A, CFG_A], slots=4, max_wait_us=30_000_000, blocking=False)
    sig = siggen.gmsk_channel(2, 6 * 1024)
    parts = [sig[k * 1024:(k + 1) * 1024] for k in range(6)]
    for k in range(5):
        bt.put(0, parts[k])  # buffers 0..3 take the four rounds, buffer 4 replaces buffer 3
    assert "queue is full" in capfd.readouterr().err
    bt.interrupt(1)  # rounds stop waiting for the silent channel
    outs = [bt.take(0) for _ in range(4)]
    exp = oracle_stream(CFG_A, [parts[0], parts[1], parts[2], parts[4]])
    for k in range(4):
        assert np.array_equal(outs[k], exp[k]), k
    bt.close()


def test_live_producer_drops_when_everything_pending_is_on_the_device(capfd):
    bt = emu_api.emu_batcher([CFG_A], slots=4, max_wait_us=10, blocking=False)
    sig = siggen.gmsk_channel(3, 6 * 1024)
    parts = [sig[k * 1024:(k + 1) * 1024] for k in range(6)]
    for k in range(4):
        bt.put(0, parts[k])
    t0 = time.time()
    while bt.rounds() < 4 and time.time() - t0 < 10:
        time.sleep(0.01)
    bt.put(0, parts[4])  # no free round, nothing left to overwrite
    assert "queue is full" in capfd.readouterr().err
    outs = [bt.take(0) for _ in range(4)]
    exp = oracle_stream(CFG_A, parts[:4])
    for k in range(4):
        assert np.array_equal(outs[k], exp[k]), k
    bt.put(0, parts[5])  # rounds were released by the consumer: accepted again
    o = orc.Fsk(*CFG_A)
    for p in parts[:4]:
        o.process(p)
    assert np.array_equal(bt.take(0), o.process(parts[5])[0])
    bt.close()


def test_poison_pill_delivers_pending_buffers_then_null():
    bt = emu_api.emu_batcher([CFG_A, CFG_B], slots=4, max_wait_us=100, blocking=True)
    sig = siggen.gmsk_channel(4, 2 * 3000)
    bt.put(0, sig[:3000])
    bt.put(0, sig[3000:])
    bt.interrupt(0)
    bt.put(0, sig[:100])  # ignored after the pill
    exp = oracle_stream(CFG_A, [sig[:3000], sig[3000:]])
    assert np.array_equal(bt.take(0), exp[0])
    assert np.array_equal(bt.take(0), exp[1])
    assert bt.take(0) is None
    # a consumer blocked on an idle channel is released by the pill
    res = []
    t = threading.Thread(target=lambda: res.append(bt.take(1)))
    t.start()
    time.sleep(0.2)
    assert t.is_alive()
    bt.interrupt(1)
    t.join(10)
    assert res == [None]
    bt.close()


def test_oversize_buffer_is_refused_with_the_reference_message(capfd):
    bt = emu_api.emu_batcher([CFG_A], slots=4, max_wait_us=100, blocking=True)
    sig = siggen.gmsk_channel(5, 5000)
    bt.put(0, sig)  # 5000 > max_input_buffer_length 4096
    assert "is more than max" in capfd.readouterr().err
    bt.put(0, sig[:4096])
    assert np.array_equal(bt.take(0), oracle_stream(CFG_A, [sig[:4096]])[0])
    bt.close()


def test_channel_changes_hands_while_the_others_keep_streaming():
    """sdrm_batcher_reset_channel: client A leaves channel 1 (poison pill), client B takes the slot with another
    configuration; channel 0 streams across the hand-over without a glitch; B's soft bits are those of a fresh
    demodulator."""
    big = (48000, 4800, 5000, 2, 2000, True, 4096)
    bt = emu_api.emu_batcher([big, big], slots=4, max_wait_us=2000, blocking=True)
    s0 = siggen.gmsk_channel(60, 4 * 4096, fs=48000, baud=4800)
    sa = siggen.gmsk_channel(61, 4096, fs=48000, baud=4800)
    sb = siggen.gmsk_channel(62, 2 * 4096, fs=48000, baud=9600)
    o0, oa = orc.Fsk(*big), orc.Fsk(*big)
    bt.put(0, s0[:4096])
    bt.put(1, sa)
    assert np.array_equal(bt.take(0), o0.process(s0[:4096])[0])
    assert np.array_equal(bt.take(1), oa.process(sa)[0])
    bt.interrupt(1)
    assert bt.take(1) is None
    cfg_b = (48000, 9600, 5000, 1, 2000, False, 4096)
    # a put that races with the hand-over: channel 0 keeps going meanwhile
    t = threading.Thread(target=lambda: bt.put(0, s0[4096:8192]))
    t.start()
    assert bt.reset_channel(1, cfg_b) == 0
    t.join(10)
    ob = orc.Fsk(*cfg_b)
    bt.put(1, sb[:4096])
    assert np.array_equal(bt.take(0), o0.process(s0[4096:8192])[0])
    assert np.array_equal(bt.take(1), ob.process(sb[:4096])[0])
    bt.put(0, s0[8192:12288])
    bt.put(1, sb[4096:])
    assert np.array_equal(bt.take(0), o0.process(s0[8192:12288])[0])
    assert np.array_equal(bt.take(1), ob.process(sb[4096:])[0])
    # a third client whose configuration needs longer filters and a DC boxcar longer than anything in the batch (207 taps,
    # 1280 samples): the batch grows under the running channel 0 (round 3; refused with -ENOTSUP before)
    bt.interrupt(1)
    assert bt.take(1) is None
    cfg_c = (48000, 1200, 5000, 1, 2000, True, 4096)
    assert bt.reset_channel(1, cfg_c) == 0
    oc = orc.Fsk(*cfg_c)
    sc = siggen.gmsk_channel(77, 4096, fs=48000, baud=1200)
    s0b = siggen.gmsk_channel(78, 4096)
    bt.put(0, s0b)
    bt.put(1, sc)
    assert np.array_equal(bt.take(0), o0.process(s0b)[0])
    assert np.array_equal(bt.take(1), oc.process(sc)[0])
    # the one thing that cannot grow: the buffer length
    bt.interrupt(1)
    assert bt.take(1) is None
    assert bt.reset_channel(1, (48000, 9600, 5000, 1, 2000, True, 8192)) != 0
    bt.close()


def test_a_dead_consumer_does_not_stall_the_other_clients():
    """A client whose consumer is gone (socket or disk error: the worker leaves its loop) must not hold the shared rounds:
    in the reference a dead consumer only fills its own queue (src/queue.c:99-154).  Channel 1 puts two buffers and never
    takes them; after sdrm_batcher_abandon its rounds retire, channel 0 streams on past the number of rounds the batcher
    owns, and the slot can be handed to a new client, whose stream starts clean."""
    cfgs = [CFG_A, CFG_A]
    n = 12  # more buffers than the batcher has rounds (4): they have to be recycled
    sig0 = siggen.gmsk_channel(0, 4096 * n)
    sig1 = siggen.gmsk_channel(1, 4096 * 4)
    chunks0 = [sig0[k * 4096:(k + 1) * 4096] for k in range(n)]
    bt = emu_api.emu_batcher(cfgs, slots=4, max_wait_us=2000, blocking=True)
    bt.put(1, sig1[:4096])
    bt.put(1, sig1[4096:8192])
    bt.abandon(1)  # consumer of channel 1 died: nobody will ever take those two
    assert bt.take(1) is None
    got = []

    def client0():
        for k in range(n):
            bt.put(0, chunks0[k])
            got.append(bt.take(0))

    t = threading.Thread(target=client0)
    t.start()
    t.join(60)
    assert not t.is_alive(), "channel 0 is stuck behind the dead client's rounds"
    exp = oracle_stream(CFG_A, chunks0)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp))
    # the slot serves a new client: reset returns (nothing of the old one is waited for) and the stream starts from scratch
    done = []
    r = threading.Thread(target=lambda: done.append(bt.reset_channel(1, None)))
    r.start()
    r.join(30)
    assert not r.is_alive() and done == [0]
    sig_b = siggen.gmsk_channel(7, 4096)
    bt.put(1, sig_b)
    assert np.array_equal(bt.take(1), oracle_stream(CFG_A, [sig_b])[0])
    bt.close()


def _failing_device(fail_submit, fail_collect):
    """three clients stream four buffers each; the device call of the third round fails"""
    cfgs = [CFG_A, CFG_B, CFG_A]
    sigs = [siggen.gmsk_channel(40 + i, 4 * 2000, fs=c[0], baud=c[1]) for i, c in enumerate(cfgs)]
    bt = emu_api.emu_batcher(cfgs, slots=4, max_wait_us=200000, blocking=True)
    emu_api.lib().emu_batcher_inject(3 if fail_submit else 0, 3 if fail_collect else 0)
    got = [[] for _ in cfgs]
    ended = [None] * len(cfgs)

    def producer(c):
        for k in range(4):
            bt.put(c, sigs[c][k * 2000:(k + 1) * 2000])

    def consumer(c):
        while True:
            r = bt.take(c)
            if r is None:
                ended[c] = bt.error()
                return
            got[c].append(r)

    th = [threading.Thread(target=f, args=(c,)) for c in range(len(cfgs)) for f in (producer, consumer)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
        assert not t.is_alive(), "a client is still waiting on a dead device"
    emu_api.lib().emu_batcher_inject(0, 0)
    return bt, cfgs, sigs, got, ended


def test_a_failed_device_call_ends_every_client_with_an_error_code(capfd):
    """A batched call that fails must not look like "no symbols for ever after": take() returns NULL for every client,
    the blocked producers return, and sdrm_batcher_error() holds the device's code (the worker mirror and the shared
    fsk_demod handles turn that into their sticky error) -- for a failed submit and for a failed collect."""
    for fail_submit, fail_collect in ((True, False), (False, True)):
        bt, cfgs, sigs, got, ended = _failing_device(fail_submit, fail_collect)
        assert ended == [-5, -5, -5], ended
        assert bt.error() == -5
        for c, cfg in enumerate(cfgs):
            # what was delivered before the failure is the oracle's stream; never an empty stand-in for a lost buffer
            assert len(got[c]) <= 2, (c, len(got[c]))
            exp = oracle_stream(cfg, [sigs[c][k * 2000:(k + 1) * 2000] for k in range(len(got[c]))])
            for k in range(len(got[c])):
                assert np.array_equal(got[c][k], exp[k]), (c, k)
        # later puts are dropped, later takes return at once
        bt.put(0, sigs[0][:100])
        assert bt.take(0) is None
        # and the dead batcher hands no slot to a new client: reset_channel answers with the device's code, with the same
        # configuration or a new one, and the channel stays closed (put dropped, take returns at once)
        assert bt.reset_channel(0) == -5
        assert bt.reset_channel(1, cfgs[0]) == -5
        bt.put(0, sigs[0][:100])
        assert bt.take(0) is None
        bt.close()
    assert "<3>batcher" in capfd.readouterr().err
