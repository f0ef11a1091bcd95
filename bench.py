#!/usr/bin/env python3
"""bench.py -- throughput of the GMSK/FSK demodulation hot path on MI355X.

Workload (BASELINE.json configs[2], the largest single-GPU configuration): 256 concurrent 48 kHz GMSK channels at
9600 baud per GPU -- fsk_demod_create(48000, 9600, 5000, 1, 2000, true) each -- fed one 131072-sample chunk per
channel per step (the reference's shipped buffer_size, src/resources/config.conf:11), inputs resident in HBM,
streaming state carried across steps.  A "step" = one pass of the whole path (LPF1 -> quadrature demod -> LPF2 ->
DC blocker -> M&M clock recovery -> int8 soft bits) over one chunk of every channel of this rank.
Channels are independent, so N GPUs = N shards with no data-path collective (weak scaling, the same 256 channels per
GPU at every N); the only collective is the RCCL broadcast of the channel configuration from rank 0 at setup.

`python bench.py --gpus N` starts the N ranks itself (one per GPU) when no launcher has set RANK/WORLD_SIZE; under
`python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.  Rank 0 prints ONE JSON line:
the contract's fields, `roofline` (front-end kernel), `cpu_baseline`, and -- informative sub-blocks, outside the timed
region -- `config3_sharded` (BASELINE configs[3]: 512 channels per GPU, N > 1 only), and at N = 1 `channel_sweep`,
`end_to_end` (host buffers in, soft bits out: PCIe-inclusive), `config5` / `config5_d1` (mixed rates with per-channel Doppler, at
decimation 5 / 8 and at the reference-default decimation 1), `blocking_call` (one call at a time, the call the reference's caller
blocks on: 256 and 1024 channels and one plain handle, with the in-call hand-off and with SDRM_HANDOFF=0) and
`perf_fsk_modem_style`.  Every timed region is followed by a spot check of its last call against the CPU restatement of the
reference (`verified_vs_oracle` / `*_verified`).
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS, BAUD, DEV, DECIM, TW, DC = 48000, 9600, 5000, 1, 2000, True
T1, T2 = 117, 57       # the two filters' lengths at this configuration (53 fs / (22 tw), made odd)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# exact-mode arithmetic ceiling of the front-end: a separately rounded multiply and add per tap = two packed fp32
# instructions per two component-MACs; 256 CUs x 4 SIMDs x 16 component-MACs per cycle at 2.4 GHz
VALU_EXACT_MACS = 256 * 4 * 16 * 2.4e9
DISTINCT = 32          # distinct seeded waveforms per rank; further channels are circular shifts of them
SWEEP_STEPS = 96       # timed steps per extra channel count of the sweep (the region ends with the pipeline's drain: ~2 steps)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--channels-per-gpu", type=int, default=256)
    ap.add_argument("--chunk", type=int, default=131072)
    ap.add_argument("--chunks-resident", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=3.0)
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the spot check against the oracle that follows every timed region (headline, sweep, config5)")
    ap.add_argument("--verify", action="store_true", help="(default since round 4; kept so older command lines still parse)")
    ap.add_argument("--sweep", type=str, default="512,1024,4096",
                    help="extra channel counts measured briefly at N=1 (reported under 'channel_sweep'); '' to skip")
    ap.add_argument("--no-extras", action="store_true", help="skip end_to_end / config5 / perf_fsk_modem_style / config3")
    ap.add_argument("--watchdog-seconds", type=float, default=900.0,
                    help="give up (exit code 3, message on stderr) if the whole run takes longer than this")
    return ap.parse_args()


def usable_cores():
    """threads this process may really run at once: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.999)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, (quota + period - 1) // period))
        except Exception:
            pass
    return max(1, n)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return None


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh child ranks (one per GPU, RCCL rendezvous on
    127.0.0.1) BEFORE this process imports torch or touches the GPU, relay rank 0's JSON line, fail if any rank fails.
    The parent never initialises HIP (a process that has must not be replaced or forked into GPU work on this pool)."""
    import socket
    import subprocess
    n = args.gpus
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    import tempfile
    procs, errs = [], []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        errs.append(tempfile.TemporaryFile())  # every rank's stderr is kept: a failing run says why, in the parent
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=errs[r]))
    deadline = time.time() + args.watchdog_seconds + 60.0
    line, code = None, 0
    try:
        out0, _ = procs[0].communicate(timeout=max(1.0, deadline - time.time()))
        for ln in out0.decode(errors="replace").splitlines():
            if ln.startswith("{"):
                line = ln
        for p in procs:
            rc = p.wait(timeout=max(1.0, deadline - time.time()))
            code = code or rc
    except subprocess.TimeoutExpired:
        code = 3
        sys.stderr.write("bench.py: ranks did not finish in time\n")
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()  # exactly the children started here
    if line is None:
        code = code or 4
        sys.stderr.write("bench.py: rank 0 printed no result\n")
    else:
        print(line, flush=True)
    # the ranks' stderr: rank 0's always (warnings travel with a good run too), and on failure the tail of every rank that failed
    for r, f in enumerate(errs):
        f.seek(0)
        text = f.read().decode(errors="replace")
        f.close()
        failed = code != 0 and (procs[r].returncode not in (0, None) or r == 0)
        if text and (r == 0 or failed):
            tail = text[-4000:] if failed else text[-1500:]
            sys.stderr.write("---- rank %d stderr%s (exit %s)\n%s\n" % (r, " tail" if len(text) > len(tail) else "", procs[r].returncode, tail))
    sys.exit(code)


def newest_traffic(channels, chunk):
    """HBM bytes per k1_front launch from the newest committed PMC measurement that matches the workload
    (profiles/*k1_front_traffic*.json, written by tools/pmc_k1.sh from separate FETCH_SIZE / WRITE_SIZE passes)."""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*k1_front_traffic*.json"))):
        try:
            tj = json.load(open(path))
        except Exception:
            continue
        if tj.get("channels") == channels and tj.get("chunk") == chunk and tj.get("hbm_bytes_per_launch"):
            key = str(tj.get("round", ""))
            if best is None or key >= best[2]:
                best = (tj["hbm_bytes_per_launch"], os.path.relpath(path, ROOT), key, tj)
    return (best[0], best[1], best[3]) if best else (None, None, {})


def newest_kernels(channels, chunk):
    """per-kernel figures of the newest committed PMC measurement of this workload (profiles/*_kernels.json,
    tools/collect_profiles.py): {kernel: {avg_ms_alone, algorithmic_bytes, counter_bytes, valu_issue, ...}}, ratio, source"""
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_kernels.json"))):
        try:
            kj = json.load(open(path))
        except Exception:
            continue
        if kj.get("channels") == channels and kj.get("chunk") == chunk and kj.get("kernels"):
            if best is None or str(kj.get("round", "")) >= best[0]:
                best = (str(kj.get("round", "")), kj, os.path.relpath(path, ROOT))
    return (best[1]["kernels"], best[1].get("whole_step_traffic_ratio"), best[2]) if best else ({}, None, None)


class Rig:
    """the device-resident workload of one rank: C channels x `resident` chunks of synthetic GMSK in HBM + a batch"""

    def __init__(self, torch, binding, siggen, dev, local_rank, cfgs, first_channel, chunk, resident, base=None):
        self.torch, self.C, self.N, self.R = torch, len(cfgs), chunk, resident
        n_total = resident * chunk
        k = min(self.C, DISTINCT)
        if base is None:
            base = np.stack([siggen.gmsk_channel(first_channel + i, n_total, cfgs[i][0], cfgs[i][1]) for i in range(k)])
        k = min(k, len(base))
        self.base, self.k = base, k
        base_t = torch.from_numpy(base[:k, :n_total].copy().view(np.float32).reshape(k, 2 * n_total)).to(dev)
        self.x = torch.empty((self.C, 2 * n_total), dtype=torch.float32, device=dev)
        for c in range(self.C):
            self.x[c] = torch.roll(base_t[c % k], shifts=2 * 977 * (c // k))
        del base_t
        torch.cuda.synchronize()
        self.batch = binding.Batch(cfgs, device=local_rank)
        if self.batch.code != 0:
            raise RuntimeError("sdrm_batch_create failed: %d" % self.batch.code)
        self.stream = torch.cuda.current_stream().cuda_stream
        self.lens = [chunk] * self.C
        self.n_total = n_total
        self.cfgs = list(cfgs)
        self.fed = []  # resident-chunk index of every call made so far, in order (what the spot check replays)

    def step(self, i):
        off = (i % self.R) * self.N * 8  # bytes into each channel row
        self.batch.process_device(self.x.data_ptr() + off, self.n_total, self.lens, self.stream)
        self.fed.append(i % self.R)

    def row(self, c):
        """channel c's resident waveform as the device holds it (complex64, R chunks)"""
        return np.roll(self.base[c % self.k][:self.n_total], 977 * (c // self.k))

    def kernel_ms(self):
        out = []
        for which in range(3):
            ms, n = self.batch.timing_read(which)
            out.append(ms / max(n, 1))
        return out

    def close(self):
        self.batch.close()
        self.batch = None
        del self.x
        self.torch.cuda.empty_cache()


class SpotChecker:
    """The oracle beside a device-resident run, for a few spot channels: replays exactly the calls the run made (same
    chunks in the same order, same Doppler batches) through the CPU restatement of the reference (tests/orc.py) and
    compares the LAST call's int8 and float soft bits bit for bit.  Never inside a timed region; the oracle objects keep
    their stream state, so check() may be called at several points of one run (each time after a synchronisation)."""

    def __init__(self, cfgs, rows, chunk, spots, segments_of=None):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc  # the checker -- test infrastructure, used only outside the timed regions
        self.orc, self.cfgs, self.chunk, self.spots = orc, cfgs, chunk, sorted(set(int(c) for c in spots))
        self.rows = {c: np.ascontiguousarray(rows(c)) for c in self.spots}
        self.fsk = {c: orc.Fsk(*cfgs[c][:6], chunk) for c in self.spots}
        self.segments_of = segments_of  # channel -> [(len, freq_hz)] applied to every call, or None
        self.nco = {c: orc.Nco(1.0, cfgs[c][0], chunk) for c in self.spots} if segments_of else {}
        self.done = 0
        self.last = {}

    def _advance(self, c, fed):
        o, row, n = self.fsk[c], self.rows[c], self.chunk
        last = None
        for j in fed:
            x = row[j * n:(j + 1) * n].view(np.float32)
            if self.segments_of is not None:
                pos, parts = 0, []
                for ln, freq in self.segments_of(c):
                    parts.append(self.nco[c].multiply(int(freq), x[2 * pos:2 * (pos + ln)]))
                    pos += ln
                x = np.concatenate(parts)
            last = o.process(x)
        return last

    def check(self, batch, fed, soft=True):
        """fed: the run's whole call history so far.  Returns (ok, detail)."""
        from concurrent.futures import ThreadPoolExecutor
        new = fed[self.done:]
        self.done = len(fed)
        if new:
            with ThreadPoolExecutor(max_workers=max(1, min(len(self.spots), usable_cores()))) as ex:  # ctypes releases the GIL
                for c, res in zip(self.spots, ex.map(lambda c: self._advance(c, new), self.spots)):
                    self.last[c] = res
        batch.sync()
        most = max(int(self.chunk * cf[1] / cf[0] * 1.05) + 64 for cf in self.cfgs)  # symbols per call: chunk x baud / fs
        data, olen = batch.fetch(min(self.chunk, most))
        bad = []
        for c in self.spots:
            o8, of = self.last[c]
            if olen[c] != len(o8) or not np.array_equal(data[c, :olen[c]], o8):
                bad.append((c, "int8"))
            elif soft and not np.array_equal(batch.last_soft(c).view(np.uint32), of.view(np.uint32)):
                bad.append((c, "f32"))
        return not bad, {"channels": self.spots, "calls": self.done, "mismatches": bad}


def spot_channels(channels):
    """first, the two sides of a 16-channel clock-stage workgroup boundary, a middle one, the last (partial workgroups)"""
    return sorted(set(c for c in (0, 15, 16, channels // 2 + 1, channels - 1) if 0 <= c < channels))


def end_to_end(binding, siggen, channels, chunk, calls=72, slots=4):
    """host buffers in, soft bits out (sdrm_batch_arena / _submit / _collect): every call copies its pinned slot to the
    device, runs the path and copies the soft bits back; three calls in flight.  PCIe-inclusive -- never `value`."""
    b = binding.Batch([(FS, BAUD, DEV, DECIM, TW, DC, chunk)] * channels)
    if b.code != 0:
        raise RuntimeError("create failed %d" % b.code)
    arena = b.arena(slots)
    base = np.stack([siggen.gmsk_channel(i, chunk) for i in range(8)]).view(np.float32)
    for s in range(slots):
        arena[s, :, :2 * chunk] = np.tile(np.roll(base, 2 * 977 * s, axis=1), (channels // 8, 1))
    lens = [chunk] * channels
    flight = 3
    for k in range(4):
        assert b.submit(k % slots, lens) == 0
        b.collect(copy=False)
    t0 = time.perf_counter()
    for k in range(calls):
        if k >= flight:
            b.collect(copy=False)
        assert b.submit(k % slots, lens) == 0
    for k in range(min(flight, calls)):
        b.collect(copy=False)
    dt = (time.perf_counter() - t0) / calls
    b.close()
    return {"value": round(channels * chunk / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_call": round(dt * 1e3, 3),
            "host_link_GBs": round(channels * chunk * 8 / dt / 1e9, 1), "channels": channels, "calls": calls,
            "path": "sdrm_batch_arena/_submit/_collect: pinned arena slot -> one host-to-device copy per call -> kernels -> "
                    "soft bits and counts back to pinned memory, 3 calls in flight (PCIe-inclusive)"}


WARMUP5 = 192


def config5_kinds(chunk, decimated=True):
    """the two kinds of channel of BASELINE configs[4]; decimated: d = 5 / 8 (2.5 / 5 samples per symbol, chunk-invariant),
    else the reference's default d = 1 (12.5 / 40 samples per symbol, LPF2 of 289 taps at the full rate, the clock stage's
    tail quirk active -- clock_recovery_mm.c:127-133 -- so the chunking must be the oracle's)"""
    return ((240000, 19200, 5000, 5 if decimated else 1, 2000, True, chunk), (48000, 1200, 5000, 8 if decimated else 1, 2000, True, chunk))


def config5_table(total, chunk, decimated=True):
    """BASELINE configs[4]: half 240 kHz / 19200 baud, half 48 kHz / 1200 baud.  One GPU's share interleaves them; a node-wide
    table keeps each source's channels together (heavy block first), which is what the cost-balanced contiguous shards are for."""
    heavy, light = config5_kinds(chunk, decimated)
    return [heavy if c < total // 2 else light for c in range(total)]


def config5_segments(channels, chunk):
    """three NCO batches per channel and call on the channel's own Doppler ramp: (global_channel, len, freq_hz)"""
    return np.array([(c, n, -10000 + (80 * c) % 20000 + 500 * k) for c in channels for k, n in enumerate((40000, 40000, chunk - 80000))],
                    dtype=np.int64).reshape(-1, 3)


DISTINCT5 = 16  # seeded waveforms per KIND of channel (32 in all, as the headline has); further channels are circular shifts


def config5(torch, binding, siggen, dev, cfgs, chunk, steps=24, plan_step=None, local_rank=-1, first_channel=0, warmup=None):
    """BASELINE configs[4]: every channel corrected by its own Doppler ramp (three NCO batches per channel and call) in
    front of the demodulator.  `plan_step()` returns the call's batches as [(local_channel, len, freq_hz)] -- with N > 1
    ranks that is the per-call fan-out from rank 0 (shard.fanout_nco_segments), inside the timed loop.
    Input: per kind DISTINCT5 waveforms of their own seed (a seed per channel; lanes of the clock stage that see the
    same samples would flatter it), the n-th channel of a kind takes waveform n % DISTINCT5 shifted by 977 (n // DISTINCT5)."""
    channels = len(cfgs)
    n_total = 2 * chunk
    kinds = sorted(set((c[0], c[1]) for c in cfgs), reverse=True)
    waves = {k: np.stack([siggen.gmsk_channel(0x500 + first_channel + 64 * j + i, n_total, fs=k[0], baud=k[1]) for i in range(DISTINCT5)])
             for j, k in enumerate(kinds)}
    waves_t = {k: torch.from_numpy(w.view(np.float32).reshape(DISTINCT5, 2 * n_total).copy()).to(dev) for k, w in waves.items()}
    seen = {k: 0 for k in kinds}
    pick = []
    x = torch.empty((channels, 2 * n_total), dtype=torch.float32, device=dev)
    for c, cf in enumerate(cfgs):
        k = (cf[0], cf[1])
        n = seen[k]
        seen[k] += 1
        pick.append((k, n % DISTINCT5, 977 * (n // DISTINCT5)))
        x[c] = torch.roll(waves_t[k][n % DISTINCT5], shifts=2 * 977 * (n // DISTINCT5))
    del waves_t
    b = binding.Batch(cfgs, device=local_rank)
    if b.code != 0:
        raise RuntimeError("create failed %d" % b.code)
    st = torch.cuda.current_stream().cuda_stream
    lens = [chunk] * channels

    lens_c = (binding.C.c_size_t * channels)(*lens)

    fed = []  # resident-chunk index of every call, in order (what the spot check replays)

    def step(i):
        b.process_device_nco(x.data_ptr() + (i % 2) * chunk * 8, 2 * chunk, lens_c, plan_step(), st)
        fed.append(i % 2)
    # warm-up: the pipeline's fill, then the batch's online refinement for calls with NCO batches (about 130 calls from the 17th on:
    # steady state, six blocks of eight calls, the winner's probation; sdrm_batch_schedule_info.online_*) -- outside the timed region, like the creation-time calibration
    for i in range(WARMUP5 if warmup is None else warmup):
        step(i)
    torch.cuda.synchronize()
    b.timing_enable(True)
    step.fed = fed
    step.row = lambda c: np.roll(waves[pick[c][0]][pick[c][1]], pick[c][2])
    return b, x, step


def config5_single(torch, binding, siggen, dev, channels, chunk, steps=96, verify=True, check_at=(), decimated=True):
    """configs[4] in one GPU's share (N = 1): the two kinds of channel interleaved, batches planned locally.  96 timed steps: the
    timed region ends with the pipeline's drain (about two steps' worth), which 24 steps overstated the step time by 8 % with.  After the
    timed loop (and at the step counts in `check_at`, for the tests) spot channels are compared with the oracle:
    orc.Nco on the channel's three batches per call, then orc.Fsk.  decimated=False: SURVEY 8d's other half of Config 5, the
    reference-default decimation 1 (every call is one oracle call of the same length: the tail quirk is chunk-faithful)."""
    heavy, light = config5_kinds(chunk, decimated)
    cfgs = [heavy if c % 2 == 0 else light for c in range(channels)]
    mine = config5_segments(range(channels), chunk)
    # (decimation 1: a shorter warm-up -- the spot check replays every call through the oracle, whose 289-tap LPF2 at the full rate
    # makes a 240 kHz channel cost a second per 1e6 samples; the online refinement then may not have settled, and the schedule says so)
    b, x, step = config5(torch, binding, siggen, dev, cfgs, chunk, steps, plan_step=lambda: mine, warmup=None if decimated else 48)
    checker, ok, bad = None, None, []
    if verify:
        per_channel = {}
        for c, ln, f in mine:
            per_channel.setdefault(int(c), []).append((int(ln), int(f)))
        spots = sorted(set(c for c in (0, 1, 16, 17, 2 * DISTINCT5, 2 * DISTINCT5 + 1, channels - 2, channels - 1) if 0 <= c < channels))
        if not decimated:
            spots = spots[:2] + spots[-2:]  # (the oracle's 289-tap LPF2 at the full rate: a channel costs ten of the others')
        checker = SpotChecker(cfgs, step.row, chunk, spots, segments_of=lambda c: per_channel[c])
    t0 = time.perf_counter()
    dt = 0.0
    for i in range(steps):
        step(i)
        if checker is not None and (i + 1) in check_at:
            torch.cuda.synchronize()
            dt += time.perf_counter() - t0
            good, detail = checker.check(b, step.fed)
            ok = good if ok is None else (ok and good)
            bad += detail["mismatches"]
            t0 = time.perf_counter()
    torch.cuda.synchronize()
    dt = (dt + time.perf_counter() - t0) / steps
    km = [b.timing_read(w) for w in range(3)]
    if checker is not None:
        good, detail = checker.check(b, step.fed)
        ok = good if ok is None else (ok and good)
        bad += detail["mismatches"]
    schedule = b.schedule()
    wild = b.wild_calls()
    b.close()
    del x
    torch.cuda.empty_cache()
    # the same workload on the schedule's starting point: the steady state the online refinement measured on these very calls
    # before it tried anything (sdrm_batch_schedule_info.online_ms[6]), so that the figure is not tied to the warm-up's length
    unrefined = None
    online = (schedule or {}).get("online") or {}
    if online.get("state") == 2 and len(online.get("ms_per_call", [])) == 8 and online["ms_per_call"][6] > 0:
        ms0 = online["ms_per_call"][6]
        unrefined = {"ms_per_step": ms0, "value": round(channels * chunk / ms0 / 1e3, 1), "unit": "Msamples/s",
                     "kept": "refined" if online.get("choice", 0) > 0 else "starting point"}
    return {"value": round(channels * chunk / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt * 1e3, 3),
            "without_online_refinement": unrefined,
            "verified_vs_oracle": ok, "verify_mismatches": bad, "schedule": schedule,
            "channels": channels, "steps": steps, "kernel_ms": [round(m / max(n, 1), 3) for m, n in km],
            "wild_channel_calls": wild,
            "workload": "half (240000,19200,5000,%d,2000,dc) + half (48000,1200,5000,%d,2000,dc), per-channel Doppler NCO "
                        "(3 batches per channel and call), %d seeded waveforms per kind + circular shifts, inputs in HBM"
                        % (heavy[3], light[3], DISTINCT5)}


def blocking_calls(torch, binding, siggen, dev, chunk, verify=True, calls=12):
    """The call the reference's caller blocks on (src/dsp_worker.c:75; test/perf_fsk_modem.c:70-98 times it in a loop): ONE call at a
    time, the next one only when its results are there -- nothing of an earlier call to hide the front-end and the DC blocker
    behind.  Batches of 256 and 1024 channels (device-resident input, sdrm_batch_process_device + sdrm_batch_sync) and one plain
    fsk_demod handle (host buffer in, soft bits out: fsk_demod_process as the reference calls it), each with the in-call
    hand-off (the default) and with SDRM_HANDOFF=0 (the variable is read when the batch is created), median of `calls` calls
    after 3, the last call checked against the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc  # the checker, outside the timed calls
    out = {"how": "host clock around one call at a time (enqueue + wait), median of %d after 3 warm-up calls; ms" % calls,
           "reference": "src/dsp_worker.c:75 (the caller blocks on fsk_demod_process), test/perf_fsk_modem.c:70-98"}
    cfg = (FS, BAUD, DEV, DECIM, TW, DC)
    base = np.stack([siggen.gmsk_channel(0x700 + i, 2 * chunk) for i in range(DISTINCT)])
    base_t = torch.from_numpy(base.view(np.float32).reshape(DISTINCT, 4 * chunk).copy()).to(dev)
    saved = os.environ.get("SDRM_HANDOFF")

    def with_handoff(on):
        if on:
            os.environ.pop("SDRM_HANDOFF", None)
        else:
            os.environ["SDRM_HANDOFF"] = "0"

    try:
        for channels in (256, 1024):
            x = torch.empty((channels, 4 * chunk), dtype=torch.float32, device=dev)
            for c in range(channels):
                x[c] = torch.roll(base_t[c % DISTINCT], shifts=2 * 977 * (c // DISTINCT))
            rec = {"channels": channels, "samples": chunk}
            for on in (True, False):
                with_handoff(on)
                b = binding.Batch([cfg + (chunk,)] * channels, device=dev.index)
                if b.code != 0:
                    raise RuntimeError("create failed %d" % b.code)
                st = torch.cuda.current_stream().cuda_stream
                lens = (binding.C.c_size_t * channels)(*([chunk] * channels))
                fed, ts = [], []
                for i in range(3 + calls):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    b.process_device(x.data_ptr() + (i % 2) * chunk * 8, 2 * chunk, lens, st)
                    b.sync()
                    ts.append((time.perf_counter() - t0) * 1e3)
                    fed.append(i % 2)
                key = "handoff" if on else "handoff_off"
                rec[key + "_ms"] = round(float(np.median(ts[3:])), 4)
                rec[key + "_calls_taken"] = b.handoff_calls()
                if verify:
                    ok, det = SpotChecker([cfg + (chunk,)] * channels, lambda c: np.roll(base[c % DISTINCT], 977 * (c // DISTINCT)), chunk,
                                          spot_channels(channels)).check(b, fed)
                    rec[key + "_verified"] = ok
                b.close()
            del x
            torch.cuda.empty_cache()
            out["%dx%d" % (channels, chunk)] = rec
        # one plain handle: the reference's own call, host buffer in.  Two configurations: the workload's (9600 baud: 26214 symbols
        # per call, the clock recursion's ~97 ns per symbol is 2.5 ms whatever overlaps it) and the reference perf harness's
        # (4800 baud, decimation 2: half the symbols), which is the one round 5 quoted at 1.53 ms
        for label, hcfg in (("one_handle_48000_9600", cfg), ("one_handle_48000_4800_d2", (48000, 4800, 5000, 2, 2000, True))):
            sig = siggen.gmsk_channel(0x7f0, 2 * chunk, fs=hcfg[0], baud=hcfg[1])
            rec = {"samples": chunk, "config": list(hcfg)}
            for on in (True, False):
                with_handoff(on)
                d = binding.FskDemod(*hcfg, chunk)
                o = orc.Fsk(*hcfg, chunk) if verify else None
                ts, got, want = [], None, None
                for i in range(3 + calls):
                    part = sig[(i % 2) * chunk:(i % 2 + 1) * chunk]
                    t0 = time.perf_counter()
                    got = d.process(part)
                    ts.append((time.perf_counter() - t0) * 1e3)
                    if o is not None:
                        want, _ = o.process(part)
                key = "handoff" if on else "handoff_off"
                rec[key + "_ms"] = round(float(np.median(ts[3:])), 4)
                if verify:
                    rec[key + "_verified"] = bool(np.array_equal(np.asarray(got), want))
                d.close()
            out[label] = rec
        taken, refused, peak = binding.handoff_stats(dev.index)
        out["device_ledger"] = {"taken": taken, "refused": refused, "peak_waiting_workgroups": peak}
    finally:
        if saved is None:
            os.environ.pop("SDRM_HANDOFF", None)
        else:
            os.environ["SDRM_HANDOFF"] = saved
    del base_t
    torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)
    # a wedged device queue must not hold the box until an outer limit expires: leave with a diagnostic instead
    import threading

    def give_up():
        sys.stderr.write("bench.py: no result after %.0f s -- giving up (device hang?)\n" % args.watchdog_seconds)
        sys.stderr.flush()
        os._exit(3)

    dog = threading.Timer(args.watchdog_seconds, give_up)
    dog.daemon = True
    dog.start()
    import torch
    import torch.distributed as dist
    import sdrm_pkg
    sdrm_pkg.load()
    from sdr_modem_amd import binding, siggen

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU (RCCL).  SDRM_BENCH_BACKEND=gloo lets the multi-rank code path be exercised on a box with fewer
    # GPUs than ranks (ranks then share devices: a functional check, not a measurement)
    backend = os.environ.get("SDRM_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()  # counting devices does not initialise the GPU
    if n_dev == 0:
        sys.exit("bench.py needs a GPU (the demodulator has no CPU fallback)")
    if backend != "nccl":
        local_rank %= n_dev
    elif world > n_dev:
        sys.exit("bench.py: %d ranks but only %d GPUs on this node (SDRM_BENCH_BACKEND=gloo gives a functional run with "
                 "shared devices)" % (world, n_dev))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the demodulator has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    C, N, R = args.channels_per_gpu, args.chunk, args.chunks_resident
    total_ch = C * world
    coll_dev = dev if backend == "nccl" else "cpu"  # RCCL broadcasts device tensors, gloo host tensors

    # --- configuration fan-out: rank 0 owns the channel table; RCCL broadcast (the only collective on this path)
    from sdr_modem_amd import shard
    table0 = [(FS, BAUD, DEV, DECIM, TW, DC, N)] * total_ch if rank == 0 else None
    cfgs, lo, hi = shard.fanout_configs(table0, total_ch, device=coll_dev)
    assert hi - lo == C

    rig = Rig(torch, binding, siggen, dev, local_rank, cfgs, rank * C, N, R)
    batch = rig.batch
    schedule = batch.schedule()  # what the batch measured at creation to be its fastest schedule (outside every timed region)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        t = torch.tensor([seconds], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for i in range(args.warmup):
        rig.step(i)
    barrier()
    batch.timing_enable(True)  # HIP events around each kernel, on the launch stream, inside the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        rig.step(args.warmup + i)
    torch.cuda.synchronize()
    own_done = time.perf_counter() - t0  # this rank's K steps, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    k_ms = rig.kernel_ms()
    batch.timing_enable(False)
    own_elapsed = own_done
    elapsed = max_over_ranks(elapsed)
    # every rank's own time for the same K steps (before the closing barrier it is what that rank needed): an imbalance shows here
    per_rank = [own_elapsed]
    if world > 1:
        t_all = [torch.zeros(1, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(t_all, torch.tensor([own_elapsed], dtype=torch.float64, device=coll_dev))
        per_rank = [float(t.item()) for t in t_all]

    samples_per_step = C * N * world
    msps = samples_per_step * args.steps / elapsed / 1e6

    # AFTER the timed region: the last call's soft bits (int8 and float) of a few spot channels against the CPU
    # restatement of the reference, fed the same chunks in the same order as the device was
    verify = verify_detail = None
    if not args.no_verify and rank == 0:
        checker = SpotChecker(cfgs, rig.row, N, spot_channels(C))
        verify, verify_detail = checker.check(batch, rig.fed)

    # AFTER the timed region as well: the steady-state step, fill excluded -- 33 more calls with the device timeline on, the
    # spacing of their clock stages' ends (the first call of a run cannot hide its front-end and DC blocker behind an earlier
    # call's clock stage; a 20-step region carries that once)
    steady = None
    if rank == 0:
        try:
            barrier_local = torch.cuda.synchronize
            barrier_local()
            batch.timeline_begin()
            for i in range(33):
                rig.step(args.warmup + args.steps + i)
            barrier_local()
            tl = batch.timeline_read()
            if len(tl) >= 9:
                ends = tl[:, 5]
                steady = {"ms_per_step": round(float(ends[-1] - ends[8]) / (len(ends) - 9), 4), "calls": int(len(ends) - 9),
                          "first_call_ms": round(float(ends[0] - tl[0, 0]), 4),
                          "how": "device timeline (100 MHz clock): spacing of the clock stages' ends over calls 9.. of a 33-call run "
                                 "made after the timed region; first_call_ms = the run's first call, first front-end workgroup to "
                                 "last clock-stage workgroup (its stages resident together: in-call hand-off)"}
        except Exception as exc:
            steady = {"error": str(exc)[:200]}
    elif world > 1:
        pass

    row0 = rig.x[0].cpu().numpy().view(np.complex64)[:2 * N] if rank == 0 else None
    base = rig.base
    rig.close()

    # --- BASELINE configs[3]: 4096 channels over 8 GPUs = 512 per GPU, same fan-out, its own short timed region
    config3 = None
    if world > 1 and not args.no_extras:
        c3 = 512
        table3 = [(FS, BAUD, DEV, DECIM, TW, DC, N)] * (c3 * world) if rank == 0 else None
        cfgs3, lo3, hi3 = shard.fanout_configs(table3, c3 * world, device=coll_dev)
        rig3 = Rig(torch, binding, siggen, dev, local_rank, cfgs3, rank * c3, N, 2, base=base)
        for i in range(4):
            rig3.step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(SWEEP_STEPS):
            rig3.step(i)
        barrier()
        dt3 = max_over_ranks(time.perf_counter() - t0)
        rig3.close()
        config3 = {"value": round(c3 * world * N * SWEEP_STEPS / dt3 / 1e6, 1), "unit": "Msamples/s",
                   "channels_total": c3 * world, "channels_per_gpu": c3, "steps": SWEEP_STEPS,
                   "ms_per_step": round(dt3 / SWEEP_STEPS * 1e3, 3),
                   "workload": "BASELINE configs[3] shape: %d channels sharded over %d GPUs, channel table broadcast from "
                               "rank 0 over RCCL" % (c3 * world, world)}

    # --- BASELINE configs[4]: mixed rates with per-channel Doppler over the node: cost-balanced shards of the table
    # (a 240 kHz / 397-tap channel weighs ten 48 kHz ones), the call's NCO batches planned on rank 0 and fanned out
    # with the same partition before every call -- the path's second (KB-sized) collective, inside the timed loop
    config5s = None
    if world > 1 and not args.no_extras:
        tot5 = 256 * world
        table5 = config5_table(tot5, N) if rank == 0 else None
        part5 = shard.fanout_configs(table5, tot5, device=coll_dev, balance="cost")
        segs5 = config5_segments(range(tot5), N) if rank == 0 else None
        b5, x5, step5 = config5(torch, binding, siggen, dev, part5.cfgs, N,
                                plan_step=lambda: shard.fanout_nco_segments(segs5, part5, device=coll_dev, as_array=True, capacity=4 * tot5), local_rank=local_rank)
        steps5 = 96
        barrier()
        t0 = time.perf_counter()
        for i in range(steps5):
            step5(i)
        barrier()
        dt5 = max_over_ranks(time.perf_counter() - t0)
        b5.close()
        del x5
        torch.cuda.empty_cache()
        counts = [None] * world
        dist.all_gather_object(counts, len(part5))
        config5s = {"value": round(tot5 * N * steps5 / dt5 / 1e6, 1), "unit": "Msamples/s", "channels_total": tot5,
                    "channels_per_rank": counts, "balance": "cost", "steps": steps5, "ms_per_step": round(dt5 / steps5 * 1e3, 3),
                    "workload": "BASELINE configs[4] shape: %d x (240000,19200,5000,5,2000,dc) + %d x (48000,1200,5000,8,2000,dc) "
                                "over %d GPUs, shards balanced by front-end cost, per-channel Doppler batches planned on rank 0 "
                                "and broadcast before every call" % (tot5 // 2, tot5 - tot5 // 2, world)}

    if rank == 0:
        front_ms = k_ms[0]
        achieved = (C * N * 8.0) / (front_ms * 1e-3) / 1e9 if front_ms > 0 else 0.0
        traffic, traffic_src, pmc = newest_traffic(C, N)
        per_kernel, traffic_ratio, per_kernel_src = newest_kernels(C, N)
        live = {"k1_front": k_ms[0], "k2_dc": k_ms[1], "k3_clock": k_ms[2]}
        for name, rec in per_kernel.items():
            rec["avg_ms"] = round(live.get(name, 0.0), 4)  # live, this run: HIP events on the kernel's own stream, inside the timed region
        macs_per_sample = 2 * T1 + T2
        ceiling_gbs = VALU_EXACT_MACS / macs_per_sample * 8.0 / 1e9
        out = {
            "metric": "IQ Msamples/s demodulated (whole node), 48 kHz GMSK 9600 baud",
            "value": round(msps, 3),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "ms_per_step_per_rank": {"max": round(max(per_rank) / args.steps * 1e3, 4), "min": round(min(per_rank) / args.steps * 1e3, 4)},
            "steady_state": steady,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic GMSK (BT 0.5, h 0.5, AWGN sigma 0.05; %d seeded waveforms per rank, further channels "
                    "are circular shifts), resident in HBM" % min(C, DISTINCT),
            "config": {"workload": "BASELINE configs[2]: %d concurrent 48 kHz / 9600 baud GMSK channels per GPU, "
                                   "fsk_demod(48000,9600,5000,1,2000,dc), %d-sample chunks" % (C, N),
                       "channels_per_gpu": C, "chunk_samples": N, "mode": "exact (bit-identical to the CPU restatement of the reference)",
                       "stages": "serial" if os.environ.get("SDRM_SERIAL_STAGES") else "pipelined across calls",
                       "schedule": schedule,
                       "parallelism": "channel-sharded x%d, no data-path collective" % world},
            "channels_at_realtime": int(msps * 1e6 / FS),
            "kernel_ms": {"front_lpf1_quad_lpf2": round(k_ms[0], 4), "dc_blocker": round(k_ms[1], 4),
                          "clock_recovery": round(k_ms[2], 4)},
            "roofline": {"kernel": "k1_front (LPF1+quadrature demod+LPF2)", "bound": "hbm",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": C * N * 8,
                         # every kernel of the step: live average duration (this run), algorithmic bytes, HBM bytes and issue
                         # share from the newest committed counter passes; the step's counted traffic over its algorithmic bytes
                         "kernels": per_kernel, "kernels_source": per_kernel_src, "whole_step_traffic_ratio": traffic_ratio,
                         # SURVEY 8d also defines the fused-pipeline figure: 8 B in + baud/fs B out per sample at the
                         # whole-path rate.  It is far below the front-end's because the step time is the clock
                         # recovery chain (latency-bound), not a memory stream.
                         "whole_path_hbm_frac": round(msps * 1e6 / world * (8.0 + BAUD / FS / DECIM) / 1e9 / HBM_PEAK_GBS, 5),
                         # what bit-exact arithmetic allows: 2 T1 + T2 separately rounded multiply-adds per sample on the
                         # packed fp32 pipes (no FMA, no MFMA: both fuse)
                         "exact_valu_ceiling": round(ceiling_gbs, 1),
                         "exact_valu_ceiling_frac": round(ceiling_gbs / HBM_PEAK_GBS, 4),
                         "frac_of_ceiling": round(achieved / ceiling_gbs, 4),
                         # from the same PMC passes as `traffic` (not live): the shader clock the kernel really ran at
                         # (GRBM_GUI_ACTIVE / duration; the chip lowers it under this load), the share of those cycles
                         # in which the vector pipes were issuing (SQ_INSTS_VALU x 4 / SIMDs), and the share of the
                         # issued vector instructions that are the filters' own multiplies and adds
                         "pmc": {"shader_clock_ghz": pmc.get("shader_clock_ghz"), "valu_issue_busy": pmc.get("valu_issue_busy"),
                                 "sq_insts_valu": pmc.get("sq_insts_valu"),
                                 "fir_share_of_valu": (round(C * N * macs_per_sample / 64.0 / pmc["sq_insts_valu"], 4)
                                                       if pmc.get("sq_insts_valu") else None),
                                 "frac_of_ceiling_at_measured_clock": (round(achieved / (ceiling_gbs * pmc["shader_clock_ghz"] / 2.4), 4)
                                                                       if pmc.get("shader_clock_ghz") else None)},
                         "note": "8 B of IQ read per input sample (SURVEY 8d LPF-stage HBM-read term). The kernel is "
                                 "fp32-VALU-bound: bit-exact parity needs a separately rounded multiply and add per tap "
                                 "(v_pk_mul_f32 + v_pk_add_f32: 16 component-MACs per SIMD and cycle; %d MACs per sample), "
                                 "which caps it at exact_valu_ceiling GB/s of IQ. THE STEP of this workload is not this kernel: "
                                 "it is the clock-recovery recursion (k3_clock, one lane per channel, 16 waves on the chip), "
                                 "a float-recursive loop with a data-dependent stride at its floor of ~36 vector instructions "
                                 "x ~4.4 cycles + one exposed LDS round trip per symbol; `frac` describes the LPF stage the north "
                                 "star asks about, whole_path_hbm_frac the step" % macs_per_sample},
        }
        if verify is not None:
            out["verified_vs_oracle"] = verify
            out["verify"] = dict(verify_detail, what="int8 and float soft bits of the last timed call, spot channels, bit for bit "
                                                     "against the CPU restatement of the reference fed the same calls")
        if config3 is not None:
            out["config3_sharded"] = config3
        if config5s is not None:
            out["config5_sharded"] = config5s
        if world == 1 and not args.no_extras:
            # BASELINE configs[4]'s mix, first among the side blocks (the ones below create and close several batches).
            # Alone in a process (tools/config5.py) it runs ~12 % faster than here beside the headline's live batch: its
            # clock-stage workgroups then find their CUs sooner between calls (profiles/r03_clock_early.txt, last paragraph)
            try:
                out["config5"] = config5_single(torch, binding, siggen, dev, C, N, verify=not args.no_verify)
            except Exception as exc:  # informative sub-blocks: never cost the headline its line
                out["config5"] = {"error": str(exc)[:200]}
            try:  # SURVEY 8d Config 5, the other half: the reference-default decimation 1 (tail quirk active)
                out["config5_d1"] = config5_single(torch, binding, siggen, dev, C, N, steps=48, verify=not args.no_verify, decimated=False)
            except Exception as exc:
                out["config5_d1"] = {"error": str(exc)[:200]}
            try:
                out["blocking_call"] = blocking_calls(torch, binding, siggen, dev, N, verify=not args.no_verify)
            except Exception as exc:
                out["blocking_call"] = {"error": str(exc)[:200]}
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import orc
            cores = usable_cores()
            cfg6 = (FS, BAUD, DEV, DECIM, TW, DC)
            one, _, _ = orc.bench_fsk(row0, N, cfg6, 1, min(2.0, args.cpu_seconds))
            allc, secs, smp = orc.bench_fsk(row0, N, cfg6, cores, args.cpu_seconds)
            out["cpu_baseline"] = {
                "value": round(allc, 3), "unit": "Msamples/s", "cores": cores, "cpu_model": cpu_model(), "kind": "port",
                "single_thread_value": round(one, 3),
                "sample": "oracle (plain-C restatement of the reference path, gcc -O2 -ffp-contract=off), one "
                          "independent channel per thread on %d threads, %d-sample chunks of channel 0 looped for "
                          "%.1f s wall (%.0f Msamples total)" % (cores, N, secs, smp / 1e6)}
            if orc.tuned_lib() is not None:
                # second figure: the same code with SIMD dot products (per-lane partial sums, -O3 -mavx2 -mfma), standing in
                # for libvolk's tuned kernels, which the reference uses outside its tests; not bit-exact, timing only
                t_one, _, _ = orc.bench_fsk(row0, N, cfg6, 1, min(2.0, args.cpu_seconds), tuned=True)
                t_all, _, _ = orc.bench_fsk(row0, N, cfg6, cores, args.cpu_seconds, tuned=True)
                out["cpu_baseline"]["simd_stand_in"] = {
                    "value": round(t_all, 3), "single_thread_value": round(t_one, 3), "unit": "Msamples/s", "cores": cores,
                    "note": "oracle source built -O3 -mavx2 -mfma with vectorised FIR dot products (different summation "
                            "order: not the pinned arithmetic, never used as a checker)"}
                # third figure, when the box has it: the real libvolk's dot-product kernels behind the same loops
                volk = orc.volk_attach()
                if volk:
                    v_one, _, _ = orc.bench_fsk(row0, N, cfg6, 1, min(2.0, args.cpu_seconds), tuned=True)
                    v_all, _, _ = orc.bench_fsk(row0, N, cfg6, cores, args.cpu_seconds, tuned=True)
                    orc.volk_detach()
                    out["cpu_baseline"]["libvolk"] = {"value": round(v_all, 3), "single_thread_value": round(v_one, 3),
                                                      "unit": "Msamples/s", "cores": cores, "library": volk}
                else:
                    out["cpu_baseline"]["libvolk"] = None  # dlopen("libvolk.so*") found nothing on this box
        if world == 1 and not args.no_cpu_baseline and not args.no_extras:
            # the reference's own perf harness (test/perf_fsk_modem.c:70-98): one handle, 100 calls of 4096 samples
            # `re = (uint8_t) i, im = 0`, fsk_demod_create(48000, 4800, 5000, 2, 2000, true, 2016000); its published
            # figures are seconds per 100 calls on one CPU core.  One channel is one
            # sequential chain: this is a latency figure, not what the GPU path is built for.
            ramp = np.zeros(4096, dtype=np.complex64)
            ramp.real = (np.arange(4096) % 256).astype(np.float32)
            d1 = binding.FskDemod(48000, 4800, 5000, 2, 2000, True, 2016000)
            o1 = orc.Fsk(48000, 4800, 5000, 2, 2000, True, 2016000)
            for _ in range(10):
                d1.process(ramp)
                o1.process(ramp)
            t_gpu = t_cpu = 1e9
            for _ in range(3):  # best of three: the host cores have just run the multi-thread baseline
                t0 = time.perf_counter()
                for _ in range(100):
                    d1.process(ramp)
                t_gpu = min(t_gpu, time.perf_counter() - t0)
                t0 = time.perf_counter()
                for _ in range(100):
                    o1.process(ramp)
                t_cpu = min(t_cpu, time.perf_counter() - t0)
            d1.close()
            out["perf_fsk_modem_style"] = {"seconds_per_100x4096_gpu_one_handle": round(t_gpu, 5),
                                           "seconds_per_100x4096_cpu_port_one_thread": round(t_cpu, 5),
                                           "note": "reference publishes 0.0368 s (MacBook Air M1) and 0.656 s (Raspberry "
                                                   "Pi 3) for this loop; includes the Python call overhead here"}
        if world == 1 and args.sweep:
            # the named workload (256 channels) is bounded by the sequential clock-recovery chain of 16 waves; show how
            # the same pipeline fills the GPU with more channels (short runs, 2 resident chunks)
            sweep = {}
            for c2 in [int(v) for v in args.sweep.split(",") if v.strip()]:
                try:
                    r2 = Rig(torch, binding, siggen, dev, local_rank, [(FS, BAUD, DEV, DECIM, TW, DC, N)] * c2, 0, N, 2, base=base)
                    for i in range(2):
                        r2.step(i)
                    torch.cuda.synchronize()
                    r2.batch.timing_enable(True)
                    t0 = time.perf_counter()
                    for i in range(SWEEP_STEPS):
                        r2.step(i)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                    km = r2.kernel_ms()
                    ok2 = None
                    if not args.no_verify:
                        ok2, _ = SpotChecker(r2.cfgs, r2.row, N, spot_channels(c2)).check(r2.batch, r2.fed)
                    sweep[str(c2)] = {"verified_vs_oracle": ok2, "schedule": r2.batch.schedule(), "value": round(c2 * N * SWEEP_STEPS / dt / 1e6, 1), "unit": "Msamples/s",
                                      "ms_per_step": round(dt / SWEEP_STEPS * 1e3, 3), "steps": SWEEP_STEPS,
                                      "kernel_ms": [round(m, 3) for m in km],
                                      "front_hbm_frac": round(c2 * N * 8.0 / (km[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      # the front-end's share of its exact-mode arithmetic ceiling (nominal 2.4 GHz) at this size
                                      "front_frac_of_ceiling": round(c2 * N * 8.0 / (km[0] * 1e-3) / 1e9 /
                                                                     (VALU_EXACT_MACS / (2 * T1 + T2) * 8.0 / 1e9), 4)}
                    r2.close()
                except Exception as exc:  # the sweep is informative only
                    sweep[str(c2)] = {"error": str(exc)[:200]}
            out["channel_sweep"] = sweep
        if world == 1 and not args.no_extras:
            try:
                out["end_to_end"] = end_to_end(binding, siggen, C, N)
                # the reference's boundary hands over HOST buffers: what the same box sustains through it (PCIe-bound)
                out["channels_at_realtime_host_path"] = int(out["end_to_end"]["value"] * 1e6 / FS)
            except Exception as exc:
                out["end_to_end"] = {"error": str(exc)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
