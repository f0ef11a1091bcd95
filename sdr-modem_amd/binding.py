"""ctypes binding of csrc/libsdrmodem_hip.so (include/sdrmodem_hip.h).  Thin: every call goes straight to the C-ABI.

No fallback: load() raises when the shared library is absent, and the C-ABI itself returns -ENODEV with a "<3>" message
when no HIP device is usable (a failing plain handle goes into its sticky error state; nothing aborts).
"""
import ctypes as C
import errno
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SDRM_LIB_PATH") or os.path.join(HERE, "csrc", "libsdrmodem_hip.so")  # override: A/B builds
_LIB = None

f32p = C.POINTER(C.c_float)
i8p = C.POINTER(C.c_int8)


class FskConfig(C.Structure):
    """== the arguments of fsk_demod_create() (reference src/dsp/fsk_demod.h:11)"""
    _fields_ = [("sampling_freq", C.c_uint64), ("baud_rate", C.c_uint32), ("deviation", C.c_int64),
                ("decimation", C.c_uint8), ("transition_width", C.c_uint32), ("use_dc_block", C.c_bool),
                ("max_input_buffer_length", C.c_uint32)]


class FskInfo(C.Structure):
    _fields_ = [("taps1_len", C.c_uint32), ("taps2_len", C.c_uint32), ("dc_length", C.c_uint32),
                ("quad_gain", C.c_float), ("sps", C.c_float), ("gain_omega", C.c_float),
                ("gain_mu", C.c_float), ("omega_lim", C.c_float)]


class NcoSegment(C.Structure):
    _fields_ = [("channel", C.c_uint32), ("len", C.c_uint32), ("freq_hz", C.c_int64)]


NCO_DTYPE = np.dtype([("channel", "<u4"), ("len", "<u4"), ("freq_hz", "<i8")])  # sdrm_nco_segment
assert NCO_DTYPE.itemsize == C.sizeof(NcoSegment)

SHIFT_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, C.c_uint64)


class WorkerConfig(C.Structure):
    _fields_ = [("rx_sampling_freq", C.c_uint64), ("demod_baud_rate", C.c_uint32), ("demod_fsk_deviation", C.c_int64),
                ("demod_decimation", C.c_uint32), ("demod_fsk_transition_width", C.c_uint32),
                ("demod_fsk_use_dc_block", C.c_bool), ("rx_dump_file", C.c_bool), ("demod_destination", C.c_int),
                ("buffer_size", C.c_uint32), ("queue_size", C.c_uint16), ("rx_file_source", C.c_bool),
                ("base_path", C.c_char_p), ("doppler_shift", C.c_void_p), ("doppler_user", C.c_void_p),
                ("batcher", C.c_void_p), ("batcher_channel", C.c_size_t),
                ("node", C.c_void_p), ("source_id", C.c_uint64), ("rx_offset_hz", C.c_int64), ("doppler_release", C.c_void_p)]


# every symbol include/sdrmodem_hip.h declares
EXPORTS = [
    "fsk_demod_create", "fsk_demod_process", "fsk_demod_destroy",
    "sdrm_batch_create", "sdrm_batch_destroy", "sdrm_batch_channels", "sdrm_batch_info", "sdrm_batch_taps",
    "sdrm_batch_process", "sdrm_batch_process_device", "sdrm_batch_device_outputs", "sdrm_batch_last_soft",
    "sdrm_batch_fetch", "sdrm_batch_wait", "sdrm_batch_sync", "sdrm_batch_timing_enable", "sdrm_batch_timing_read", "sdrm_batch_wild_calls", "sdrm_batch_handoff_calls", "sdrm_handoff_stats", "sdrm_fsk_demod_share",
    "sdrm_batch_process_nco", "sdrm_batch_process_device_nco", "sdrm_batch_last_mixed",
    "sdrm_batch_arena", "sdrm_batch_submit", "sdrm_batch_collect", "sdrm_batch_reset_channel",
    "sdrm_dsp_worker_create", "sdrm_batcher_create", "sdrm_batcher_put", "sdrm_batcher_take", "sdrm_batcher_complete", "sdrm_batcher_interrupt", "sdrm_batcher_abandon", "sdrm_fsk_demod_error", "sdrm_last_error", "sdrm_batch_wait_input", "sdrm_wire_write_response",
    "sdrm_wire_read_header", "sdrm_wire_decode_rx_request",
    "sdrm_batcher_set_doppler", "sdrm_batcher_reset_channel", "sdrm_batcher_reset_channel_offset", "sdrm_batch_set_pre_offset", "sdrm_batcher_channels", "sdrm_batcher_rounds", "sdrm_batcher_error", "sdrm_batcher_destroy",
    "sdrm_node_create", "sdrm_node_attach", "sdrm_node_detach", "sdrm_node_batchers", "sdrm_node_stat_read",
    "sdrm_node_destroy", "sdrm_channel_cost",
    "sdrm_doppler_create", "sdrm_doppler_plan", "sdrm_doppler_destroy",
    "sdrm_probe_atan2", "sdrm_probe_quad", "sdrm_probe_boxcar_div", "sdrm_version", "sdrm_device_count",
    "sdrm_batch_k3_stamps", "sdrm_batch_timeline", "sdrm_batch_schedule",
    "create_queue", "queue_put", "take_buffer_for_processing", "complete_buffer_processing",
    "interrupt_waiting_the_data", "destroy_queue",
    "dsp_worker_create", "dsp_worker_put", "dsp_worker_shutdown", "dsp_worker_find_by_id", "dsp_worker_destroy",
]


def handoff_stats(device=-1):
    """(taken, refused, peak_waiting): in-call hand-offs admitted on the device by every batch and handle of this process, calls
    that qualified but found the device's budget of waiting workgroups taken, most workgroups waiting at once"""
    t, r, p = C.c_uint64(), C.c_uint64(), C.c_uint32()
    assert load().sdrm_handoff_stats(device, C.byref(t), C.byref(r), C.byref(p)) == 0
    return t.value, r.value, p.value


class BatcherConfig(C.Structure):
    _fields_ = [("slots", C.c_uint32), ("max_wait_us", C.c_uint32), ("blocking", C.c_bool)]


class ScheduleInfo(C.Structure):
    _fields_ = [("k3_lanes", C.c_int), ("k3_ring", C.c_int), ("k3_plain", C.c_int), ("front_hold", C.c_int),
                ("company_blocks", C.c_int), ("calibrated", C.c_int), ("ms_before", C.c_float), ("ms_after", C.c_float),
                ("ms_spent", C.c_float), ("online_state", C.c_int), ("online_choice", C.c_int), ("online_ms", C.c_float * 8)]


class NodeConfig(C.Structure):
    _fields_ = [("devices", C.POINTER(C.c_int)), ("n_batchers", C.c_size_t), ("slots_per_batcher", C.c_size_t),
                ("geometry", FskConfig), ("batcher", BatcherConfig)]


class NodeSlot(C.Structure):
    _fields_ = [("batcher", C.c_void_p), ("channel", C.c_size_t), ("device", C.c_int), ("batcher_index", C.c_size_t)]


class NodeStat(C.Structure):
    _fields_ = [("device", C.c_int), ("batcher", C.c_void_p), ("slots", C.c_size_t), ("clients", C.c_size_t),
                ("load", C.c_double), ("attached", C.c_uint64), ("error", C.c_int)]


def bind_node(L):
    """argtypes of the node front door's calls (also used on the test suite's emulation-backed build)"""
    vp = C.c_void_p
    L.sdrm_node_attach.argtypes = [vp, C.POINTER(FskConfig), C.c_uint64, C.POINTER(NodeSlot)]
    L.sdrm_node_detach.argtypes = [vp, C.POINTER(NodeSlot)]
    L.sdrm_node_batchers.argtypes = [vp]
    L.sdrm_node_batchers.restype = C.c_size_t
    L.sdrm_node_stat_read.argtypes = [vp, C.c_size_t, C.POINTER(NodeStat)]
    L.sdrm_node_destroy.argtypes = [vp]
    L.sdrm_node_destroy.restype = None


def bind_batcher(L):
    """argtypes of the batcher's per-channel calls (also used for the test suite's emulation-backed build)"""
    vp = C.c_void_p
    L.sdrm_batcher_put.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
    L.sdrm_batcher_put.restype = None
    L.sdrm_batcher_take.argtypes = [vp, C.c_size_t, C.POINTER(C.POINTER(C.c_int8)), C.POINTER(C.c_size_t)]
    L.sdrm_batcher_take.restype = None
    L.sdrm_batcher_complete.argtypes = [vp, C.c_size_t]
    L.sdrm_batcher_complete.restype = None
    L.sdrm_batcher_interrupt.argtypes = [vp, C.c_size_t]
    L.sdrm_batcher_interrupt.restype = None
    L.sdrm_batcher_abandon.argtypes = [vp, C.c_size_t]
    L.sdrm_batcher_abandon.restype = None
    L.sdrm_batcher_reset_channel.argtypes = [vp, C.c_size_t, C.POINTER(FskConfig)]
    L.sdrm_batcher_channels.argtypes = [vp]
    L.sdrm_batcher_channels.restype = C.c_size_t
    L.sdrm_batcher_rounds.argtypes = [vp]
    if hasattr(L, "sdrm_batcher_error"):  # absent only from older builds loaded through SDRM_LIB_PATH for A/B measurements
        L.sdrm_batcher_error.argtypes = [vp]
        L.sdrm_batcher_error.restype = C.c_int
    L.sdrm_batcher_rounds.restype = C.c_uint64
    L.sdrm_batcher_destroy.argtypes = [vp]
    L.sdrm_batcher_destroy.restype = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64; if torch is going to be used in this
    # process (bench.py, the device-resident test) it must be the copy that gets loaded, so import it first and let
    # our library's libamdhip64.so dependency resolve to the already-loaded one.  Without torch we use /opt/rocm's.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsdrmodem_hip.so is not built (run __graft_entry__.build() or make -C sdr-modem_amd/csrc); "
                           "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.sdrm_version.restype = C.c_char_p
    L.sdrm_device_count.restype = C.c_int
    L.sdrm_batch_create.argtypes = [C.POINTER(FskConfig), C.c_size_t, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.sdrm_batch_destroy.argtypes = [vp]
    L.sdrm_batch_destroy.restype = None
    L.sdrm_batch_channels.argtypes = [vp]
    L.sdrm_batch_channels.restype = C.c_size_t
    L.sdrm_batch_info.argtypes = [vp, C.c_size_t, C.POINTER(FskInfo)]
    L.sdrm_batch_taps.argtypes = [vp, C.c_size_t, C.c_int, f32p, C.c_size_t]
    L.sdrm_batch_taps.restype = C.c_size_t
    L.sdrm_batch_process.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(i8p), C.POINTER(C.c_size_t)]
    L.sdrm_batch_process_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp]
    L.sdrm_batch_device_outputs.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(vp), C.POINTER(vp)]
    L.sdrm_batch_last_soft.argtypes = [vp, C.c_size_t, f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdrm_batch_fetch.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.sdrm_batch_wait.argtypes = [vp, vp]
    L.sdrm_batch_wait_input.argtypes = [vp, vp]
    L.sdrm_batch_sync.argtypes = [vp]
    L.sdrm_batch_process_nco.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(NcoSegment), C.c_size_t,
                                         C.POINTER(i8p), C.POINTER(C.c_size_t)]
    L.sdrm_batch_process_device_nco.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(NcoSegment), C.c_size_t, vp]
    L.sdrm_batch_last_mixed.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    bind_batcher(L)
    if hasattr(L, "sdrm_node_create"):  # absent only from older builds loaded through SDRM_LIB_PATH for A/B measurements
        bind_node(L)
        L.sdrm_node_create.argtypes = [C.POINTER(NodeConfig), C.POINTER(vp)]
        L.sdrm_channel_cost.argtypes = [C.POINTER(FskConfig)]
        L.sdrm_channel_cost.restype = C.c_double
    L.sdrm_batcher_create.argtypes = [C.POINTER(FskConfig), C.c_size_t, C.c_int, C.POINTER(BatcherConfig), C.POINTER(vp)]
    L.sdrm_batcher_set_doppler.argtypes = [vp, C.c_size_t, vp]
    L.sdrm_batch_reset_channel.argtypes = [vp, C.c_size_t, C.POINTER(FskConfig)]
    L.sdrm_batch_arena.argtypes = [vp, C.c_size_t, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.sdrm_batch_submit.argtypes = [vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t]
    L.sdrm_batch_collect.argtypes = [vp, C.POINTER(i8p), C.POINTER(C.c_size_t)]
    L.sdrm_doppler_create.argtypes = [C.c_uint64, SHIFT_FN, vp, C.POINTER(vp)]
    L.sdrm_doppler_plan.argtypes = [vp, C.c_uint32, C.c_size_t, C.POINTER(NcoSegment), C.c_size_t]
    L.sdrm_doppler_plan.restype = C.c_size_t
    L.sdrm_doppler_destroy.argtypes = [vp]
    L.sdrm_doppler_destroy.restype = None
    if hasattr(L, "sdrm_batch_schedule"):
        L.sdrm_batch_schedule.argtypes = [vp, C.POINTER(ScheduleInfo)]
    L.sdrm_batch_timing_enable.argtypes = [vp, C.c_int]
    L.sdrm_batch_timing_read.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    if hasattr(L, "sdrm_batch_handoff_calls"):
        L.sdrm_batch_handoff_calls.argtypes = [vp, C.POINTER(C.c_uint64)]
    if hasattr(L, "sdrm_fsk_demod_share"):
        L.sdrm_fsk_demod_share.argtypes = [C.c_size_t, C.c_uint32]
    if hasattr(L, "sdrm_handoff_stats"):
        L.sdrm_handoff_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    if hasattr(L, "sdrm_batch_wild_calls"):  # absent only from older builds loaded through SDRM_LIB_PATH for A/B measurements
        L.sdrm_batch_wild_calls.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.sdrm_probe_atan2.argtypes = [vp, vp, vp, C.c_size_t]
    L.sdrm_probe_boxcar_div.argtypes = [vp, C.c_uint32, vp, C.c_size_t]
    L.sdrm_probe_quad.argtypes = [vp, C.c_size_t, C.c_float, vp, vp]
    L.fsk_demod_create.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_uint8, C.c_uint32, C.c_bool, C.c_uint32,
                                   C.POINTER(vp)]
    L.fsk_demod_process.argtypes = [vp, C.c_size_t, C.POINTER(i8p), C.POINTER(C.c_size_t), vp]
    L.fsk_demod_process.restype = None
    L.fsk_demod_destroy.argtypes = [vp]
    L.fsk_demod_destroy.restype = None
    L.create_queue.argtypes = [C.c_uint32, C.c_uint16, C.c_bool, C.POINTER(vp)]
    L.queue_put.argtypes = [vp, C.c_size_t, vp]
    L.take_buffer_for_processing.argtypes = [C.POINTER(vp), C.POINTER(C.c_size_t), vp]
    L.take_buffer_for_processing.restype = None
    L.complete_buffer_processing.argtypes = [vp]
    L.complete_buffer_processing.restype = None
    L.interrupt_waiting_the_data.argtypes = [vp]
    L.interrupt_waiting_the_data.restype = None
    L.destroy_queue.argtypes = [vp]
    L.destroy_queue.restype = None
    L.dsp_worker_create.argtypes = [C.c_uint32, C.c_int, C.POINTER(WorkerConfig), C.POINTER(vp)]
    L.dsp_worker_put.argtypes = [vp, C.c_size_t, vp]
    L.dsp_worker_put.restype = None
    L.dsp_worker_shutdown.argtypes = [vp, vp]
    L.dsp_worker_shutdown.restype = None
    L.dsp_worker_find_by_id.argtypes = [vp, vp]
    L.dsp_worker_find_by_id.restype = C.c_bool
    L.dsp_worker_destroy.argtypes = [vp]
    L.dsp_worker_destroy.restype = None
    _LIB = L
    return L


def make_configs(cfgs):
    """cfgs: iterable of (fs, baud, deviation, decimation, transition_width, use_dc, max_len)."""
    arr = (FskConfig * len(cfgs))()
    for i, (fs, baud, dev, decim, tw, dc, maxlen) in enumerate(cfgs):
        arr[i] = FskConfig(fs, baud, dev, decim, tw, dc, maxlen)
    return arr


def _as_f32(iq):
    iq = np.ascontiguousarray(iq)
    if iq.dtype == np.complex64:
        return iq.view(np.float32)
    return np.ascontiguousarray(iq, dtype=np.float32)


class Batch:
    """sdrm_batch_*: many channels per launch."""

    def __init__(self, cfgs, device=-1, keep_soft=False, calibrate=True):
        self.L = load()
        self.n = len(cfgs)
        self._cfgs = make_configs(list(cfgs))
        self.h = C.c_void_p()
        # SDRM_FLAG_KEEP_SOFT_F32 | SDRM_FLAG_NO_CALIBRATION
        flags = (1 if keep_soft else 0) | (0 if calibrate else 4)
        self.code = self.L.sdrm_batch_create(self._cfgs, self.n, device, flags, C.byref(self.h))
        if self.code != 0:
            self.h = C.c_void_p()

    def close(self):
        if self.h:
            self.L.sdrm_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self, c=0):
        inf = FskInfo()
        assert self.L.sdrm_batch_info(self.h, c, C.byref(inf)) == 0
        return inf

    def taps(self, c, stage):
        n = self.L.sdrm_batch_taps(self.h, c, stage, None, 0)
        out = np.zeros(n, dtype=np.float32)
        self.L.sdrm_batch_taps(self.h, c, stage, out.ctypes.data_as(f32p), n)
        return out

    def process(self, inputs):
        """inputs: list (len C) of complex64 / interleaved-float32 arrays (None = an empty call, ABSENT = the channel takes
        no part in the call, SDRM_LEN_ABSENT). Returns list of int8 arrays."""
        keep = [None if (x is None or x is ABSENT) else _as_f32(x) for x in inputs]
        ptrs = (C.c_void_p * self.n)(*[None if k is None or len(k) == 0 else k.ctypes.data for k in keep])
        lens = (C.c_size_t * self.n)(*[LEN_ABSENT if x is ABSENT else (0 if k is None else len(k) // 2) for x, k in zip(inputs, keep)])
        outs = (i8p * self.n)()
        olens = (C.c_size_t * self.n)()
        code = self.L.sdrm_batch_process(self.h, ptrs, lens, outs, olens)
        if code != 0:
            raise RuntimeError("sdrm_batch_process failed: %d" % code)
        res = []
        for c in range(self.n):
            n = olens[c]
            res.append(np.ctypeslib.as_array(outs[c], shape=(n,)).copy() if n else np.zeros(0, np.int8))
        return res

    def process_nco(self, inputs, segments):
        """segments: list of (channel, len, freq_hz), grouped by channel, covering each corrected channel's input."""
        keep = [None if x is None else _as_f32(x) for x in inputs]
        ptrs = (C.c_void_p * self.n)(*[None if k is None or len(k) == 0 else k.ctypes.data for k in keep])
        lens = (C.c_size_t * self.n)(*[0 if k is None else len(k) // 2 for k in keep])
        segs = (NcoSegment * max(len(segments), 1))(*[NcoSegment(*s) for s in segments])
        outs = (i8p * self.n)()
        olens = (C.c_size_t * self.n)()
        code = self.L.sdrm_batch_process_nco(self.h, ptrs, lens, segs, len(segments), outs, olens)
        if code != 0:
            raise RuntimeError("sdrm_batch_process_nco failed: %d" % code)
        return [np.ctypeslib.as_array(outs[c], shape=(olens[c],)).copy() if olens[c] else np.zeros(0, np.int8)
                for c in range(self.n)]

    def reset_channel(self, c, cfg=None):
        """hand channel c to a new stream; cfg = 7-tuple for a new configuration, None to keep the current one"""
        arr = make_configs([cfg]) if cfg is not None else None
        return self.L.sdrm_batch_reset_channel(self.h, c, arr)

    def arena(self, slots=3):
        """Pinned input arena of the pipelined host path: float32 view [slots][C][2*chan_stride] (interleaved I,Q)."""
        base, cs, ss = C.c_void_p(), C.c_size_t(), C.c_size_t()
        code = self.L.sdrm_batch_arena(self.h, slots, C.byref(base), C.byref(cs), C.byref(ss))
        if code != 0:
            raise RuntimeError("sdrm_batch_arena failed: %d" % code)
        buf = (C.c_float * (slots * ss.value * 2)).from_address(base.value)
        return np.frombuffer(buf, dtype=np.float32).reshape(slots, self.n, 2 * cs.value)

    def submit(self, slot, lens, segments=None):
        arr = (C.c_size_t * self.n)(*[int(x) for x in lens])
        segs, nseg = None, 0
        if segments:
            segs = (NcoSegment * len(segments))(*[NcoSegment(*s) for s in segments])
            nseg = len(segments)
        return self.L.sdrm_batch_submit(self.h, slot, arr, C.cast(segs, C.c_void_p) if segs is not None else None, nseg)

    def collect(self, copy=True):
        outs = (i8p * self.n)()
        olens = (C.c_size_t * self.n)()
        code = self.L.sdrm_batch_collect(self.h, outs, olens)
        if code != 0:
            raise RuntimeError("sdrm_batch_collect failed: %d" % code)
        if not copy:
            return [int(olens[c]) for c in range(self.n)]
        return [np.ctypeslib.as_array(outs[c], shape=(olens[c],)).copy() if olens[c] else np.zeros(0, np.int8)
                for c in range(self.n)]

    def last_mixed(self, c):
        n = C.c_size_t()
        if self.L.sdrm_batch_last_mixed(self.h, c, None, 0, C.byref(n)) != 0:
            raise RuntimeError("sdrm_batch_last_mixed failed")
        out = np.zeros(2 * n.value, dtype=np.float32)
        if n.value:
            self.L.sdrm_batch_last_mixed(self.h, c, out.ctypes.data, n.value, C.byref(n))
        return out

    def process_device(self, d_ptr, in_stride, lens, stream=None):
        arr = lens if isinstance(lens, C.Array) else (C.c_size_t * self.n)(*[int(x) for x in lens])
        code = self.L.sdrm_batch_process_device(self.h, C.c_void_p(d_ptr), in_stride, arr, C.c_void_p(stream or 0))
        if code != 0:
            raise RuntimeError("sdrm_batch_process_device failed: %d" % code)

    def process_device_nco(self, d_ptr, in_stride, lens, segments, stream=None, n_segments=None):
        """device-resident call with Doppler pre-correction; segments: ctypes array of NcoSegment (n_segments of it, all
        by default), an (n, 3) integer numpy array or a list of (channel, len, freq_hz) tuples"""
        arr = lens if isinstance(lens, C.Array) else (C.c_size_t * self.n)(*[int(x) for x in lens])
        if isinstance(segments, np.ndarray):  # (n, 3) integers: channel, len, freq_hz -- packed without a Python loop
            n_segments = len(segments)
            packed = np.zeros(max(n_segments, 1), dtype=NCO_DTYPE)
            packed["channel"][:n_segments], packed["len"][:n_segments], packed["freq_hz"][:n_segments] = segments[:, 0], segments[:, 1], segments[:, 2]
            segments = packed.ctypes.data_as(C.POINTER(NcoSegment))
        elif not isinstance(segments, C.Array):
            n_segments = len(segments)
            segments = (NcoSegment * max(len(segments), 1))(*[NcoSegment(*t) for t in segments])
        code = self.L.sdrm_batch_process_device_nco(self.h, C.c_void_p(d_ptr), in_stride, arr, segments,
                                                    len(segments) if n_segments is None else n_segments, C.c_void_p(stream or 0))
        if code != 0:
            raise RuntimeError("sdrm_batch_process_device_nco failed: %d" % code)

    def sync(self):
        if self.L.sdrm_batch_sync(self.h) != 0:
            raise RuntimeError("sdrm_batch_sync failed")

    def last_soft(self, c):
        n = C.c_size_t()
        code = self.L.sdrm_batch_last_soft(self.h, c, None, 0, C.byref(n))
        if code != 0:
            raise RuntimeError("sdrm_batch_last_soft failed: %d (created without keep_soft?)" % code)
        out = np.zeros(n.value, dtype=np.float32)
        if n.value:
            self.L.sdrm_batch_last_soft(self.h, c, out.ctypes.data_as(f32p), n.value, C.byref(n))
        return out

    def fetch(self, stride):
        data = np.zeros((self.n, stride), dtype=np.int8)
        lens = (C.c_size_t * self.n)()
        code = self.L.sdrm_batch_fetch(self.h, data.ctypes.data, stride, lens)
        if code != 0:
            raise RuntimeError("sdrm_batch_fetch failed: %d" % code)
        return data, np.array(list(lens), dtype=np.int64)

    def schedule(self):
        """the schedule the batch runs with (measured at creation for batches of 32 channels or more): dict"""
        if not hasattr(self.L, "sdrm_batch_schedule"):
            return None
        inf = ScheduleInfo()
        if self.L.sdrm_batch_schedule(self.h, C.byref(inf)) != 0:
            return None
        return {"clock_stage": "%dx%d%s" % (inf.k3_lanes, inf.k3_ring, "p" if inf.k3_plain else ""), "front_hold": bool(inf.front_hold),
                "company_blocks": inf.company_blocks, "calibrated": bool(inf.calibrated),
                "ms_per_call_before": round(inf.ms_before, 3), "ms_per_call_after": round(inf.ms_after, 3),
                "calibration_ms": round(inf.ms_spent, 1),
                "online": {"state": inf.online_state, "choice": inf.online_choice, "ms_per_call": [round(v, 3) for v in inf.online_ms]}}

    def timing_enable(self, on=True):
        self.L.sdrm_batch_timing_enable(self.h, 1 if on else 0)

    def timing_read(self, which):
        ms, n = C.c_double(), C.c_uint64()
        self.L.sdrm_batch_timing_read(self.h, which, C.byref(ms), C.byref(n))
        return ms.value, n.value

    def timeline_begin(self):
        """diagnostics: from the next call on (up to 64 calls) every stage records when its first workgroup started and its
        last one ended (device clock, 100 MHz ticks)"""
        self.L.sdrm_batch_timeline.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        self.L.sdrm_batch_timeline(self.h, 1, None, 0)

    def timeline_read(self):
        """rows [front start, front end, dc start, dc end, clock start, clock end] in ms, one per recorded call"""
        tl = np.zeros(64 * 6, dtype=np.uint64)
        rows = self.L.sdrm_batch_timeline(self.h, 0, tl.ctypes.data, 64)
        return tl[:rows * 6].reshape(rows, 6).astype(np.float64) / 1e5

    def handoff_calls(self):
        """calls of this batch that ran with the in-call hand-off (stages of one call resident together)"""
        n = C.c_uint64()
        assert self.L.sdrm_batch_handoff_calls(self.h, C.byref(n)) == 0
        return n.value

    def wild_calls(self):
        """channel-calls the clock stage ran from global memory (timing loop outside its tame range)"""
        n = C.c_uint64()
        assert self.L.sdrm_batch_wild_calls(self.h, C.byref(n)) == 0
        return n.value


class DopplerPlanner:
    """sdrm_doppler_*: the reference's batching of the correction (doppler.c:128-180); shift_fn(second) -> Hz."""

    def __init__(self, fs, shift_fn):
        self.L = load()
        self._cb = SHIFT_FN(lambda user, k: float(shift_fn(int(k))))
        self.h = C.c_void_p()
        self.code = self.L.sdrm_doppler_create(fs, self._cb, None, C.byref(self.h))

    def plan(self, channel, n):
        segs = (NcoSegment * 64)()
        k = self.L.sdrm_doppler_plan(self.h, channel, n, segs, 64)
        return [(int(segs[i].channel), int(segs[i].len), int(segs[i].freq_hz)) for i in range(k)]

    def __del__(self):
        if getattr(self, "h", None):
            self.L.sdrm_doppler_destroy(self.h)


class Batcher:
    """sdrm_batcher_*: many per-client producer/consumer pairs in front of one batch (include/sdrmodem_hip.h).
    `lib`/`handle` let the CPU test-suite wrap its emulation-backed build of the same host code."""

    def __init__(self, cfgs, slots=4, max_wait_us=2000, blocking=True, device=-1, lib=None, handle=None):
        self.n = len(cfgs)
        if lib is not None:
            self.L, self.h, self.code = lib, handle, 0
            return
        self.L = load()
        self._cfgs = make_configs(list(cfgs))
        self.h = C.c_void_p()
        bc = BatcherConfig(slots, max_wait_us, blocking)
        self.code = self.L.sdrm_batcher_create(self._cfgs, self.n, device, C.byref(bc), C.byref(self.h))
        if self.code != 0:
            self.h = C.c_void_p()

    def put(self, channel, iq):
        k = _as_f32(iq)
        self.L.sdrm_batcher_put(self.h, channel, k.ctypes.data, len(k) // 2)

    def take(self, channel):
        """soft bits of the channel's oldest undelivered buffer (copied, then released), or None after the poison pill"""
        out, n = i8p(), C.c_size_t()
        self.L.sdrm_batcher_take(self.h, channel, C.byref(out), C.byref(n))
        if not out:
            return None
        res = np.ctypeslib.as_array(out, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int8)
        self.L.sdrm_batcher_complete(self.h, channel)
        return res

    def interrupt(self, channel):
        self.L.sdrm_batcher_interrupt(self.h, channel)

    def abandon(self, channel):
        self.L.sdrm_batcher_abandon(self.h, channel)

    def error(self):
        return int(self.L.sdrm_batcher_error(self.h))

    def rounds(self):
        return int(self.L.sdrm_batcher_rounds(self.h))

    def reset_channel(self, channel, cfg=None):
        arr = make_configs([cfg]) if cfg is not None else None
        return self.L.sdrm_batcher_reset_channel(self.h, channel, arr)

    def set_doppler(self, channel, planner):
        return self.L.sdrm_batcher_set_doppler(self.h, channel, planner.h if planner is not None else None)

    def close(self):
        if self.h:
            self.L.sdrm_batcher_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Node:
    """sdrm_node_*: one process, a batcher per device, cost-based placement of every new client (include/sdrmodem_hip.h).
    `lib`/`handle` let the CPU test-suite wrap its emulation-backed build (virtual devices) of the same host code."""

    def __init__(self, geometry, slots_per_batcher, n_batchers=0, devices=None, batcher=(4, 2000, True), lib=None, handle=None):
        if lib is not None:
            self.L, self.h, self.code = lib, handle, 0
            return
        self.L = load()
        self.h = C.c_void_p()
        self._dev = (C.c_int * len(devices))(*devices) if devices else None
        cfg = node_config(geometry, slots_per_batcher, n_batchers, self._dev, batcher)
        self.code = self.L.sdrm_node_create(C.byref(cfg), C.byref(self.h))
        if self.code != 0:
            self.h = C.c_void_p()

    def attach(self, cfg, source_id=0):
        """-> (code, NodeSlot)"""
        slot = NodeSlot()
        arr = make_configs([cfg])
        return self.L.sdrm_node_attach(self.h, arr, source_id, C.byref(slot)), slot

    def detach(self, slot):
        return self.L.sdrm_node_detach(self.h, C.byref(slot))

    def batchers(self):
        return int(self.L.sdrm_node_batchers(self.h))

    def stat(self, i):
        st = NodeStat()
        assert self.L.sdrm_node_stat_read(self.h, i, C.byref(st)) == 0
        return st

    def close(self):
        if self.h:
            self.L.sdrm_node_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def node_config(geometry, slots_per_batcher, n_batchers=0, devices=None, batcher=(4, 2000, True)):
    cfg = NodeConfig()
    cfg.devices = devices if devices is not None else None
    cfg.n_batchers = n_batchers
    cfg.slots_per_batcher = slots_per_batcher
    cfg.geometry = FskConfig(*geometry)
    cfg.batcher = BatcherConfig(*batcher)
    return cfg


def channel_cost(cfg):
    return float(load().sdrm_channel_cost(make_configs([cfg])))


ABSENT = object()            # in a list of inputs: the channel takes no part in the call (its state stays as it is)
LEN_ABSENT = C.c_size_t(-1).value  # SDRM_LEN_ABSENT


class FskDemod:
    """fsk_demod_create/process/destroy: the reference operator (src/dsp/fsk_demod.h:11-15)."""

    def __init__(self, fs, baud, dev, decim, tw, dc, maxlen):
        self.L = load()
        self.h = C.c_void_p()
        self.code = self.L.fsk_demod_create(fs, baud, dev, decim, tw, dc, maxlen, C.byref(self.h))
        if self.code != 0:
            self.h = C.c_void_p()

    def process(self, iq):
        k = _as_f32(iq)
        out, n = i8p(), C.c_size_t()
        self.L.fsk_demod_process(k.ctypes.data, len(k) // 2, C.byref(out), C.byref(n), self.h)
        return np.ctypeslib.as_array(out, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int8)

    def close(self):
        if self.h:
            self.L.fsk_demod_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Queue:
    """create_queue & co (reference src/queue.h:10-18); host-only, works without a GPU."""

    def __init__(self, buffer_size, queue_size, blocking):
        self.L = load()
        self.h = C.c_void_p()
        self.code = self.L.create_queue(buffer_size, queue_size, blocking, C.byref(self.h))
        if self.code != 0:
            self.h = C.c_void_p()

    def put(self, iq):
        if iq is None:
            return self.L.queue_put(None, 25, self.h)
        k = _as_f32(iq)
        return self.L.queue_put(k.ctypes.data if len(k) else None, len(k) // 2, self.h)

    def put_raw(self, ptr, n):
        return self.L.queue_put(ptr, n, self.h)

    def take(self):
        buf, n = C.c_void_p(), C.c_size_t()
        self.L.take_buffer_for_processing(C.byref(buf), C.byref(n), self.h)
        if not buf.value:
            return None
        return np.ctypeslib.as_array(C.cast(buf, f32p), shape=(2 * n.value,)).copy()

    def complete(self):
        self.L.complete_buffer_processing(self.h)

    def interrupt(self):
        self.L.interrupt_waiting_the_data(self.h)

    def close(self):
        if self.h:
            self.L.destroy_queue(self.h)
            self.h = C.c_void_p()


def strerror(code):
    return errno.errorcode.get(-code, str(code))
