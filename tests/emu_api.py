"""ctypes view of tests/emu/libsdrm_emu.so: the kernel bodies of the HIP path driven thread-by-thread on the host.
TEST INFRASTRUCTURE (CPU-only suite); never used by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

import sdrm_pkg

sdrm_pkg.load()
from sdr_modem_amd.binding import FskConfig, FskInfo, NcoSegment, make_configs  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        subprocess.check_call(["make", "-s", "-C", EMU_DIR], stdout=subprocess.DEVNULL)
        # SDRM_EMU_LIB: another build of the same sources (tests/san/run.sh: AddressSanitizer / UBSan)
        L = C.CDLL(os.environ.get("SDRM_EMU_LIB") or os.path.join(EMU_DIR, "libsdrm_emu.so"))
        L.emu_create.argtypes = [C.POINTER(FskConfig), C.c_size_t, C.POINTER(C.c_void_p)]
        L.emu_reset_channel.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(FskConfig)]
        L.emu_destroy.argtypes = [C.c_void_p]
        L.emu_destroy.restype = None
        L.emu_process.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.emu_process_nco.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(NcoSegment),
                                      C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.emu_wild_calls.argtypes = [C.c_void_p]
        L.emu_wild_calls.restype = C.c_uint64
        L.emu_mixed.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.emu_mixed.restype = C.c_size_t
        L.emu_doppler_plan_stream.argtypes = [C.c_uint64, C.POINTER(C.c_double), C.c_size_t, C.POINTER(C.c_size_t),
                                              C.c_size_t, C.POINTER(NcoSegment), C.c_size_t, C.POINTER(C.c_size_t)]
        L.emu_doppler_plan_stream.restype = C.c_size_t
        L.emu_fast_atan2f_flat.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.emu_fast_atan2f_tree.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.emu_taps.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.emu_taps.restype = C.c_size_t
        L.emu_info.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(FskInfo)]
        L.emu_info.restype = None
        from sdr_modem_amd.binding import bind_batcher
        bind_batcher(L)
        L.emu_batcher_create.argtypes = [C.POINTER(FskConfig), C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, C.c_uint,
                                         C.POINTER(C.c_void_p)]
        from sdr_modem_amd.binding import bind_node, NodeConfig
        bind_node(L)
        L.emu_node_create.argtypes = [C.POINTER(NodeConfig), C.c_int, C.POINTER(C.c_void_p)]
        L.emu_node_fail_device.argtypes = [C.c_int, C.c_int]
        L.emu_node_fail_device.restype = None
        _LIB = L
    return _LIB


def emu_node(geometry, slots_per_batcher, virtual_devices, batcher=(4, 2000, True)):
    """the product's node front door (sdr-modem_amd/host/node.cpp) over `virtual_devices` emulation-backed batchers"""
    from sdr_modem_amd.binding import Node, node_config
    cfg = node_config(geometry, slots_per_batcher, 0, None, batcher)
    h = C.c_void_p()
    code = lib().emu_node_create(C.byref(cfg), virtual_devices, C.byref(h))
    assert code == 0, code
    return Node(None, 0, lib=lib(), handle=h)


def emu_batcher(cfgs, slots=4, max_wait_us=2000, blocking=True, device_delay_us=0):
    """the product's batcher host code (sdr-modem_amd/host/batcher.cpp) over the kernel emulation"""
    from sdr_modem_amd.binding import Batcher
    arr = make_configs(list(cfgs))
    h = C.c_void_p()
    code = lib().emu_batcher_create(arr, len(cfgs), slots, max_wait_us, 1 if blocking else 0, device_delay_us, C.byref(h))
    assert code == 0, code
    return Batcher(cfgs, lib=lib(), handle=h)


ABSENT = object()  # in a list of inputs: the channel takes no part in the call (SDRM_LEN_ABSENT)


class EmuBatch:
    def __init__(self, cfgs):
        self.n = len(cfgs)
        self._cfgs = make_configs(list(cfgs))
        self.h = C.c_void_p()
        self.code = lib().emu_create(self._cfgs, self.n, C.byref(self.h))

    def process(self, inputs, segments=None):
        """inputs: list of complex64 arrays. Returns (list of int8 arrays, list of float32 arrays).
        segments: optional list of (channel, len, freq_hz) NCO batches."""
        keep = [np.ascontiguousarray(x).view(np.float32) if (x is not None and x is not ABSENT) else np.zeros(0, np.float32)
                for x in inputs]
        dummy = np.zeros(2, np.float32)
        ptrs = (C.c_void_p * self.n)(*[(k.ctypes.data if len(k) else dummy.ctypes.data) for k in keep])
        lens = (C.c_size_t * self.n)(*[C.c_size_t(-1).value if x is ABSENT else len(k) // 2 for x, k in zip(inputs, keep)])
        o8 = (C.c_void_p * self.n)()
        of = (C.c_void_p * self.n)()
        ol = (C.c_size_t * self.n)()
        if segments is None:
            lib().emu_process(self.h, ptrs, lens, o8, of, ol)
        else:
            segs = (NcoSegment * max(len(segments), 1))(*[NcoSegment(*s) for s in segments])
            assert lib().emu_process_nco(self.h, ptrs, lens, segs, len(segments), o8, of, ol) == 0
        r8, rf = [], []
        for c in range(self.n):
            n = ol[c]
            r8.append(np.ctypeslib.as_array(C.cast(o8[c], C.POINTER(C.c_int8)), shape=(n,)).copy() if n else np.zeros(0, np.int8))
            rf.append(np.ctypeslib.as_array(C.cast(of[c], C.POINTER(C.c_float)), shape=(n,)).copy() if n else np.zeros(0, np.float32))
        return r8, rf

    def wild_calls(self):
        return lib().emu_wild_calls(self.h)

    def reset_channel(self, c, cfg=None):
        arr = make_configs([cfg]) if cfg is not None else None
        return lib().emu_reset_channel(self.h, c, arr)

    def mixed(self, c):
        n = lib().emu_mixed(self.h, c, None, 0)
        out = np.zeros(2 * n, np.float32)
        lib().emu_mixed(self.h, c, out.ctypes.data, n)
        return out

    def taps(self, c, stage):
        n = lib().emu_taps(self.h, c, stage, None, 0)
        out = np.zeros(n, np.float32)
        lib().emu_taps(self.h, c, stage, out.ctypes.data, n)
        return out

    def info(self, c):
        inf = FskInfo()
        lib().emu_info(self.h, c, C.byref(inf))
        return inf

    def __del__(self):
        if getattr(self, "h", None):
            lib().emu_destroy(self.h)


def doppler_plan_stream(fs, shifts, call_lens):
    """the product's Doppler planner over consecutive calls: list (per call) of [(len, freq_hz), ...]"""
    sh = (C.c_double * len(shifts))(*shifts)
    cl = (C.c_size_t * len(call_lens))(*call_lens)
    segs = (NcoSegment * (16 * len(call_lens) + 16))()
    counts = (C.c_size_t * len(call_lens))()
    lib().emu_doppler_plan_stream(fs, sh, len(shifts), cl, len(call_lens), segs, len(segs), counts)
    out, pos = [], 0
    for k in counts:
        out.append([(int(segs[pos + i].len), int(segs[pos + i].freq_hz)) for i in range(k)])
        pos += k
    return out
