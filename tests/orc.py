"""ctypes view of the CPU oracle (oracle/libsdrm_oracle.so) and of oracle/_ref/libsdrm_ref.so.

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
_LIB = None
_REF = None

f32p = C.POINTER(C.c_float)
i8p = C.POINTER(C.c_int8)


class FskInfo(C.Structure):
    _fields_ = [("taps1_len", C.c_uint32), ("taps2_len", C.c_uint32), ("dc_length", C.c_uint32),
                ("quad_gain", C.c_float), ("sps", C.c_float), ("gain_omega", C.c_float),
                ("gain_mu", C.c_float), ("omega_lim", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", ORC_DIR], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORC_DIR, "libsdrm_oracle.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(ORC_DIR, "sdrm_oracle.c")):
        build()
    L = C.CDLL(path)
    L.orc_lowpass_taps.argtypes = [C.c_float, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    L.orc_lowpass_taps.restype = C.c_int
    L.orc_fir_create.argtypes = [C.c_uint8, f32p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    L.orc_lpf_create.argtypes = [C.c_uint8, C.c_uint64, C.c_uint64, C.c_uint32, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    L.orc_fir_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    L.orc_fir_process.restype = None
    L.orc_fir_destroy.argtypes = [C.c_void_p]
    L.orc_fast_atan2f.argtypes = [C.c_float, C.c_float]
    L.orc_fast_atan2f.restype = C.c_float
    L.orc_quad_create.argtypes = [C.c_float, C.c_uint32, C.POINTER(C.c_void_p)]
    L.orc_quad_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    L.orc_quad_process.restype = None
    L.orc_quad_destroy.argtypes = [C.c_void_p]
    L.orc_dc_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.orc_dc_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.orc_dc_process.restype = None
    L.orc_dc_destroy.argtypes = [C.c_void_p]
    L.orc_mmse_interp.argtypes = [C.c_void_p, C.c_size_t, C.c_float]
    L.orc_mmse_interp.restype = C.c_float
    L.orc_clock_create.argtypes = [C.c_float] * 5 + [C.c_size_t, C.POINTER(C.c_void_p)]
    L.orc_clock_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    L.orc_clock_process.restype = None
    L.orc_clock_destroy.argtypes = [C.c_void_p]
    L.orc_fsk_create.argtypes = [C.c_uint64, C.c_uint32, C.c_int64, C.c_uint8, C.c_uint32, C.c_bool, C.c_uint32,
                                 C.POINTER(C.c_void_p)]
    L.orc_fsk_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(i8p), C.POINTER(C.c_size_t)]
    L.orc_fsk_process.restype = None
    L.orc_fsk_last_soft.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.orc_fsk_last_soft.restype = f32p
    L.orc_fsk_get_info.argtypes = [C.c_void_p, C.POINTER(FskInfo), C.POINTER(f32p), C.POINTER(f32p)]
    L.orc_fsk_get_info.restype = None
    L.orc_fsk_destroy.argtypes = [C.c_void_p]
    L.orc_nco_create.argtypes = [C.c_float, C.c_uint64, C.c_uint32, C.POINTER(C.c_void_p)]
    L.orc_nco_process.argtypes = [C.c_void_p, C.c_int64, C.c_size_t, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    L.orc_nco_process.restype = None
    L.orc_nco_multiply.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    L.orc_nco_multiply.restype = None
    L.orc_nco_destroy.argtypes = [C.c_void_p]
    L.orc_doppler_create.argtypes = [C.c_uint64, C.POINTER(C.c_double), C.c_size_t, C.c_uint32, C.POINTER(C.c_void_p)]
    L.orc_doppler_plan.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.c_size_t]
    L.orc_doppler_plan.restype = C.c_size_t
    L.orc_doppler_process.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    L.orc_doppler_process.restype = None
    L.orc_doppler_destroy.argtypes = [C.c_void_p]
    L.orc_bench_fsk.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int64, C.c_uint8,
                                C.c_uint32, C.c_bool, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.orc_bench_fsk.restype = C.c_double
    _LIB = L
    return L


def ref_lib():
    """The reference's own lpf_taps.c / dc_blocker.c / fast_atan2f.c (oracle/_ref), or None if not built."""
    global _REF
    if _REF is not None:
        return _REF
    path = os.path.join(ORC_DIR, "_ref", "libsdrm_ref.so")
    if not os.path.exists(path):
        return None
    R = C.CDLL(path)
    R.create_low_pass_filter.argtypes = [C.c_float, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(f32p), C.POINTER(C.c_size_t)]
    R.create_low_pass_filter.restype = C.c_int
    R.dc_blocker_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    R.dc_blocker_process.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(f32p), C.POINTER(C.c_size_t), C.c_void_p]
    R.dc_blocker_process.restype = None
    R.dc_blocker_destroy.argtypes = [C.c_void_p]
    R.fast_atan2f.argtypes = [C.c_float, C.c_float]
    R.fast_atan2f.restype = C.c_float
    _REF = R
    return R


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _take(ptr, n, dtype=np.float32):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def lowpass_taps(fs, fc, tw, gain=1.0):
    p, n = f32p(), C.c_size_t()
    code = lib().orc_lowpass_taps(gain, fs, fc, tw, C.byref(p), C.byref(n))
    if code != 0:
        return code, None
    out = _take(p, n.value)
    _libc.free(p)
    return 0, out


def ref_lowpass_taps(fs, fc, tw, gain=1.0):
    p, n = f32p(), C.c_size_t()
    code = ref_lib().create_low_pass_filter(gain, fs, fc, tw, C.byref(p), C.byref(n))
    if code != 0:
        return code, None
    out = _take(p, n.value)
    _libc.free(p)
    return 0, out


class Fir:
    def __init__(self, decim, fs, fc, tw, maxlen, width, taps=None):
        self.h = C.c_void_p()
        self.width = width
        if taps is None:
            code = lib().orc_lpf_create(decim, fs, fc, tw, maxlen, width, C.byref(self.h))
        else:
            t = np.ascontiguousarray(taps, dtype=np.float32)
            code = lib().orc_fir_create(decim, t.ctypes.data_as(f32p), len(t), maxlen, width, C.byref(self.h))
        self.code = code

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        p, n = f32p(), C.c_size_t()
        lib().orc_fir_process(self.h, x.ctypes.data, len(x) // self.width, C.byref(p), C.byref(n))
        return _take(p, n.value * self.width)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_fir_destroy(self.h)


class Quad:
    def __init__(self, gain, maxlen):
        self.h = C.c_void_p()
        self.code = lib().orc_quad_create(gain, maxlen, C.byref(self.h))

    def process(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        p, n = f32p(), C.c_size_t()
        lib().orc_quad_process(self.h, iq.ctypes.data, len(iq) // 2, C.byref(p), C.byref(n))
        return _take(p, n.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_quad_destroy(self.h)


class Dc:
    def __init__(self, length):
        self.h = C.c_void_p()
        self.code = lib().orc_dc_create(length, C.byref(self.h))

    def process(self, x):
        x = np.array(x, dtype=np.float32, copy=True)
        lib().orc_dc_process(self.h, x.ctypes.data, len(x))
        return x

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_dc_destroy(self.h)


class RefDc:
    def __init__(self, length):
        self.h = C.c_void_p()
        self.code = ref_lib().dc_blocker_create(length, C.byref(self.h))

    def process(self, x):
        x = np.array(x, dtype=np.float32, copy=True)
        p, n = f32p(), C.c_size_t()
        ref_lib().dc_blocker_process(x.ctypes.data, len(x), C.byref(p), C.byref(n), self.h)
        return x

    def __del__(self):
        if getattr(self, "h", None):
            ref_lib().dc_blocker_destroy(self.h)


def mmse_interp(samples, idx, mu):
    """samples: float32 array copied to a 16-byte aligned buffer; idx = position of the first of 8 samples."""
    raw = np.zeros(len(samples) + 16, dtype=np.float32)
    off = (-raw.ctypes.data // 4) % 4
    buf = raw[off:off + len(samples)]
    buf[:] = samples
    assert buf.ctypes.data % 16 == 0
    return float(lib().orc_mmse_interp(buf.ctypes.data, idx, mu))


class Clock:
    def __init__(self, omega, gain_omega, mu, gain_mu, rel_limit, maxlen):
        self.h = C.c_void_p()
        self.code = lib().orc_clock_create(omega, gain_omega, mu, gain_mu, rel_limit, maxlen, C.byref(self.h))

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        p, n = f32p(), C.c_size_t()
        lib().orc_clock_process(self.h, x.ctypes.data, len(x), C.byref(p), C.byref(n))
        return _take(p, n.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_clock_destroy(self.h)


class Fsk:
    """The oracle operator; mirrors fsk_demod_create/process/destroy."""

    def __init__(self, fs, baud, dev, decim, tw, dc, maxlen):
        self.h = C.c_void_p()
        self.code = lib().orc_fsk_create(fs, baud, dev, decim, tw, dc, maxlen, C.byref(self.h))

    def process(self, iq):
        """iq: complex64 or interleaved float32. Returns (int8 soft bits, float32 soft bits)."""
        iq = np.ascontiguousarray(iq)
        if iq.dtype == np.complex64:
            iq = iq.view(np.float32)
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        p, n = i8p(), C.c_size_t()
        lib().orc_fsk_process(self.h, iq.ctypes.data, len(iq) // 2, C.byref(p), C.byref(n))
        out = _take(p, n.value, np.int8)
        m = C.c_size_t()
        sp = lib().orc_fsk_last_soft(self.h, C.byref(m))
        return out, _take(sp, m.value)

    def info(self):
        inf, t1, t2 = FskInfo(), f32p(), f32p()
        lib().orc_fsk_get_info(self.h, C.byref(inf), C.byref(t1), C.byref(t2))
        return inf, _take(t1, inf.taps1_len), _take(t2, inf.taps2_len)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:  # (module globals are gone at interpreter shutdown)
            lib().orc_fsk_destroy(self.h)


class Nco:
    def __init__(self, amplitude, fs, maxlen):
        self.h = C.c_void_p()
        self.code = lib().orc_nco_create(amplitude, fs, maxlen, C.byref(self.h))

    def process(self, freq, n):
        p, m = f32p(), C.c_size_t()
        lib().orc_nco_process(self.h, freq, n, C.byref(p), C.byref(m))
        return _take(p, 2 * m.value)

    def multiply(self, freq, iq):
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        p, m = f32p(), C.c_size_t()
        lib().orc_nco_multiply(self.h, freq, iq.ctypes.data, len(iq) // 2, C.byref(p), C.byref(m))
        return _take(p, 2 * m.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_nco_destroy(self.h)


class Doppler:
    """orc_doppler_*: the reference's Doppler batching (doppler.c:116-190) driven by per-second shifts."""

    def __init__(self, fs, shifts, maxlen):
        self.h = C.c_void_p()
        arr = (C.c_double * len(shifts))(*shifts)
        self.code = lib().orc_doppler_create(fs, arr, len(shifts), maxlen, C.byref(self.h))

    def plan(self, n):
        lens = (C.c_uint32 * 64)()
        freqs = (C.c_int64 * 64)()
        k = lib().orc_doppler_plan(self.h, n, lens, freqs, 64)
        return [(int(lens[i]), int(freqs[i])) for i in range(k)]

    def process(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.float32)
        p, m = f32p(), C.c_size_t()
        lib().orc_doppler_process(self.h, iq.ctypes.data if len(iq) else None, len(iq) // 2, C.byref(p), C.byref(m))
        return _take(p, 2 * m.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_doppler_destroy(self.h)


def demod_stream(cfg, iq, chunk):
    """Run the oracle over a whole cf32 stream in `chunk`-sample calls; returns (int8, float32) concatenated."""
    fs, baud, dev, decim, tw, dc = cfg
    d = Fsk(fs, baud, dev, decim, tw, dc, chunk)
    assert d.code == 0
    iq = np.ascontiguousarray(iq).view(np.float32) if iq.dtype == np.complex64 else np.asarray(iq, dtype=np.float32)
    n = len(iq) // 2
    i8s, f32s = [], []
    for off in range(0, n, chunk):
        a, b = d.process(iq[2 * off: 2 * min(n, off + chunk)])
        i8s.append(a)
        f32s.append(b)
    return np.concatenate(i8s) if i8s else np.zeros(0, np.int8), np.concatenate(f32s) if f32s else np.zeros(0, np.float32)


_BENCH_ARGS = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint32, C.c_int64, C.c_uint8,
               C.c_uint32, C.c_bool, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
_TUNED = None


def tuned_lib():
    """libsdrm_oracle_tuned.so (the oracle's source with SIMD dot products, -O3 -mavx2 -mfma): a TIMING stand-in for
    libvolk's tuned kernels, different summation order, never a checker.  None when missing or the CPU lacks AVX2/FMA."""
    global _TUNED
    if _TUNED is None:
        _TUNED = False
        path = os.path.join(ORC_DIR, "libsdrm_oracle_tuned.so")
        try:
            flags = open("/proc/cpuinfo").read()
            if os.path.exists(path) and " avx2" in flags and " fma" in flags:
                L = C.CDLL(path)
                L.orc_bench_fsk.argtypes = _BENCH_ARGS
                L.orc_bench_fsk.restype = C.c_double
                _TUNED = L
        except OSError:
            _TUNED = False
    return _TUNED or None


def volk_attach(path=None):
    """Put the box's real libvolk (when it has one) behind the tuned build's FIR loops: bench.py's cpu_baseline.libvolk,
    timing only.  Returns the library name that loaded, or None."""
    L = tuned_lib()
    if L is None or not hasattr(L, "orc_volk_attach"):
        return None
    import ctypes.util
    names = [path] if path else [ctypes.util.find_library("volk"), "libvolk.so", "libvolk.so.3.1", "libvolk.so.3.0",
                                 "libvolk.so.2.5", "libvolk.so.2.4", "libvolk.so.2.2", "libvolk.so.2.0"]
    L.orc_volk_attach.argtypes = [C.c_char_p]
    for name in names:
        if name and L.orc_volk_attach(name.encode()) == 0:
            return name
    return None


def volk_detach():
    L = tuned_lib()
    if L is not None and hasattr(L, "orc_volk_detach"):
        L.orc_volk_detach.restype = None
        L.orc_volk_detach()


def bench_fsk(iq, chunk, cfg, threads, min_seconds, tuned=False):
    fs, baud, dev, decim, tw, dc = cfg
    iq = np.ascontiguousarray(iq).view(np.float32)
    secs, samples = C.c_double(), C.c_uint64()
    L = tuned_lib() if tuned else lib()
    msps = L.orc_bench_fsk(iq.ctypes.data, len(iq) // 2, chunk, fs, baud, dev, decim, tw, dc, threads, min_seconds,
                           C.byref(secs), C.byref(samples))
    return msps, secs.value, samples.value
