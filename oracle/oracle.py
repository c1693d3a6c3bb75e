"""TEST INFRASTRUCTURE ONLY: ctypes loader for the CPU oracle (oracle/psd_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
The shipped library never does.  See psd_oracle.c for the parity status.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_psd.so")


def build(force=False):
    """Compile oracle/liboracle_psd.so with gcc (oracle/Makefile)."""
    src = [os.path.join(_HERE, f) for f in ("psd_oracle.c", "psd_oracle_impl.h", "hbf_taps_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B", "liboracle_psd.so"], check=True,
                   stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Break(C.Structure):
    """Break (src/psd.rs:290-311); bins: Range<usize> flattened."""
    _fields_ = [("start", C.c_uint64), ("include", C.c_uint32), ("count", C.c_uint32),
                ("avg", C.c_uint32), ("_pad", C.c_uint32),
                ("bins_start", C.c_uint64), ("bins_end", C.c_uint64),
                ("fft_size", C.c_uint64), ("decimation", C.c_uint64),
                ("pending", C.c_uint64), ("processed", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "_pad"}


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _declare(_lib)
    return _lib


def _declare(L):
    vp, i32, u32, u64, sz = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_size_t
    fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
    for sfx, rp, rt in (("_f32", fp, C.c_float), ("_f64", dp, C.c_double)):
        def f(name, res, args):
            fn = getattr(L, name + sfx)
            fn.restype, fn.argtypes = res, args
        f("ora_cascade_new", vp, [i32, i32])
        f("ora_cascade_free", None, [vp])
        f("ora_cascade_set_avg", None, [vp, u32, u32])
        f("ora_cascade_set_fast_fft", i32, [vp])
        f("ora_cascade_set_detrend", i32, [vp, i32])
        f("ora_cascade_process", i32, [vp, fp, sz])
        f("ora_cascade_num_stages", i32, [vp])
        f("ora_cascade_ref_would_panic", i32, [vp])
        f("ora_cascade_stage_info", i32, [vp, i32, C.POINTER(u32), C.POINTER(u32),
                                         C.POINTER(u64), C.POINTER(u64)])
        f("ora_cascade_stage_spectrum", i32, [vp, i32, rp])
        f("ora_cascade_stage_buf", i32, [vp, i32, rp])
        f("ora_cascade_stage_gain", rt, [vp, i32])
        f("ora_cascade_psd", C.c_long, [vp, i32, u32, i32, rp, C.POINTER(Break)])
        f("ora_cascade_new_window", vp, [i32, fp, C.c_float, C.c_float, i32])
        f("ora_stage_new_window", vp, [i32, fp, C.c_float, C.c_float, i32])
        f("ora_stage_new", vp, [i32, i32])
        f("ora_stage_free", None, [vp])
        f("ora_stage_set", None, [vp, i32, u32])
        f("ora_stage_process", C.c_long, [vp, fp, sz, rp])
        f("ora_stage_spectrum", None, [vp, rp])
        f("ora_stage_gain", rt, [vp])
        f("ora_stage_count", u32, [vp])
        f("ora_stage_pending", i32, [vp])
        f("ora_window", i32, [i32, i32, rp, rp, rp, C.POINTER(i32)])
        f("ora_fft_forward", i32, [i32, rp])
        f("ora_detrend_apply", i32, [i32, i32, i32, rp, rp])
        f("ora_hbf_dec8", C.c_long, [rp, sz, rp])
    L.ora_hbf_response_length.restype, L.ora_hbf_response_length.argtypes = i32, [i32]
    L.ora_frequencies.restype = C.c_long
    L.ora_frequencies.argtypes = [C.POINTER(Break), i32, fp]
    L.ora_var_eval.restype = C.c_float
    L.ora_var_eval.argtypes = [i32, i32, C.c_float, sz, fp, fp, sz, C.c_float]
    L.ora_trace_plot.restype = C.c_long
    L.ora_trace_plot.argtypes = [fp, fp, sz, C.c_float, i32, C.c_float, C.c_float, fp, dp]
    L.ora_frame_decode.restype = i32
    L.ora_frame_decode.argtypes = [C.POINTER(C.c_uint8), sz, fp, fp, fp, fp] + [C.POINTER(C.c_uint32)] * 5
    L.ora_adcdac_decode.restype = i32
    L.ora_adcdac_decode.argtypes = [C.POINTER(C.c_uint8), sz, fp, fp, fp, fp,
                                    C.POINTER(u32), C.POINTER(u32)]


DETREND = {"none": 0, "midpoint": 1, "span": 2, "mean": 3}
U32_MAX = 0xFFFFFFFF


def _dt(prec):
    return (np.float32, C.c_float, "_f32") if prec == "f32" else (np.float64, C.c_double, "_f64")


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class PsdCascade:
    """Oracle PsdCascade<N> (src/psd.rs:399-544). prec: 'f32' mirrors the reference, 'f64' truth."""

    def __init__(self, n, prec="f32", window="hann"):
        """window: "hann" / "rect", or a caller-built Window<N> as (win[n] f32, power, nenbw, overlap)
        (src/psd.rs:12-20: pub struct, pub fields)."""
        self.n, self.prec = n, prec
        self.np_t, self.c_t, self.sfx = _dt(prec)
        self.L = lib()
        if isinstance(window, str):
            self.h = getattr(self.L, "ora_cascade_new" + self.sfx)(n, 1 if window == "hann" else 0)
        else:
            w, power, nenbw, overlap = window
            w = np.ascontiguousarray(w, dtype=np.float32)
            assert w.size == n
            self.h = getattr(self.L, "ora_cascade_new_window" + self.sfx)(n, _ptr(w, C.c_float), power, nenbw, overlap)
        if not self.h:
            raise ValueError("oracle: bad N")

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.L, "ora_cascade_free" + self.sfx)(self.h)
            self.h = None

    def _f(self, name):
        return getattr(self.L, name + self.sfx)

    def set_detrend(self, d):
        d = DETREND[d] if isinstance(d, str) else d
        if self._f("ora_cascade_set_detrend")(self.h, d):
            raise ValueError("unimplemented detrend")  # src/psd.rs:110

    def set_avg(self, limit=U32_MAX, count=U32_MAX):
        self._f("ora_cascade_set_avg")(self.h, limit, count)

    def set_fast_fft(self):
        """bench.py's cpu_baseline: the radix-4 Stockham FFT whose loops gcc vectorises (same DFT, other rounding)."""
        if self._f("ora_cascade_set_fast_fft")(self.h):
            raise ValueError("no fast plan for this N")

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        if self._f("ora_cascade_process")(self.h, _ptr(x, C.c_float), x.size):
            raise RuntimeError("oracle process failed (reference would panic)")

    @property
    def num_stages(self):
        return self._f("ora_cascade_num_stages")(self.h)

    @property
    def ref_would_panic(self):
        """True once a feed pattern occurred on which the reference's [f32; N] ping-pong buffers
        overflow (src/psd.rs:253 with :457): a short first call, then a full 8N chunk."""
        return bool(self._f("ora_cascade_ref_would_panic")(self.h))

    def stage_info(self, k):
        c, a, p, q = C.c_uint32(), C.c_uint32(), C.c_uint64(), C.c_uint64()
        if self._f("ora_cascade_stage_info")(self.h, k, c, a, p, q):
            raise IndexError(k)
        return {"count": c.value, "avg": a.value, "pending": p.value, "processed": q.value}

    def stage_spectrum(self, k):
        out = np.empty(self.n // 2 + 1, dtype=self.np_t)
        if self._f("ora_cascade_stage_spectrum")(self.h, k, _ptr(out, self.c_t)):
            raise IndexError(k)
        return out

    def stage_buf(self, k):
        out = np.empty(self.stage_info(k)["pending"], dtype=self.np_t)
        self._f("ora_cascade_stage_buf")(self.h, k, _ptr(out, self.c_t))
        return out

    def stage_gain(self, k):
        return float(self._f("ora_cascade_stage_gain")(self.h, k))

    def psd(self, keep_overlap=False, min_count=1, keep_transition_band=False):
        ns = self.num_stages
        out = np.empty(max(1, ns * (self.n // 2 + 1)), dtype=self.np_t)
        br = (Break * max(1, ns))()
        m = self._f("ora_cascade_psd")(self.h, int(keep_overlap), min_count,
                                       int(keep_transition_band), _ptr(out, self.c_t), br)
        return out[:m].copy(), [br[i].as_dict() for i in range(ns)], br

    def frequencies(self, br):
        out = np.empty(max(1, len(br) * (self.n // 2 + 1)), dtype=np.float32)
        m = self.L.ora_frequencies(br, len(br), _ptr(out, C.c_float))
        return out[:m].copy()


class Psd:
    """Oracle single stage Psd<N> (src/psd.rs:122-288)."""

    def __init__(self, n, prec="f32", window="hann", detrend="none", avg=U32_MAX):
        self.n = n
        self.np_t, self.c_t, self.sfx = _dt(prec)
        self.L = lib()
        if isinstance(window, str):
            self.h = getattr(self.L, "ora_stage_new" + self.sfx)(n, 1 if window == "hann" else 0)
        else:  # Psd::new(fft, win) with a caller-built Window<N>: (win[n] f32, power, nenbw, overlap)
            w, power, nenbw, overlap = window
            w = np.ascontiguousarray(w, dtype=np.float32)
            assert w.size == n
            self.h = getattr(self.L, "ora_stage_new_window" + self.sfx)(n, _ptr(w, C.c_float), power, nenbw, overlap)
        if not self.h:
            raise ValueError("oracle: bad N")
        getattr(self.L, "ora_stage_set" + self.sfx)(self.h, DETREND[detrend], avg)

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.L, "ora_stage_free" + self.sfx)(self.h)
            self.h = None

    def process(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.empty(x.size // 8 + self.n // 8 + 8, dtype=self.np_t)
        m = getattr(self.L, "ora_stage_process" + self.sfx)(self.h, _ptr(x, C.c_float), x.size,
                                                            _ptr(y, self.c_t))
        if m < 0:
            raise RuntimeError("oracle stage process failed (reference would panic)")
        return y[:m].copy()

    def spectrum(self):
        out = np.empty(self.n // 2 + 1, dtype=self.np_t)
        getattr(self.L, "ora_stage_spectrum" + self.sfx)(self.h, _ptr(out, self.c_t))
        return out

    def gain(self):
        return float(getattr(self.L, "ora_stage_gain" + self.sfx)(self.h))

    def count(self):
        return getattr(self.L, "ora_stage_count" + self.sfx)(self.h)

    def pending(self):
        return getattr(self.L, "ora_stage_pending" + self.sfx)(self.h)


def window(n, kind="hann", prec="f32"):
    np_t, c_t, sfx = _dt(prec)
    w = np.empty(n, dtype=np_t)
    p, e, ov = c_t(), c_t(), C.c_int()
    getattr(lib(), "ora_window" + sfx)(n, 1 if kind == "hann" else 0, _ptr(w, c_t), p, e, ov)
    return w, p.value, e.value, ov.value


def fft_forward(c, prec="f32"):
    """c: complex array; returns forward unnormalised DFT."""
    np_t, c_t, sfx = _dt(prec)
    n = c.size
    buf = np.empty(2 * n, dtype=np_t)
    buf[0::2], buf[1::2] = c.real, c.imag
    if getattr(lib(), "ora_fft_forward" + sfx)(n, _ptr(buf, c_t)):
        raise ValueError("N must be a power of two")
    return buf[0::2] + 1j * buf[1::2]


def detrend_apply(x, detrend="none", window_kind="hann", prec="f32"):
    np_t, c_t, sfx = _dt(prec)
    x = np.ascontiguousarray(x, dtype=np_t)
    c = np.empty(2 * x.size, dtype=np_t)
    if getattr(lib(), "ora_detrend_apply" + sfx)(x.size, 1 if window_kind == "hann" else 0,
                                                 DETREND[detrend], _ptr(x, c_t), _ptr(c, c_t)):
        raise ValueError("unimplemented detrend")
    return c[0::2] + 1j * c[1::2]


def hbf_dec8(x, prec="f32"):
    np_t, c_t, sfx = _dt(prec)
    x = np.ascontiguousarray(x, dtype=np_t)
    y = np.empty(x.size // 8 + 1, dtype=np_t)
    m = getattr(lib(), "ora_hbf_dec8" + sfx)(_ptr(x, c_t), x.size, _ptr(y, c_t))
    return y[:m].copy()


def hbf_response_length(depth=3):
    return lib().ora_hbf_response_length(depth)


def var_eval(phase_psd, frequencies, tau, x_exp=-2, sinx_exp=4, clip=3.4028234663852886e38, dc_cut=2):
    p = np.ascontiguousarray(phase_psd, dtype=np.float32)
    f = np.ascontiguousarray(frequencies, dtype=np.float32)
    return float(lib().ora_var_eval(x_exp, sinx_exp, clip, dc_cut, _ptr(p, C.c_float),
                                    _ptr(f, C.c_float), p.size, tau))


def trace_plot(psd, frequencies, fs=1.0, integrate=False, integral_start=0.0, integral_end=float("inf")):
    """Trace::plot (src/bin/psd.rs:125-157): (rms, plot points)."""
    p = np.ascontiguousarray(psd, dtype=np.float32)
    f = np.ascontiguousarray(frequencies, dtype=np.float32)
    xy = np.empty((max(1, p.size), 2), dtype=np.float64)
    rms = C.c_float()
    m = lib().ora_trace_plot(_ptr(p, C.c_float), _ptr(f, C.c_float), p.size, fs, int(integrate), integral_start,
                             integral_end, C.byref(rms), _ptr(xy, C.c_double))
    return rms.value, xy[:m].copy()


def adcdac_decode(frame):
    """frame: bytes -> (status, seq, batches, [ADC0, ADC1, DAC0, DAC1])"""
    b = np.frombuffer(bytes(frame), dtype=np.uint8)
    nb = max(0, (b.size - 8) // 64)
    tr = [np.zeros(8 * nb + 8, dtype=np.float32) for _ in range(4)]
    seq, bat = C.c_uint32(), C.c_uint32()
    st = lib().ora_adcdac_decode(_ptr(b, C.c_uint8) if b.size else None, b.size,
                                 *[_ptr(t, C.c_float) for t in tr], seq, bat)
    n = 8 * bat.value if st == 0 else 0
    return st, seq.value, bat.value, [t[:n].copy() for t in tr]


# trace labels of Payload::traces (src/de/data.rs:38-80, 98-138, 155, 181-207), by Format id (src/de/mod.rs:12-17)
TRACE_NAMES = {1: ("ADC0", "ADC1", "DAC0", "DAC1"), 2: ("AR", "AP", "BI", "BQ"), 3: ("T00", "T20", "I0", "I1"),
               4: ("phase (rad)", "frequency (kHz)", "amplitude (V/G10)")}


def frame_decode(frame):
    """Frame::from_bytes + Payload::traces for any of the four formats.
    frame: bytes -> (status, format id, seq, batches, [(name, f32 array), ...])"""
    b = np.frombuffer(bytes(frame), dtype=np.uint8)
    cap = max(0, b.size - 8) // 8 + 8
    tr = [np.zeros(cap, dtype=np.float32) for _ in range(4)]
    fmt, ntr, ns, seq, bat = (C.c_uint32() for _ in range(5))
    st = lib().ora_frame_decode(_ptr(b, C.c_uint8) if b.size else None, b.size, *[_ptr(t, C.c_float) for t in tr],
                                fmt, ntr, ns, seq, bat)
    out = []
    if st == 0:
        out = [(TRACE_NAMES[fmt.value][i], tr[i][:ns.value].copy()) for i in range(ntr.value)]
    return st, fmt.value, seq.value, bat.value, out
