"""MI355X-native cascaded PSD estimator -- Python mirror of the reference surface.

Thin ctypes layer over the C ABI in include/psdcascade.h (libpsdcascade.so, HIP
kernels for gfx950).  Class and method names follow quartiq/stabilizer-stream
src/psd.rs so that tests read like the reference's own:

    PsdCascade(n)            PsdCascade::<N>::default()        src/psd.rs:408-423
      .set_detrend(Detrend)  PsdCascade::set_detrend           src/psd.rs:438-443
      .set_avg(AvgOpts)      PsdCascade::set_avg               src/psd.rs:431-436
      .process(x)            PsdCascade::process               src/psd.rs:456-468
      .psd(MergeOpts)        PsdCascade::psd -> (psd, breaks)  src/psd.rs:479-543
    Break.frequencies(b)     Break::frequencies                src/psd.rs:315-327

There is no CPU fallback: constructing a PsdCascade without a HIP device raises.
The directory name carries a hyphen; import it through `__graft_entry__.load_package()`
or tests/conftest.py (module name `stabilizer_stream_amd`).
"""
import ctypes as C
import enum
import importlib.util
import os
import subprocess
import sys
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $PSDC_LIB: another build of the SAME library (tools/gpu.sh variants: timing-only ablation builds) -- never a fallback
LIB_PATH = os.environ.get("PSDC_LIB") or os.path.join(_HERE, "libpsdcascade.so")
U32_MAX = 0xFFFFFFFF

PSDC_OK = 0
ERR_ARG, ERR_DEVICE, ERR_NOMEM, ERR_UNIMPLEMENTED = -1, -2, -3, -4
ERR_FRAME_HEADER, ERR_FRAME_FORMAT, ERR_FRAME_SIZE, ERR_CAPACITY = -5, -6, -7, -8
OPT_QUANTUM, OPT_PROFILE, OPT_COALESCE, OPT_MIN_PAIRS, OPT_EAGER, OPT_MERGE = 1, 2, 3, 4, 5, 6


class PsdError(RuntimeError):
    """A contract violation the reference would panic on, or a device error."""

    def __init__(self, code, msg):
        super().__init__(f"psdcascade error {code}: {msg}")
        self.code = code


class FrameError(PsdError):
    """de::Error (src/de/mod.rs:19-27)."""


class Detrend(enum.IntEnum):
    """src/psd.rs:59-72"""
    NONE = 0
    MIDPOINT = 1
    SPAN = 2
    MEAN = 3
    LINEAR = 4


class Format(enum.IntEnum):
    """Stream payload formats (src/de/mod.rs:9-17)"""
    ADC_DAC = 1
    FLS = 2
    THERMOSTAT_EEM = 3
    MPLL = 4


# labels of Payload::traces by format (src/de/data.rs:38-80, 98-138, 155, 181-207); trace i feeds channel i (src/bin/psd.rs:174-182)
TRACE_NAMES = {
    Format.ADC_DAC: ("ADC0", "ADC1", "DAC0", "DAC1"),
    Format.FLS: ("AR", "AP", "BI", "BQ"),
    Format.THERMOSTAT_EEM: ("T00", "T20", "I0", "I1"),
    Format.MPLL: ("phase (rad)", "frequency (kHz)", "amplitude (V/G10)"),
}
BATCH_BYTES = {Format.ADC_DAC: 64, Format.FLS: 56, Format.THERMOSTAT_EEM: 80, Format.MPLL: 24}  # src/de/data.rs:13, 86, 144, 168


class Window(enum.IntEnum):
    """Window::rectangular / Window::hann (src/psd.rs:24-55) by kind"""
    RECTANGULAR = 0
    HANN = 1
    CUSTOM = 2


@dataclass(frozen=True, eq=False)
class WindowTable:
    """`Window<N>` (src/psd.rs:12-20): a public struct with public fields, so a caller may build any.

        WindowTable.hann(n) / .rectangular(n)        Window::hann() / Window::rectangular()   src/psd.rs:24-55
        WindowTable(win, power, nenbw, overlap)      Window { win, power, nenbw, overlap }
    """
    win: np.ndarray  # [n] f32
    power: float     # src/psd.rs:15
    nenbw: float     # src/psd.rs:17
    overlap: int     # src/psd.rs:19

    @staticmethod
    def _kind(n, kind):
        w = np.empty(n, dtype=np.float32)
        p, e, ov = C.c_float(), C.c_float(), C.c_size_t()
        rc = lib().psdc_window_table(n, int(kind), _fptr(w), C.byref(p), C.byref(e), C.byref(ov))
        if rc < 0:
            _raise(rc)
        return WindowTable(w, p.value, e.value, ov.value)

    @staticmethod
    def hann(n):
        return WindowTable._kind(n, Window.HANN)

    @staticmethod
    def rectangular(n):
        return WindowTable._kind(n, Window.RECTANGULAR)

    def as_tuple(self):
        """(win, power, nenbw, overlap)"""
        return (self.win, self.power, self.nenbw, self.overlap)


@dataclass(frozen=True)
class MergeOpts:
    """src/psd.rs:339-358"""
    keep_overlap: bool = False
    min_count: int = 1
    keep_transition_band: bool = False


@dataclass(frozen=True)
class AvgOpts:
    """src/psd.rs:360-376"""
    limit: int = U32_MAX
    count: int = U32_MAX


class _CBreak(C.Structure):
    _fields_ = [("start", C.c_uint64), ("include", C.c_uint32), ("count", C.c_uint32),
                ("avg", C.c_uint32), ("_pad", C.c_uint32),
                ("bins_start", C.c_uint64), ("bins_end", C.c_uint64),
                ("fft_size", C.c_uint64), ("decimation", C.c_uint64),
                ("pending", C.c_uint64), ("processed", C.c_uint64)]


class _CStageStat(C.Structure):
    _fields_ = [("count", C.c_uint32), ("avg", C.c_uint32),
                ("pending", C.c_uint64), ("processed", C.c_uint64)]


class _CLoss(C.Structure):
    _fields_ = [("received", C.c_uint64), ("dropped", C.c_uint64),
                ("next_seq", C.c_uint32), ("have_seq", C.c_uint32)]


class _CProfile(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("kernel_ms", C.c_double),
                ("samples", C.c_uint64), ("stage0_samples", C.c_uint64)]


@dataclass(frozen=True)
class Break:
    """Stage break information (src/psd.rs:290-311)."""
    start: int
    include: bool
    count: int
    avg: int
    bins: range
    fft_size: int
    decimation: int
    pending: int
    processed: int

    def effective_fft_size(self):  # src/psd.rs:329-331
        return self.fft_size * self.decimation

    def rbw(self):  # src/psd.rs:334-336
        return float(np.float32(1.0) / np.float32(self.effective_fft_size()))

    def _c(self):
        return _CBreak(self.start, int(self.include), self.count, self.avg, 0, self.bins.start,
                       self.bins.stop, self.fft_size, self.decimation, self.pending, self.processed)

    @staticmethod
    def _from_c(b):
        return Break(int(b.start), bool(b.include), int(b.count), int(b.avg),
                     range(int(b.bins_start), int(b.bins_end)), int(b.fft_size),
                     int(b.decimation), int(b.pending), int(b.processed))

    @staticmethod
    def frequencies(breaks):
        """Break::frequencies (src/psd.rs:315-327)."""
        L = lib()
        arr = (_CBreak * max(1, len(breaks)))(*[b._c() for b in breaks])
        n = L.psdc_frequencies(arr, len(breaks), None, 0)
        out = np.empty(n, dtype=np.float32)
        L.psdc_frequencies(arr, len(breaks), out.ctypes.data_as(C.POINTER(C.c_float)), n)
        return out


def build(force=False, verbose=False):
    """Compile libpsdcascade.so for gfx950 with hipcc (csrc/Makefile)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc] + (["-B"] if force else [])
    subprocess.run(cmd, check=True, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def _one_hip_runtime():
    """A process must hold ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64.so; if
    libpsdcascade.so were loaded first it would bind /opt/rocm's copy, torch would later bring its
    own, and whichever initialises second finds "no ROCm-capable device".  So when torch is
    installed but not imported yet, load the runtime it will use (same SONAME) before our library;
    no torch import is forced on callers that never use it."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    """Load the C-ABI library.  Fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() "
                          "(there is no Python/CPU fallback for the PSD kernels)")
    _one_hip_runtime()
    L = C.CDLL(LIB_PATH)
    H, u32, u64, i32, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_size_t
    fp = C.POINTER(C.c_float)

    def f(name, res, args):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args

    f("psdc_abi_version", i32, [])
    f("psdc_last_error", C.c_char_p, [H])
    f("psdc_create", H, [u32, i32, u32, i32])
    f("psdc_create_window", H, [u32, fp, C.c_float, C.c_float, sz, u32, i32])
    f("psdc_window_get", i32, [H, C.POINTER(i32), fp, fp, C.POINTER(sz), fp])
    f("psdc_window_table", i32, [u32, i32, fp, fp, fp, C.POINTER(sz)])
    f("psdc_stage_create_window", H, [u32, fp, C.c_float, C.c_float, sz, i32])
    f("psdc_stitch_window", i32, [u32, C.c_float, C.c_float, sz, u32, C.POINTER(u64), C.POINTER(u32), C.POINTER(u64), fp,
                                  i32, u32, i32, fp, sz, C.POINTER(sz), C.POINTER(_CBreak), sz, C.POINTER(sz)])
    f("psdc_readout_bytes", sz, [u32, u32])
    f("psdc_pack_readout", i32, [H, C.c_void_p, sz, C.POINTER(sz)])
    f("psdc_pack_init", i32, [C.c_void_p, sz, u32, C.c_float, C.c_float, sz, u32])
    f("psdc_pack_channel", i32, [C.c_void_p, sz, u32, u32, C.POINTER(u64), C.POINTER(u32), C.POINTER(u64), fp])
    f("psdc_pack_pad", i32, [C.c_void_p, sz, C.c_void_p, sz, u32])
    f("psdc_unpack_info", i32, [C.c_void_p, sz, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)])
    f("psdc_unpack_stitch", i32, [C.c_void_p, sz, u32, i32, u32, i32, fp, sz, C.POINTER(sz), C.POINTER(_CBreak), sz,
                                  C.POINTER(sz)])
    f("psdc_destroy", None, [H])
    f("psdc_clone", H, [H])
    f("psdc_reset", i32, [H])
    f("psdc_configure", i32, [H, i32, C.c_int64])
    f("psdc_set_detrend", i32, [H, i32])
    f("psdc_set_avg", i32, [H, u32, u32])
    f("psdc_process", i32, [H, u32, fp, sz])
    f("psdc_process_device", i32, [H, u32, C.c_void_p, sz])
    f("psdc_process_device_after", i32, [H, u32, C.c_void_p, sz, C.c_void_p])
    f("psdc_record_consumed", i32, [H, C.c_void_p])
    f("psdc_process_adcdac_frames", i32, [H, C.c_void_p, sz, sz, C.POINTER(sz)])
    f("psdc_process_frames", i32, [H, C.c_void_p, sz, sz, C.POINTER(sz)])
    f("psdc_process_frames_device", i32, [H, C.c_void_p, sz, sz, C.POINTER(sz)])
    f("psdc_process_adcdac_frames_device", i32, [H, C.c_void_p, sz, sz, C.POINTER(sz)])
    f("psdc_loss_read", i32, [H, C.POINTER(_CLoss), i32])
    f("psdc_flush", i32, [H])
    f("psdc_sync", i32, [H])
    f("psdc_num_stages", i32, [H, u32])
    f("psdc_stage_info", i32, [H, u32, u32, C.POINTER(_CStageStat)])
    f("psdc_stage_spectrum", i32, [H, u32, u32, fp])
    f("psdc_stage_gain", i32, [H, u32, u32, fp])
    f("psdc_stage_buf", i32, [H, u32, u32, fp, sz, C.POINTER(sz)])
    f("psdc_read_channel", i32, [H, u32, u32, C.POINTER(u32), C.POINTER(_CStageStat), fp])
    f("psdc_psd", i32, [H, u32, i32, u32, i32, fp, sz, C.POINTER(sz), C.POINTER(_CBreak), sz,
                        C.POINTER(sz)])
    f("psdc_rbw", C.c_float, [H])
    f("psdc_frequencies", sz, [C.POINTER(_CBreak), sz, fp, sz])
    f("psdc_hbf_response_length", i32, [i32])
    f("psdc_stitch", i32, [u32, i32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), fp, i32,
                           u32, i32, fp, sz, C.POINTER(sz), C.POINTER(_CBreak), sz, C.POINTER(sz)])
    f("psdc_plan_counts", i32, [u32, i32, u64, u32, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)])
    f("psdc_var_eval", C.c_float, [i32, i32, C.c_float, sz, fp, fp, sz, C.c_float])
    f("psdc_hbf_dec8", i32, [i32, fp, sz, fp])
    f("psdc_fill_noise_device", i32, [i32, C.c_void_p, sz, u64, u64])
    f("psdc_profile_read", i32, [H, C.POINTER(_CProfile), i32])
    f("psdc_trace_plot", i32, [fp, fp, sz, C.c_float, i32, C.c_float, C.c_float, fp, C.POINTER(C.c_double), sz,
                               C.POINTER(sz)])
    f("psdc_stage_create", H, [u32, i32, i32])
    f("psdc_stage_destroy", None, [H])
    f("psdc_stage_clone", H, [H])
    f("psdc_stage_set_avg", i32, [H, u32])
    f("psdc_stage_set_detrend", i32, [H, i32])
    f("psdc_stage_process", i32, [H, fp, sz, fp, sz, C.POINTER(sz)])
    f("psdc_stage_process_device", i32, [H, C.c_void_p, sz, C.c_void_p, sz, C.POINTER(sz)])
    f("psdc_stage_get_spectrum", i32, [H, fp])
    f("psdc_stage_get_count", i32, [H, C.POINTER(u32)])
    f("psdc_stage_get_gain", i32, [H, fp])
    f("psdc_stage_get_buf", i32, [H, fp, sz, C.POINTER(sz)])
    f("psdc_stage_last_error", C.c_char_p, [H])
    _lib = L
    return L


EXPORTS = [
    "psdc_abi_version", "psdc_last_error", "psdc_create", "psdc_destroy", "psdc_clone", "psdc_reset",
    "psdc_configure", "psdc_set_detrend", "psdc_set_avg", "psdc_process", "psdc_process_device",
    "psdc_process_adcdac_frames", "psdc_process_frames", "psdc_process_frames_device", "psdc_loss_read", "psdc_flush", "psdc_sync", "psdc_num_stages", "psdc_stage_info",
    "psdc_stage_spectrum", "psdc_stage_gain", "psdc_stage_buf", "psdc_read_channel", "psdc_psd", "psdc_rbw",
    "psdc_frequencies", "psdc_hbf_response_length", "psdc_stitch", "psdc_plan_counts",
    "psdc_var_eval", "psdc_hbf_dec8", "psdc_fill_noise_device", "psdc_profile_read",
    "psdc_process_device_after", "psdc_record_consumed", "psdc_trace_plot", "psdc_process_adcdac_frames_device",
    "psdc_stage_create", "psdc_stage_destroy", "psdc_stage_clone", "psdc_stage_set_avg", "psdc_stage_set_detrend",
    "psdc_stage_process", "psdc_stage_process_device", "psdc_stage_get_spectrum", "psdc_stage_get_count",
    "psdc_stage_get_gain", "psdc_stage_get_buf", "psdc_stage_last_error",
    "psdc_create_window", "psdc_window_get", "psdc_window_table", "psdc_stage_create_window", "psdc_stitch_window",
    "psdc_readout_bytes", "psdc_pack_readout", "psdc_unpack_info", "psdc_unpack_stitch",
    "psdc_pack_init", "psdc_pack_channel", "psdc_pack_pad",
]


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _raise(code, h=None):
    msg = lib().psdc_last_error(h)
    msg = msg.decode() if msg else ""
    cls = FrameError if code in (ERR_FRAME_HEADER, ERR_FRAME_FORMAT, ERR_FRAME_SIZE) else PsdError
    raise cls(code, msg)


class PsdCascadeBank:
    """`n_channels` independent PsdCascade<N> batched on one GPU (one handle of the C ABI)."""

    def __init__(self, n, n_channels=1, window=Window.HANN, device=0, _handle=None):
        """window: a Window kind, or a caller-built WindowTable (any Window<N>, src/psd.rs:12-20).
        device: HIP device index, or -1 = the index in $PSDC_DEVICE (0 when unset)."""
        self.n, self.n_channels, self.window, self.device = n, n_channels, window, device
        self._L = lib()
        if _handle is not None:
            self._h = _handle
        elif isinstance(window, WindowTable):
            w = np.ascontiguousarray(window.win, dtype=np.float32)
            if w.size != n:
                raise PsdError(ERR_ARG, "window table length != n")
            self._h = self._L.psdc_create_window(n, _fptr(w), window.power, window.nenbw, window.overlap, n_channels, device)
        else:
            self._h = self._L.psdc_create(n, int(window), n_channels, device)
        if not self._h:
            _raise(ERR_DEVICE)

    def window_get(self):
        """(kind, WindowTable) as the library holds it: a table equal to Window::hann() is recognised as HANN."""
        k, p, e, ov = C.c_int(), C.c_float(), C.c_float(), C.c_size_t()
        w = np.empty(self.n, dtype=np.float32)
        self._ck(self._L.psdc_window_get(self._h, C.byref(k), C.byref(p), C.byref(e), C.byref(ov), _fptr(w)))
        return Window(k.value), WindowTable(w, p.value, e.value, ov.value)

    def pack_readout(self):
        """psdc_pack_readout: the fixed-size byte record of this handle's raw accumulators and counters."""
        need = self._L.psdc_readout_bytes(self.n, self.n_channels)
        buf = np.empty(need, dtype=np.uint8)
        ln = C.c_size_t()
        self._ck(self._L.psdc_pack_readout(self._h, buf.ctypes.data_as(C.c_void_p), buf.size, C.byref(ln)))
        assert ln.value == need
        return buf

    def close(self):
        if getattr(self, "_h", None):
            self._L.psdc_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc):
        if rc < 0:
            _raise(rc, self._h)
        return rc

    def clone(self):
        h = self._L.psdc_clone(self._h)
        if not h:
            _raise(ERR_DEVICE)
        return PsdCascadeBank(self.n, self.n_channels, self.window, self.device, _handle=h)

    def reset(self):
        self._ck(self._L.psdc_reset(self._h))

    def configure(self, quantum=None, profile=None, coalesce=None, min_pairs=None, eager=None, merge=None):
        if merge is not None:
            self._ck(self._L.psdc_configure(self._h, OPT_MERGE, int(bool(merge))))
        if eager is not None:
            self._ck(self._L.psdc_configure(self._h, OPT_EAGER, int(bool(eager))))
        if min_pairs is not None:
            self._ck(self._L.psdc_configure(self._h, OPT_MIN_PAIRS, int(min_pairs)))
        if quantum is not None:
            self._ck(self._L.psdc_configure(self._h, OPT_QUANTUM, int(quantum)))
        if profile is not None:
            self._ck(self._L.psdc_configure(self._h, OPT_PROFILE, int(bool(profile))))
        if coalesce is not None:
            self._ck(self._L.psdc_configure(self._h, OPT_COALESCE, int(coalesce)))

    def rbw(self):
        return float(self._L.psdc_rbw(self._h))

    def set_detrend(self, d):
        self._ck(self._L.psdc_set_detrend(self._h, int(d)))

    def set_avg(self, avg):
        self._ck(self._L.psdc_set_avg(self._h, avg.limit, avg.count))

    def process(self, channel, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        self._ck(self._L.psdc_process(self._h, channel, _fptr(x), x.size))

    def process_device(self, channel, ptr, length, after=None):
        """ptr: device address (e.g. torch tensor .data_ptr()) of `length` f32 samples.  The producer of the
        samples must have completed -- or pass `after`: a hipEvent_t handle (torch.cuda.Event.cuda_event)
        recorded behind the producing work; the library's stream then waits for it on the device."""
        if after is None:
            self._ck(self._L.psdc_process_device(self._h, channel, C.c_void_p(ptr), length))
        else:
            self._ck(self._L.psdc_process_device_after(self._h, channel, C.c_void_p(ptr), length, C.c_void_p(after)))

    def record_consumed(self, event):
        """Record the hipEvent_t handle `event` behind the last read of every span handed over so far."""
        self._ck(self._L.psdc_record_consumed(self._h, C.c_void_p(event)))

    def process_adcdac_frames(self, data, frame_size):
        """data: bytes-like holding whole frames; returns the number of frames ingested."""
        buf = np.frombuffer(data, dtype=np.uint8)
        n_frames = buf.size // frame_size
        ok = C.c_size_t(0)
        rc = self._L.psdc_process_adcdac_frames(self._h, buf.ctypes.data_as(C.c_void_p), frame_size,
                                                n_frames, C.byref(ok))
        if rc < 0:
            _raise(rc, self._h)
        return ok.value

    def process_frames(self, data, frame_size):
        """Frames of any of the four payload formats (src/de/mod.rs:12-17), each frame's own header naming its format: trace i of
        every frame goes to channel i.  data: bytes-like holding whole frames; returns the number of frames ingested."""
        buf = np.frombuffer(data, dtype=np.uint8)
        ok = C.c_size_t(0)
        rc = self._L.psdc_process_frames(self._h, buf.ctypes.data_as(C.c_void_p), frame_size, buf.size // frame_size, C.byref(ok))
        if rc < 0:
            _raise(rc, self._h)
        return ok.value

    def process_frames_device(self, ptr, frame_size, n_frames):
        """process_frames for frames resident in device memory at address `ptr`; returns the number of frames ingested."""
        ok = C.c_size_t(0)
        rc = self._L.psdc_process_frames_device(self._h, C.c_void_p(ptr), frame_size, n_frames, C.byref(ok))
        if rc < 0:
            _raise(rc, self._h)
        return ok.value

    def process_adcdac_frames_device(self, ptr, frame_size, n_frames):
        """Frames resident in device memory at address `ptr`; returns the number of frames ingested."""
        ok = C.c_size_t(0)
        rc = self._L.psdc_process_adcdac_frames_device(self._h, C.c_void_p(ptr), frame_size, n_frames, C.byref(ok))
        if rc < 0:
            _raise(rc, self._h)
        return ok.value

    def loss(self, reset=False):
        """Loss counters (src/loss.rs): batches received / dropped over the ingested frames."""
        l = _CLoss()
        self._ck(self._L.psdc_loss_read(self._h, C.byref(l), int(reset)))
        return {"received": l.received, "dropped": l.dropped}

    def flush(self):
        self._ck(self._L.psdc_flush(self._h))

    def sync(self):
        self._ck(self._L.psdc_sync(self._h))

    def num_stages(self, channel=0):
        return self._ck(self._L.psdc_num_stages(self._h, channel))

    def stage_info(self, channel, stage):
        st = _CStageStat()
        self._ck(self._L.psdc_stage_info(self._h, channel, stage, C.byref(st)))
        return {"count": st.count, "avg": st.avg, "pending": st.pending, "processed": st.processed}

    def stage_spectrum(self, channel, stage):
        out = np.empty(self.n // 2 + 1, dtype=np.float32)
        self._ck(self._L.psdc_stage_spectrum(self._h, channel, stage, _fptr(out)))
        return out

    def stage_gain(self, channel, stage):
        g = C.c_float()
        self._ck(self._L.psdc_stage_gain(self._h, channel, stage, C.byref(g)))
        return g.value

    def stage_buf(self, channel, stage):
        ln = C.c_size_t()
        self._ck(self._L.psdc_stage_buf(self._h, channel, stage, None, 0, C.byref(ln)))
        out = np.empty(ln.value, dtype=np.float32)
        self._ck(self._L.psdc_stage_buf(self._h, channel, stage, _fptr(out), out.size, C.byref(ln)))
        return out

    def read_channel(self, channel=0):
        """All stages of one channel in one call: ([{count, avg, pending, processed}], spectra[ns, n/2+1])."""
        ns = C.c_uint32()
        self._ck(self._L.psdc_read_channel(self._h, channel, 0, C.byref(ns), None, None))
        st = (_CStageStat * max(1, ns.value))()
        sp = np.empty((ns.value, self.n // 2 + 1), dtype=np.float32)
        self._ck(self._L.psdc_read_channel(self._h, channel, ns.value, C.byref(ns), st, _fptr(sp) if ns.value else None))
        info = [{"count": st[k].count, "avg": st[k].avg, "pending": st[k].pending, "processed": st[k].processed}
                for k in range(ns.value)]
        return info, sp

    def psd(self, channel=0, opts=MergeOpts()):
        ns = self.num_stages(channel)
        out = np.empty(max(1, ns * (self.n // 2 + 1)), dtype=np.float32)
        br = (_CBreak * max(1, ns))()
        plen, nb = C.c_size_t(), C.c_size_t()
        self._ck(self._L.psdc_psd(self._h, channel, int(opts.keep_overlap), opts.min_count,
                                  int(opts.keep_transition_band), _fptr(out), out.size, C.byref(plen),
                                  br, ns, C.byref(nb)))
        return out[:plen.value].copy(), [Break._from_c(br[i]) for i in range(nb.value)]

    def profile_read(self, reset=False):
        p = _CProfile()
        self._ck(self._L.psdc_profile_read(self._h, C.byref(p), int(reset)))
        return {"launches": p.launches, "kernel_ms": p.kernel_ms, "samples": p.samples,
                "stage0_samples": p.stage0_samples}


class PsdCascade:
    """Online cascaded PSD estimator, one trace (src/psd.rs:399-544)."""

    def __init__(self, n, window=Window.HANN, device=0, _bank=None):
        self.n = n
        self._b = _bank if _bank is not None else PsdCascadeBank(n, 1, window, device)

    def clone(self):
        return PsdCascade(self.n, _bank=self._b.clone())

    def rbw(self):
        return self._b.rbw()

    def set_avg(self, avg):
        self._b.set_avg(avg)

    def set_detrend(self, d):
        self._b.set_detrend(d)

    def process(self, x):
        self._b.process(0, x)

    def process_device(self, ptr, length, after=None):
        self._b.process_device(0, ptr, length, after)

    def psd(self, opts=MergeOpts()):
        return self._b.psd(0, opts)

    # PsdStage accessors of stage i (src/psd.rs:271-287)
    def num_stages(self):
        return self._b.num_stages(0)

    def stage_spectrum(self, i):
        return self._b.stage_spectrum(0, i)

    def stage_gain(self, i):
        return self._b.stage_gain(0, i)

    def stage_count(self, i):
        return self._b.stage_info(0, i)["count"]

    def stage_buf(self, i):
        return self._b.stage_buf(0, i)

    def stage_info(self, i):
        return self._b.stage_info(0, i)

    def configure(self, **kw):
        self._b.configure(**kw)

    def sync(self):
        self._b.sync()

    def close(self):
        self._b.close()


class Psd:
    """One stage: `Psd<N>` with the `PsdStage` trait (src/psd.rs:122-288).

        s = Psd(512)                     Psd::<N>::new(fft, Arc::new(Window::hann()))   src/psd.rs:137
        y = s.process(x)                 PsdStage::process(&x, &mut y) -> &mut y[..n]   src/psd.rs:196-269
        s.spectrum(), s.gain(), s.count(), s.buf()                                      src/psd.rs:271-287
    """

    def __init__(self, n, window=Window.HANN, device=0, _handle=None):
        self.n, self.window, self.device = n, window, device
        self._L = lib()
        if _handle is not None:
            self._s = _handle
        elif isinstance(window, WindowTable):
            w = np.ascontiguousarray(window.win, dtype=np.float32)
            if w.size != n:
                raise PsdError(ERR_ARG, "window table length != n")
            self._s = self._L.psdc_stage_create_window(n, _fptr(w), window.power, window.nenbw, window.overlap, device)
        else:
            self._s = self._L.psdc_stage_create(n, int(window), device)
        if not self._s:
            _raise(ERR_DEVICE)

    @staticmethod
    def new(fft_len, win, n=None, device=0):
        """`Psd::<N>::new(fft, win)` (src/psd.rs:137-152): `fft_len` stands for the plan (`fft.len()`), `win` is a
        WindowTable; assert_eq!(N, fft.len()) (src/psd.rs:139) is kept."""
        n = len(win.win) if n is None else n
        if fft_len != n:
            raise PsdError(ERR_ARG, f"assertion failed: N == fft.len() ({n} vs {fft_len}) (src/psd.rs:139)")
        return Psd(n, win, device)

    def close(self):
        if getattr(self, "_s", None):
            self._L.psdc_stage_destroy(self._s)
            self._s = None

    __del__ = close

    def _ck(self, rc):
        if rc < 0:
            msg = self._L.psdc_stage_last_error(self._s)
            raise PsdError(rc, msg.decode() if msg else "")
        return rc

    def clone(self):
        h = self._L.psdc_stage_clone(self._s)
        if not h:
            _raise(ERR_DEVICE)
        return Psd(self.n, self.window, self.device, _handle=h)

    def set_avg(self, avg):
        self._ck(self._L.psdc_stage_set_avg(self._s, avg))

    def set_detrend(self, d):
        self._ck(self._L.psdc_stage_set_detrend(self._s, int(d)))

    def process(self, x, y=None):
        """Returns the written prefix of y (allocated with x.len()/8 + N/8 items when not given)."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        if y is None:
            y = np.empty(x.size // 8 + self.n // 8, dtype=np.float32)
        assert y.dtype == np.float32 and y.flags.c_contiguous
        m = C.c_size_t()
        self._ck(self._L.psdc_stage_process(self._s, _fptr(x), x.size, _fptr(y), y.size, C.byref(m)))
        return y[:m.value]

    def process_device(self, x_ptr, length, y_ptr, cap):
        m = C.c_size_t()
        self._ck(self._L.psdc_stage_process_device(self._s, C.c_void_p(x_ptr), length, C.c_void_p(y_ptr), cap,
                                                   C.byref(m)))
        return m.value

    def spectrum(self):
        out = np.empty(self.n // 2 + 1, dtype=np.float32)
        self._ck(self._L.psdc_stage_get_spectrum(self._s, _fptr(out)))
        return out

    def count(self):
        c = C.c_uint32()
        self._ck(self._L.psdc_stage_get_count(self._s, C.byref(c)))
        return c.value

    def gain(self):
        g = C.c_float()
        self._ck(self._L.psdc_stage_get_gain(self._s, C.byref(g)))
        return g.value

    def buf(self):
        ln = C.c_size_t()
        self._ck(self._L.psdc_stage_get_buf(self._s, None, 0, C.byref(ln)))
        out = np.empty(ln.value, dtype=np.float32)
        if ln.value:
            self._ck(self._L.psdc_stage_get_buf(self._s, _fptr(out), out.size, C.byref(ln)))
        return out


# ---- pure host helpers (no device) -----------------------------------------

def hbf_response_length(depth=3):
    """idsp::hbf::hbf_dec_response_length (src/psd.rs:149,622)."""
    return lib().psdc_hbf_response_length(depth)


def plan_counts(n, total, window=Window.HANN, cap=32):
    """Closed-form (received, segments, pending) per stage after `total` samples."""
    r = (C.c_uint64 * cap)()
    s = (C.c_uint64 * cap)()
    p = (C.c_uint64 * cap)()
    k = lib().psdc_plan_counts(n, int(window), total, cap, r, s, p)
    if k < 0:
        _raise(k)
    return [(int(r[i]), int(s[i]), int(p[i])) for i in range(min(k, cap))]


def stitch(n, counts, avgs, pendings, spectra, opts=MergeOpts(), window=Window.HANN):
    """PsdCascade::psd (src/psd.rs:479-543) on gathered per-stage data (stage 0 first).  `counts` may exceed
    u32 (the library counts in 64 bits); `window` is a kind or a WindowTable."""
    L = lib()
    ns = len(counts)
    if isinstance(window, WindowTable) or any(int(c) > U32_MAX for c in counts):
        wt = window if isinstance(window, WindowTable) else WindowTable._kind(n, window)
        c64 = (C.c_uint64 * max(1, ns))(*[int(c) for c in counts])
        aa = (C.c_uint32 * max(1, ns))(*avgs)
        pp = (C.c_uint64 * max(1, ns))(*pendings)
        sp = np.ascontiguousarray(spectra, dtype=np.float32).reshape(ns, n // 2 + 1) if ns else np.zeros((1, 1), np.float32)
        out = np.empty(max(1, ns * (n // 2 + 1)), dtype=np.float32)
        br = (_CBreak * max(1, ns))()
        plen, nb = C.c_size_t(), C.c_size_t()
        rc = L.psdc_stitch_window(n, wt.power, wt.nenbw, wt.overlap, ns, c64, aa, pp, _fptr(sp), int(opts.keep_overlap),
                                  opts.min_count, int(opts.keep_transition_band), _fptr(out), out.size, C.byref(plen), br,
                                  ns, C.byref(nb))
        if rc < 0:
            _raise(rc)
        return out[:plen.value].copy(), [Break._from_c(br[i]) for i in range(nb.value)]
    cc = (C.c_uint32 * max(1, ns))(*counts)
    aa = (C.c_uint32 * max(1, ns))(*avgs)
    pp = (C.c_uint64 * max(1, ns))(*pendings)
    sp = np.ascontiguousarray(spectra, dtype=np.float32).reshape(ns, n // 2 + 1) if ns else np.zeros((1, 1), np.float32)
    out = np.empty(max(1, ns * (n // 2 + 1)), dtype=np.float32)
    br = (_CBreak * max(1, ns))()
    plen, nb = C.c_size_t(), C.c_size_t()
    rc = L.psdc_stitch(n, int(window), ns, cc, aa, pp, _fptr(sp), int(opts.keep_overlap),
                       opts.min_count, int(opts.keep_transition_band), _fptr(out), out.size,
                       C.byref(plen), br, ns, C.byref(nb))
    if rc < 0:
        _raise(rc)
    return out[:plen.value].copy(), [Break._from_c(br[i]) for i in range(nb.value)]


def readout_bytes(n, n_channels):
    return lib().psdc_readout_bytes(n, n_channels)


def pack_record(n, channels, window=Window.HANN, rows=None):
    """A packed read-out record (psdc_pack_init / psdc_pack_channel, pure host) from stage data held by the caller:
    channels = [(counts, avgs, pendings, spectra[ns, n/2+1]) ...]; `rows` >= len(channels) pads with empty channels."""
    L = lib()
    wt = window if isinstance(window, WindowTable) else WindowTable._kind(n, window)
    rows = max(rows or 0, len(channels))
    buf = np.empty(L.psdc_readout_bytes(n, rows), dtype=np.uint8)
    rc = L.psdc_pack_init(buf.ctypes.data_as(C.c_void_p), buf.size, n, wt.power, wt.nenbw, wt.overlap, rows)
    if rc < 0:
        _raise(rc)
    for c, (counts, avgs, pend, sp) in enumerate(channels):
        ns = len(counts)
        sp = np.ascontiguousarray(sp, dtype=np.float32).reshape(ns, n // 2 + 1) if ns else np.zeros((1, 1), np.float32)
        rc = L.psdc_pack_channel(buf.ctypes.data_as(C.c_void_p), buf.size, c, ns,
                                 (C.c_uint64 * max(1, ns))(*[int(v) for v in counts]),
                                 (C.c_uint32 * max(1, ns))(*[int(v) for v in avgs]),
                                 (C.c_uint64 * max(1, ns))(*[int(v) for v in pend]), _fptr(sp))
        if rc < 0:
            _raise(rc)
    return buf


def pack_pad(rec, rows):
    """psdc_pack_pad: `rec` as a record of `rows` channels (the added ones empty) -- equal blocks for a gather."""
    b = np.ascontiguousarray(np.frombuffer(rec, dtype=np.uint8))
    if b.size < 32:
        _raise(ERR_ARG)
    n, nc = (int(v) for v in np.frombuffer(b[8:16].tobytes(), np.uint32))  # header {magic, version, n, n_channels, ...}
    if rows == nc:
        return b
    out = np.empty(lib().psdc_readout_bytes(n, rows), dtype=np.uint8)
    rc = lib().psdc_pack_pad(b.ctypes.data_as(C.c_void_p), b.size, out.ctypes.data_as(C.c_void_p), out.size, rows)
    if rc < 0:
        _raise(rc)
    return out


def unpack_info(buf, channel=0):
    """(n, n_channels, n_stages of `channel`) of a packed read-out record (bytes-like / uint8 array)."""
    b = np.frombuffer(buf, dtype=np.uint8)
    n, nc, ns = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = lib().psdc_unpack_info(b.ctypes.data_as(C.c_void_p), b.size, channel, C.byref(n), C.byref(nc), C.byref(ns))
    if rc < 0:
        _raise(rc)
    return n.value, nc.value, ns.value


def unpack_stitch(buf, channel=0, opts=MergeOpts()):
    """PsdCascade::psd of one channel of a packed read-out record: identical to psd() on the handle that packed it."""
    b = np.frombuffer(buf, dtype=np.uint8)
    n, _, ns = unpack_info(b, channel)
    out = np.empty(max(1, ns * (n // 2 + 1)), dtype=np.float32)
    br = (_CBreak * max(1, ns))()
    plen, nb = C.c_size_t(), C.c_size_t()
    rc = lib().psdc_unpack_stitch(b.ctypes.data_as(C.c_void_p), b.size, channel, int(opts.keep_overlap), opts.min_count,
                                  int(opts.keep_transition_band), _fptr(out), out.size, C.byref(plen), br, ns, C.byref(nb))
    if rc < 0:
        _raise(rc)
    return out[:plen.value].copy(), [Break._from_c(br[i]) for i in range(nb.value)]


def var_eval(phase_psd, frequencies, tau, x_exp=-2, sinx_exp=4, clip=3.4028234663852886e38, dc_cut=2):
    """Var::eval (src/var.rs:26-45) with VarBuilder defaults (src/var.rs:7-17)."""
    p = np.ascontiguousarray(phase_psd, dtype=np.float32)
    f = np.ascontiguousarray(frequencies, dtype=np.float32)
    return float(lib().psdc_var_eval(x_exp, sinx_exp, clip, dc_cut, _fptr(p), _fptr(f), p.size, tau))


def trace_plot(psd, frequencies, fs=1.0, integrate=False, integral_start=0.0, integral_end=float("inf"), plot=True):
    """Trace::plot (src/bin/psd.rs:125-157): (integrated rms over [integral_start, integral_end] Hz, plot points)."""
    p = np.ascontiguousarray(psd, dtype=np.float32)
    f = np.ascontiguousarray(frequencies, dtype=np.float32)
    assert p.size == f.size
    rms, npts = C.c_float(), C.c_size_t()
    xy = np.empty((max(1, p.size), 2), dtype=np.float64)
    rc = lib().psdc_trace_plot(_fptr(p), _fptr(f), p.size, fs, int(integrate), integral_start, integral_end,
                               C.byref(rms), xy.ctypes.data_as(C.POINTER(C.c_double)) if plot else None, p.size,
                               C.byref(npts))
    if rc < 0:
        _raise(rc)
    return rms.value, (xy[:npts.value].copy() if plot else None)


def hbf_dec8(x, device=0):
    """HbfDec8 block processing from zero state on the device (src/psd.rs:246-253)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty(x.size // 8, dtype=np.float32)
    rc = lib().psdc_hbf_dec8(device, _fptr(x), x.size, _fptr(y))
    if rc < 0:
        _raise(rc)
    return y


def fill_noise_device(ptr, length, seed, first_index=0, device=0):
    rc = lib().psdc_fill_noise_device(device, C.c_void_p(ptr), length, seed, first_index)
    if rc < 0:
        _raise(rc)


def noise_host(length, seed, first_index=0):
    """Host twin of psdc_fill_noise_device: (u - 0.5) * sqrt(12), u from SplitMix64 seeded with
    mix64(seed + GAMMA), outputs first_index, first_index + 1, ... (src/psd.rs:604-606)."""
    gamma = np.uint64(0x9E3779B97F4A7C15)

    def mix64(x):
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))

    with np.errstate(over="ignore"):
        key = mix64(np.array([seed & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64) + gamma)[0]
        i = np.arange(length, dtype=np.uint64) + np.uint64((first_index + 1) & 0xFFFFFFFFFFFFFFFF)
        x = mix64(key + i * gamma)
    u = (x >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return ((u - np.float32(0.5)) * np.float32(3.4641016151377544)).astype(np.float32)


# ---- feed side (src/source.rs, src/de): byte formats only -------------------

ADCDAC_TRACES = ("ADC0", "ADC1", "DAC0", "DAC1")  # src/de/data.rs:37-80


def make_adcdac_frames(traces_i16, batches, seq0=0):
    """Serialise four int16 traces into AdcDac frames (src/de/frame.rs:5-37, data.rs:13).

    traces_i16: array [4, n_samples] of the RAW wire words (the DAC words are
    offset-binary on the wire, src/de/data.rs:64).  n_samples must be a multiple
    of 8*batches.  Returns (bytes, frame_size)."""
    t = np.ascontiguousarray(traces_i16, dtype="<i2")
    assert t.shape[0] == 4 and t.shape[1] % (8 * batches) == 0 and 0 < batches < 256
    nf = t.shape[1] // (8 * batches)
    frame_size = 8 + 64 * batches
    out = np.zeros((nf, frame_size), dtype=np.uint8)
    out[:, 0], out[:, 1], out[:, 2], out[:, 3] = 0x7B, 0x05, 1, batches
    seq = (seq0 + np.arange(nf, dtype=np.uint64) * batches).astype("<u4")
    out[:, 4:8] = seq.view(np.uint8).reshape(nf, 4)
    # payload: frame, batch, channel, 8 samples
    pay = t.reshape(4, nf, batches, 8).transpose(1, 2, 0, 3)
    out[:, 8:] = np.ascontiguousarray(pay).view(np.uint8).reshape(nf, 64 * batches)
    return out.tobytes(), frame_size
