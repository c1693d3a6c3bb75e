"""Feed side: the reference's `Source` for its two file formats (src/source.rs), plus the batched
feeders a GPU path needs.

`Source.get()` keeps the reference contract -- one call returns `[(name, f32 array), ...]`, one entry per
trace, with the reference's own granularity (`Data::Raw`: at most 2048 bytes = 512 samples per call,
src/source.rs:150-157; `Data::File`: one frame of `frame_size` bytes per call, :136-142; `--repeat` wraps
at EOF, :143-145,152-155).  At 10 GS/s that granularity would mean 2e7 calls/s, so `feed()` reads MB-scale
spans of the SAME byte formats and hands them to `psdc_process` / `psdc_process_frames` in bulk:
the cascade's result depends only on the concatenated stream, so both ways give the same PSD.

UDP, the noise generator and the DSM source are host I/O outside the accelerated path.
"""
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np


@dataclass
class SourceOpts:
    """File-backed subset of SourceOpts (src/source.rs:15-48)."""
    file: Optional[str] = None          # frames file (--file)
    frame_size: int = 8 + 30 * 2 * 6 * 4  # default of the reference CLI (src/source.rs:31)
    repeat: bool = False                # --repeat
    raw: Optional[str] = None           # single f32 raw trace, native endian (--raw)


class Source:
    """Source::new / get / finish (src/source.rs:66-171) for Data::File and Data::Raw."""

    RAW_CHUNK = 2048  # bytes per get() (src/source.rs:151)

    def __init__(self, opts: SourceOpts, pkg=None):
        if (opts.file is None) == (opts.raw is None):
            raise ValueError("exactly one of file / raw (UDP, noise, dsm sources are out of scope)")
        self.opts = opts
        self._f = open(opts.file if opts.file else opts.raw, "rb")
        self._pkg = pkg
        self.received = 0  # Loss (src/loss.rs) for the get() path
        self.dropped = 0
        self._seq = None
        self.eof = False  # a raw file is exhausted (get() then returns empty traces, like the reference)

    def close(self):
        self._f.close()

    # -- reference-granularity path -------------------------------------------------
    def get(self):
        """One reference-sized chunk: [(name, np.float32 array)].  End of file without --repeat, as in the reference:
        Data::Raw keeps answering with an EMPTY "raw" trace (`read` returns 0 bytes, src/source.rs:151-157; also for a
        tail shorter than one f32) and sets `self.eof`; Data::File raises EOFError (`read_exact` -> UnexpectedEof,
        src/source.rs:137-146)."""
        if self.opts.raw is not None:
            buf = self._f.read(self.RAW_CHUNK)
            if len(buf) == 0 and self.opts.repeat:  # src/source.rs:152-155
                self._f.seek(0)
                buf = self._f.read(self.RAW_CHUNK)
            self.eof = len(buf) == 0
            n = len(buf) // 4 * 4  # bytemuck::cast_slice(&buf[..len / 4 * 4]) (src/source.rs:156)
            return [("raw", np.frombuffer(buf[:n], dtype="<f4").astype(np.float32))]
        while True:
            buf = self._f.read(self.opts.frame_size)
            if len(buf) < self.opts.frame_size:
                if self.opts.repeat and os.path.getsize(self.opts.file) >= self.opts.frame_size:
                    self._f.seek(0)
                    continue
                raise EOFError  # read_exact: UnexpectedEof (src/source.rs:137-146)
            _, seq, batches, traces = decode_frame(buf)
            self.received += batches  # Loss::update (src/loss.rs:11-26)
            if self._seq is not None:
                self.dropped += (seq - self._seq) & 0xFFFFFFFF
            self._seq = (seq + batches) & 0xFFFFFFFF
            return traces

    # -- batched path ---------------------------------------------------------------
    def feed(self, bank, max_bytes=64 << 20, channel=0):
        """Read up to max_bytes and ingest them in one call.  Returns the bytes consumed (0 at EOF)."""
        if self.opts.raw is not None:
            buf = self._f.read(max_bytes // 4 * 4)
            if len(buf) < 4:
                if self.opts.repeat and os.path.getsize(self.opts.raw) >= 4 and len(buf) == 0:
                    self._f.seek(0)
                    return self.feed(bank, max_bytes, channel)
                return 0
            n = len(buf) // 4 * 4
            bank.process(channel, np.frombuffer(buf[:n], dtype="<f4"))
            return n
        fs = self.opts.frame_size
        buf = self._f.read(max(1, max_bytes // fs) * fs)
        nframes = len(buf) // fs
        if nframes == 0:
            if self.opts.repeat and os.path.getsize(self.opts.file) >= fs:
                self._f.seek(0)
                return self.feed(bank, max_bytes, channel)
            return 0
        bank.process_frames(buf[:nframes * fs], fs)
        return nframes * fs

    def finish(self):
        """Loss::analyze (src/loss.rs:28-38): fraction of dropped batches on the get() path."""
        tot = self.received + self.dropped
        return (self.dropped / tot) if self.received else 0.0


def decode_frame(buf):
    """Frame::from_bytes + Payload::traces on the host for the four formats (src/de/frame.rs:25-60, src/de/data.rs:11-212).
    Only used by the reference-granularity `get()` path; bulk ingest decodes on the device (psdc_process_frames).
    Returns (format id, seq, batches, [(name, f32 array), ...])."""
    if len(buf) < 8:
        raise ValueError("frame shorter than its header")
    if buf[0] != 0x7B or buf[1] != 0x05:
        raise ValueError("Invalid frame header")
    fmt = buf[2]
    if not 1 <= fmt <= 4:
        raise ValueError("Unknown format ID")
    batches = buf[3]
    seq = int.from_bytes(buf[4:8], "little")
    bb = (64, 56, 80, 24)[fmt - 1]  # bytes per batch (src/de/data.rs:13, 86, 144, 168)
    if (len(buf) - 8) % bb or (len(buf) - 8) // bb != batches:
        raise ValueError("Payload size")
    f32 = np.float32
    if fmt == 1:
        pay = np.frombuffer(buf[8:], dtype="<i2")
        lsb = f32(4.096) * f32(2.5) / f32(32768)  # src/de/data.rs:28-35
        d = pay.reshape(batches, 4, 8)
        out = []
        for c, name in enumerate(("ADC0", "ADC1", "DAC0", "DAC1")):
            v = d[:, c, :].reshape(-1)
            if c >= 2:  # i16.wrapping_add(i16::MIN) (src/de/data.rs:64,75)
                v = (v.astype(np.int32) + 32768 + 32768) % 65536 - 32768
            out.append((name, v.astype(np.float32) * lsb))
        return fmt, seq, batches, out
    w = np.frombuffer(buf[8:], dtype="<i4").reshape(batches, bb // 4)
    hyp = lambda a, b: np.sqrt(a.astype(f32) * a.astype(f32) + b.astype(f32) * b.astype(f32))  # powi(2) + powi(2), sqrt: all f32
    two31, two32 = f32(2147483648.0), f32(4294967296.0)  # i32::MAX as f32, (1u64 << 32) as f32
    tau = f32(6.283185307179586)
    if fmt == 2:  # Fls (src/de/data.rs:97-139)
        ph = (w[:, 2].astype(np.int64) & 0xFFFFFFFF) | (w[:, 3].astype(np.int64) << 32)  # b[0][2..4] as one i64
        return fmt, seq, batches, [
            ("AR", hyp(w[:, 0], w[:, 1]) * (f32(1.0) / two31)),
            ("AP", ph.astype(f32) * (tau / f32(65536.0))),
            ("BI", w[:, 7].astype(f32) / two31),
            ("BQ", w[:, 8].astype(f32) / two31)]
    if fmt == 3:  # ThermostatEem (src/de/data.rs:154-163)
        f = np.frombuffer(buf[8:], dtype="<f4").reshape(batches, 20)
        return fmt, seq, batches, [(name, f[:, i].astype(f32)) for name, i in zip(("T00", "T20", "I0", "I1"), (0, 8, 13, 16))]
    return fmt, seq, batches, [  # Mpll (src/de/data.rs:178-211)
        ("phase (rad)", w[:, 4].astype(f32) * (tau / two32)),
        ("frequency (kHz)", w[:, 5].astype(f32) * (f32(1.0) / f32(1.28e-3) / two32)),
        ("amplitude (V/G10)", hyp(w[:, 0], w[:, 1]) * (f32(10.24) / f32(10.0) * f32(2.0) * f32(2.0) / two32))]


def decode_adcdac_frame(buf):
    """decode_frame restricted to AdcDac: (seq, batches, traces)."""
    if len(buf) >= 3 and buf[0] == 0x7B and buf[1] == 0x05 and 2 <= buf[2] <= 4:
        raise ValueError("not an AdcDac frame")
    _, seq, batches, out = decode_frame(buf)
    return seq, batches, out
