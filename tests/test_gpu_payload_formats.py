"""The reference's other three payload formats -- Fls, ThermostatEem, Mpll (src/de/mod.rs:12-17, src/de/data.rs:84-212) -- through
psdc_process_frames: Frame::from_bytes + Payload::traces + process() of trace i into cascade i (src/bin/psd.rs:174-182).  The decode
is held to the oracle's restatement BIT FOR BIT (it is the same f32 operations in the same order), the cascades to the usual parity."""
import struct

import numpy as np
import pytest

from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu

BB = {1: 64, 2: 56, 3: 80, 4: 24}


def make_frames(fmt, batches, payloads, seq0=0):
    """payloads: [n_frames] byte strings of batches * BB[fmt] bytes"""
    out = bytearray()
    for k, p in enumerate(payloads):
        assert len(p) == batches * BB[fmt]
        out += bytes([0x7B, 0x05, fmt, batches]) + struct.pack("<I", (seq0 + k * batches) & 0xFFFFFFFF) + p
    return bytes(out), 8 + batches * BB[fmt]


def random_payloads(rng, fmt, batches, nframes, wild):
    """wild: every bit pattern (i32 extremes, NaN / infinite f32 words); else values a spectrum can be taken of"""
    words = BB[fmt] // 4
    if wild:
        w = rng.integers(0, 1 << 32, size=(nframes, batches, words), dtype=np.uint64).astype(np.uint32)
        w[:, :, 0][rng.random((nframes, batches)) < 0.1] = 0x7FFFFFFF
        w[:, :, 1][rng.random((nframes, batches)) < 0.1] = 0x80000000
    elif fmt == 3:
        w = rng.standard_normal((nframes, batches, words)).astype(np.float32).view(np.uint32)
    else:
        w = rng.integers(-(1 << 31), 1 << 31, size=(nframes, batches, words), dtype=np.int64).astype(np.int32).view(np.uint32).copy()
        if fmt == 2:  # the 64-bit phase word: keep it within 48 bits, its square must stay finite in f32 after N-fold summation
            w[:, :, 3] = (w[:, :, 3].view(np.int32) >> 16).view(np.uint32)
    return [w[f].astype("<u4").tobytes() for f in range(nframes)]


def oracle_traces(ora, data, fs):
    traces, names = None, None
    for f in range(len(data) // fs):
        st, fmt, seq, nb, tr = ora.frame_decode(data[f * fs:(f + 1) * fs])
        assert st == 0
        if traces is None:
            traces = [[] for _ in range(4)]
        for i, (_, v) in enumerate(tr):
            traces[i].append(v)
    return [np.concatenate(t) if t else np.zeros(0, np.float32) for t in traces]


@pytest.mark.parametrize("fmt,batches", [(2, 25), (3, 18), (4, 60), (4, 255), (3, 1)])
def test_decode_is_bit_identical(pkg, ora, gpu_required, fmt, batches):
    """Fewer samples than one segment: the stage-0 buffer (`Psd::buf`) then holds every decoded sample -- compared with the oracle's
    Payload::traces as bit patterns, on payloads of arbitrary bits (i32::MIN / MAX, NaN and infinite f32 words)."""
    n = 16384
    rng = np.random.default_rng(100 * fmt + batches)
    nframes = max(1, 12000 // batches)
    data, fs = make_frames(fmt, batches, random_payloads(rng, fmt, batches, nframes, wild=True), seq0=0xFFFFFF00)
    g = pkg.PsdCascadeBank(n, 4)
    assert g.process_frames(data[: (nframes // 2) * fs], fs) == nframes // 2
    assert g.process_frames(data[(nframes // 2) * fs:], fs) == nframes - nframes // 2
    want = oracle_traces(ora, data, fs)
    for c in range(len(pkg.TRACE_NAMES[pkg.Format(fmt)])):
        got = g.stage_buf(c, 0)
        assert got.shape == want[c].shape == (nframes * batches,)
        assert np.array_equal(got.view(np.uint32), want[c].view(np.uint32)), (fmt, pkg.TRACE_NAMES[pkg.Format(fmt)][c])
    assert g.loss() == {"received": nframes * batches, "dropped": 0}
    if fmt == 4:
        assert g.num_stages(3) == 0  # Mpll carries three traces: the fourth cascade saw nothing
    g.close()


@pytest.mark.parametrize("fmt,batches,n", [(2, 25, 256), (3, 18, 512), (4, 60, 1024)])
def test_cascades_of_the_three_formats(pkg, ora, gpu_required, fmt, batches, n):
    """Frames in three uneven calls (the reference CLI's default --frame-size 1448 = 60 Mpll / 18 ThermostatEem batches,
    src/source.rs:31): every trace's cascade against the f64 oracle on the oracle-decoded samples."""
    rng = np.random.default_rng(7 * fmt)
    nframes = 40 * n // batches + 3
    data, fs = make_frames(fmt, batches, random_payloads(rng, fmt, batches, nframes, wild=False), seq0=5)
    if fmt in (3, 4):
        assert fs == 1448
    g = pkg.PsdCascadeBank(n, 4)
    cuts = [0, nframes // 7, nframes // 2 + 1, nframes]
    for a, b in zip(cuts, cuts[1:]):
        assert g.process_frames(data[a * fs:b * fs], fs) == b - a
    want = oracle_traces(ora, data, fs)
    for c, name in enumerate(pkg.TRACE_NAMES[pkg.Format(fmt)]):
        check_against_oracle(pkg, ora, g, [want[c]], n, channel=c, what=f"format {fmt} trace {name}")
    g.close()


def test_runs_of_different_formats_in_one_call(pkg, ora, gpu_required):
    """frame_size 8 + 1344 holds 21 AdcDac, 24 Fls or 56 Mpll batches: one call with runs of all three, one frame's format after the
    other as Frame::from_bytes takes them -- trace i of every frame goes to cascade i (src/bin/psd.rs:174-182), channel 3 gets nothing
    from the Mpll frames; Loss counts batches across the formats and sees the gap between the runs."""
    n = 256
    rng = np.random.default_rng(33)
    parts, seq = [], 10
    for fmt, batches, nframes in ((4, 56, 30), (1, 21, 50), (2, 24, 40), (4, 56, 1), (1, 21, 25)):
        if fmt == 1:
            pay = [rng.integers(-3000, 3000, size=batches * 32).astype("<i2").tobytes() for _ in range(nframes)]
        else:
            pay = random_payloads(rng, fmt, batches, nframes, wild=False)
        d, fs = make_frames(fmt, batches, pay, seq0=seq)
        assert fs == 8 + 1344
        parts.append(d)
        seq += nframes * batches + (7 if fmt == 2 else 0)  # seven batches lost after the Fls run
    data = b"".join(parts)
    g = pkg.PsdCascadeBank(n, 4)
    assert g.process_frames(data, fs) == len(data) // fs
    want = oracle_traces(ora, data, fs)
    for c in range(4):
        check_against_oracle(pkg, ora, g, [want[c]], n, channel=c, what=f"mixed formats, cascade {c}")
    tot = 31 * 56 + 75 * 21 + 40 * 24
    assert g.loss() == {"received": tot, "dropped": 7}
    g.close()


def test_frame_errors_of_the_general_call(pkg, ora, gpu_required):
    """de::Error per frame (src/de/frame.rs:27-30, src/de/data.rs:91-93, 149-150, 173-174): the frames before the bad one are ingested."""
    n, batches = 256, 60
    rng = np.random.default_rng(3)
    data, fs = make_frames(4, batches, random_payloads(rng, 4, batches, 12, wild=False))
    want = oracle_traces(ora, data[: 5 * fs], fs)

    def run(mutate, code, n_channels=4):
        bad = bytearray(data)
        mutate(bad)
        g = pkg.PsdCascadeBank(n, n_channels)
        with pytest.raises(pkg.PsdError) as e:
            g.process_frames(bytes(bad), fs)
        assert e.value.code == code, e.value
        got = g.stage_buf(0, 0) if g.num_stages(0) else np.zeros(0, np.float32)
        g.close()
        return got

    def magic(b): b[5 * fs + 1] = 0
    def unknown(b): b[5 * fs + 2] = 5
    def nbatch(b): b[5 * fs + 3] = batches - 1
    for m, code in ((magic, pkg.ERR_FRAME_HEADER), (unknown, pkg.ERR_FRAME_FORMAT), (nbatch, pkg.ERR_FRAME_SIZE)):
        got = run(m, code)
        # five frames went in: 300 samples, one 256-sample segment consumed -> the buffer keeps the last 300 - 128 of them
        assert np.array_equal(got.view(np.uint32), want[0][128:].view(np.uint32))
    # a format whose batch size does not divide the payload: PayloadSize at its first frame
    def as_fls(b): b[5 * fs + 2] = 2
    run(as_fls, pkg.ERR_FRAME_SIZE)
    # ThermostatEem carries four traces (1440 = 18 x 80 fits): three channels are not enough -- but they are for Mpll
    def as_eem(b):
        b[5 * fs + 2] = 3
        b[5 * fs + 3] = 18
    run(as_eem, pkg.ERR_ARG, n_channels=3)
    g = pkg.PsdCascadeBank(n, 3)
    assert g.process_frames(data, fs) == 12
    g.close()
    # psdc_process_adcdac_frames keeps refusing the other formats
    g = pkg.PsdCascadeBank(n, 4)
    with pytest.raises(pkg.FrameError) as e:
        g.process_adcdac_frames(data, fs)
    assert e.value.code == pkg.ERR_FRAME_FORMAT
    g.close()


def test_source_feeds_a_frame_file_of_any_format(pkg, ora, gpu_required, tmp_path):
    """stabilizer-stream_amd/source.py: `get()` decodes one frame per call on the host (the reference's granularity), `feed()` hands
    the same bytes to psdc_process_frames in bulk -- same cascades (Mpll frames at the reference CLI's default frame size)."""
    from stabilizer_stream_amd import source as src
    n, batches = 256, 60
    rng = np.random.default_rng(8)
    data, fs = make_frames(4, batches, random_payloads(rng, 4, batches, 90, wild=False), seq0=77)
    path = tmp_path / "mpll.bin"
    path.write_bytes(data)
    s = src.Source(src.SourceOpts(file=str(path), frame_size=fs), pkg)
    traces = [[] for _ in range(3)]
    try:
        while True:
            t = s.get()
            assert [nm for nm, _ in t] == list(pkg.TRACE_NAMES[pkg.Format.MPLL])
            for i, (_, v) in enumerate(t):
                traces[i].append(v)
    except EOFError:
        pass
    s.close()
    want = oracle_traces(ora, data, fs)
    for i in range(3):
        assert np.array_equal(np.concatenate(traces[i]).view(np.uint32), want[i].view(np.uint32))
    g = pkg.PsdCascadeBank(n, 3)
    s = src.Source(src.SourceOpts(file=str(path), frame_size=fs), pkg)
    while s.feed(g, max_bytes=20 * fs):
        pass
    s.close()
    for i in range(3):
        check_against_oracle(pkg, ora, g, [want[i]], n, channel=i, what=f"Source.feed Mpll trace {i}")
    g.close()


def test_frames_resident_in_device_memory(pkg, ora, gpu_required):
    """psdc_process_frames_device: the same runs of AdcDac / Fls / Mpll frames (frame size 8 + 1344) from a device buffer -- the headers
    are gathered to the host by one small kernel, Fls / Mpll payloads are decoded from the caller's buffer, the AdcDac runs are read in place by
    the fused kernels (N = 1024) -- against the oracle, and against the host-memory call (same Loss, same pending samples bit for bit);
    then the de::Error of a bad frame in the middle of a device-resident Mpll run."""
    import torch
    n = 1024
    rng = np.random.default_rng(44)
    parts, seq = [], 3
    for fmt, batches, nframes in ((2, 24, 300), (1, 21, 500), (4, 56, 200), (1, 21, 90), (2, 24, 1)):
        if fmt == 1:
            pay = [rng.integers(-3000, 3000, size=batches * 32).astype("<i2").tobytes() for _ in range(nframes)]
        else:
            pay = random_payloads(rng, fmt, batches, nframes, wild=False)
        d, fs = make_frames(fmt, batches, pay, seq0=seq)
        parts.append(d)
        seq += nframes * batches + (5 if fmt == 4 else 0)
    data = b"".join(parts)
    nf = len(data) // fs
    dev = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    g = pkg.PsdCascadeBank(n, 4)
    cut = 450  # inside the first AdcDac run
    assert g.process_frames_device(dev.data_ptr(), fs, cut) == cut
    assert g.process_frames_device(dev.data_ptr() + cut * fs, fs, nf - cut) == nf - cut
    g.sync()
    want = oracle_traces(ora, data, fs)
    for c in range(4):
        check_against_oracle(pkg, ora, g, [want[c]], n, channel=c, what=f"device-resident mixed formats, cascade {c}")
    gh = pkg.PsdCascadeBank(n, 4)
    assert gh.process_frames(data, fs) == nf
    assert g.loss() == gh.loss() == {"received": 301 * 24 + 590 * 21 + 200 * 56, "dropped": 5}
    for c in range(4):
        assert np.array_equal(g.stage_buf(c, 0).view(np.uint32), gh.stage_buf(c, 0).view(np.uint32))
    gh.close()
    g.close()
    # a bad frame inside a device-resident run
    d4, fs4 = make_frames(4, 60, random_payloads(rng, 4, 60, 40, wild=False))
    for pos, val, code in ((0, 0x00, pkg.ERR_FRAME_HEADER), (2, 9, pkg.ERR_FRAME_FORMAT), (3, 59, pkg.ERR_FRAME_SIZE)):
        b = bytearray(d4)
        b[17 * fs4 + pos] = val
        dv = torch.from_numpy(np.frombuffer(bytes(b), dtype=np.uint8).copy()).cuda()
        torch.cuda.synchronize()
        g = pkg.PsdCascadeBank(256, 3)
        with pytest.raises(pkg.FrameError) as e:
            g.process_frames_device(dv.data_ptr(), fs4, 40)
        assert e.value.code == code
        w = oracle_traces(ora, d4[: 17 * fs4], fs4)
        check_against_oracle(pkg, ora, g, [w[0]], 256, channel=0, what="frames before the bad one")
        assert g.loss()["received"] == 17 * 60
        g.close()


@pytest.mark.parametrize("fmt,batches,shift", [(4, 60, 1), (2, 25, 2), (3, 18, 7)])
def test_device_frames_at_an_unaligned_base(pkg, ora, gpu_required, fmt, batches, shift):
    """Any base address is legal (the API takes bytes): the decode kernel then assembles its words from bytes -- same bits."""
    import torch
    rng = np.random.default_rng(50 + fmt)
    nframes = 9000 // batches
    data, fs = make_frames(fmt, batches, random_payloads(rng, fmt, batches, nframes, wild=True))
    dev = torch.zeros(len(data) + 16, dtype=torch.uint8, device="cuda")
    dev[shift:shift + len(data)] = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    g = pkg.PsdCascadeBank(16384, 4)
    assert g.process_frames_device(dev.data_ptr() + shift, fs, nframes) == nframes
    want = oracle_traces(ora, data, fs)
    for c in range(len(pkg.TRACE_NAMES[pkg.Format(fmt)])):
        assert np.array_equal(g.stage_buf(c, 0).view(np.uint32), want[c].view(np.uint32))
    g.close()
