"""Feed side: the reference's file formats through the Source mirror."""
import os
import numpy as np
import pytest

from conftest import test_signal as make_signal


def _raw_file(tmp_path, x):
    p = tmp_path / "trace.f32"
    x.astype("<f4").tofile(p)  # stream_to_raw.rs:24-25 layout
    return str(p)


def test_raw_get_granularity_and_repeat(pkg, tmp_path):
    from stabilizer_stream_amd import source
    x = make_signal(pkg, 1300, seed=1)
    s = source.Source(source.SourceOpts(raw=_raw_file(tmp_path, x)))
    got = []
    sizes = []
    while True:
        (name, v), = s.get()
        assert name == "raw"
        if s.eof:
            break
        sizes.append(v.size)
        got.append(v)
    for _ in range(3):  # past the end the reference keeps returning Ok(vec![("raw", vec![])]) (src/source.rs:151-157)
        (name, v), = s.get()
        assert name == "raw" and v.size == 0 and s.eof
    assert sizes == [512, 512, 276]  # <= 2048 B per call (src/source.rs:150-157)
    assert np.array_equal(np.concatenate(got), x)
    r = source.Source(source.SourceOpts(raw=_raw_file(tmp_path, x), repeat=True))
    tot = sum(r.get()[0][1].size for _ in range(7))
    assert tot == 2 * 1300 + 512  # wraps at EOF (src/source.rs:152-155)


def test_frame_get_matches_oracle_and_counts_loss(pkg, ora, tmp_path):
    from stabilizer_stream_amd import source
    raw = np.random.default_rng(2).integers(-32768, 32768, size=(4, 8 * 3 * 10)).astype(np.int16)
    data, fs = pkg.make_adcdac_frames(raw, 3, seq0=0xFFFFFFF0)
    frames = [data[i * fs:(i + 1) * fs] for i in range(10)]
    del frames[4:6]  # lose two frames = 6 batches
    p = tmp_path / "frames.bin"
    p.write_bytes(b"".join(frames))
    s = source.Source(source.SourceOpts(file=str(p), frame_size=fs))
    n = 0
    while True:
        try:
            traces = s.get()
        except EOFError:
            break
        st, seq, nb, tr = ora.adcdac_decode(frames[n])
        assert [t[0] for t in traces] == list(pkg.ADCDAC_TRACES)
        for (name, v), r in zip(traces, tr):
            assert np.array_equal(v, r)
        n += 1
    assert n == 8 and (s.received, s.dropped) == (24, 6)
    assert s.finish() == pytest.approx(6 / 30)
    with pytest.raises(ValueError):
        source.decode_adcdac_frame(b"\x00" * fs)


@pytest.mark.gpu
def test_feeders_on_gpu(pkg, ora, gpu_required, tmp_path):
    """Bulk feeders give the same PSD as the reference-granularity get() loop, and the frame path
    reports the reference's Loss counters."""
    from stabilizer_stream_amd import source
    from test_gpu_parity import check_against_oracle
    n = 256
    x = make_signal(pkg, 200000, seed=3, tone=0.2)
    path = _raw_file(tmp_path, x)
    bulk = pkg.PsdCascadeBank(n)
    s = source.Source(source.SourceOpts(raw=path))
    while s.feed(bulk, max_bytes=1 << 18):
        pass
    check_against_oracle(pkg, ora, bulk, [x], n, what="raw bulk feed")
    small = pkg.PsdCascadeBank(n)
    s = source.Source(source.SourceOpts(raw=path))
    while True:
        v = s.get()[0][1]
        if s.eof:
            break
        small.process(0, v)
    for k in range(bulk.num_stages()):
        assert bulk.stage_info(0, k) == small.stage_info(0, k)
        a, b = bulk.stage_spectrum(0, k), small.stage_spectrum(0, k)
        assert np.allclose(a, b, rtol=2e-6, atol=1e-6 * float(np.mean(a)))
    # frames with a gap
    raw = np.random.default_rng(4).integers(-32768, 32768, size=(4, 8 * 22 * 200)).astype(np.int16)
    data, fs = pkg.make_adcdac_frames(raw, 22, seq0=5)
    frames = [data[i * fs:(i + 1) * fs] for i in range(200)]
    del frames[50:53]
    fp = tmp_path / "frames.bin"
    fp.write_bytes(b"".join(frames))
    g = pkg.PsdCascadeBank(n, 4)
    s = source.Source(source.SourceOpts(file=str(fp), frame_size=fs))
    while s.feed(g, max_bytes=60 * fs):
        pass
    assert g.loss() == {"received": 197 * 22, "dropped": 3 * 22}
    keep = np.r_[0:50, 53:200]
    for c in range(4):
        st_all = [ora.adcdac_decode(frames[i])[3][c] for i in range(197)]
        check_against_oracle(pkg, ora, g, [np.concatenate(st_all)], n, channel=c, what=f"frames ch{c}")
    bulk.close()
    small.close()
    g.close()


def test_stream_to_raw_tool(pkg, ora, tmp_path):
    """tools/stream_to_raw.py = src/bin/stream_to_raw.rs over `Source.get()`: trace 2 of a Mpll frame file comes out as the f32 file
    `--raw` reads back (the oracle's decode of the same frames, bit for bit)."""
    import struct
    import subprocess
    import sys
    rng = np.random.default_rng(5)
    nb, fs = 60, 8 + 24 * 60
    frames = b"".join(bytes([0x7B, 0x05, 4, nb]) + struct.pack("<I", k * nb) + rng.integers(0, 256, size=24 * nb, dtype=np.uint8).tobytes()
                      for k in range(7))
    p = tmp_path / "mpll.bin"
    p.write_bytes(frames)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "stream_to_raw.py"), "--file", str(p), "--frame-size", str(fs),
                        "--trace", "2"], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    want = np.concatenate([ora.frame_decode(frames[k * fs:(k + 1) * fs])[4][2][1] for k in range(7)])
    got = np.frombuffer(r.stdout, dtype="<f4")
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
def test_psd_cli_tool(pkg, ora, gpu_required, tmp_path):
    """tools/psd_cli.py = the receiver loop of src/bin/psd.rs:158-221 without the GUI: a Mpll frame file (the reference CLI's default
    frame size) through `Source` into one PsdCascade<512> per trace with the reference's default AcqOpts (detrend mean, avg_max 1000),
    read out as `Cmd::Send` does.  The printed integrated RMS of every trace against the oracle's cascade + Trace::plot on the oracle's
    decode of the same frames."""
    import struct
    import subprocess
    import sys
    rng = np.random.default_rng(12)
    nb, fs, nframes = 60, 1448, 700
    w = rng.integers(-(1 << 31), 1 << 31, size=(nframes, nb, 6), dtype=np.int64).astype(np.int32)
    frames = b"".join(bytes([0x7B, 0x05, 4, nb]) + struct.pack("<I", k * nb) + w[k].astype("<i4").tobytes() for k in range(nframes))
    p = tmp_path / "mpll.bin"
    p.write_bytes(frames)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "psd_cli.py"), "--file", str(p), "--fs", "781250", "--integrate",
                        "--csv", str(tmp_path / "csv")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    traces = [[] for _ in range(3)]
    for k in range(nframes):
        st, fmt, seq, bat, tr = ora.frame_decode(frames[k * fs:(k + 1) * fs])
        assert st == 0
        for i, (_, v) in enumerate(tr):
            traces[i].append(v)
    for i, name in enumerate(("phase (rad)", "frequency (kHz)", "amplitude (V/G10)")):
        o = ora.PsdCascade(512, "f64")
        o.set_detrend("mean")
        o.set_avg(999, 0xFFFFFFFE)  # AcqOpts::avg_opts: avg_max - 1, avg - 1 (src/bin/psd.rs:74-79)
        o.process(np.concatenate(traces[i]))
        psd, _, br = o.psd(keep_overlap=False, min_count=1, keep_transition_band=False)
        rms, xy = ora.trace_plot(psd.astype(np.float32), o.frequencies(br), fs=781250.0, integrate=True, integral_start=1e-6 * 781250.0,
                                 integral_end=0.5 * 781250.0)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith(name + ":")]
        assert len(line) == 1, r.stdout
        got = float(line[0].rsplit("rms", 1)[1])
        assert got == pytest.approx(rms, rel=2e-5), (name, got, rms)
        assert f"stages {o.num_stages} " in line[0] and f"bins {psd.size} " in line[0]
        csv = np.loadtxt(tmp_path / "csv" / ("".join(ch if ch.isalnum() else "_" for ch in name) + ".csv"), delimiter=",", ndmin=2)
        assert csv.shape == xy.shape and np.allclose(csv[:, 0], xy[:, 0], rtol=1e-6)
    assert "loss: 0 of 42000 batches" in r.stdout
