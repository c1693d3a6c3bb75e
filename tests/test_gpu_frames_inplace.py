"""AdcDac frames resident in HBM read IN PLACE by the stage-0 loads of the fused kernels (N >= 256, Hann):
psdc_process_adcdac_frames_device never writes the four f32 traces to memory -- a lane's four consecutive samples are
one 8-byte load of wire words (src/de/data.rs:13), converted in registers (:28-35, :64, :75).  The traces the
reference's decode produces (the oracle's restatement of src/de/frame.rs + src/de/data.rs), fed to oracle cascades,
are the truth; counters, Loss (src/loss.rs) and de::Error behaviour as on the host-memory path."""
import numpy as np
import pytest

from conftest import assert_psd_close
from test_gpu_parity import check_against_oracle

pytestmark = pytest.mark.gpu


def make_frames(pkg, ora, nframes, batches, seed, seq0=0, scale=3000.0):
    """(frame bytes as [nframes, frame_size] uint8, frame_size, the four traces as the reference decodes them)"""
    rng = np.random.default_rng(seed)
    raw = np.clip(np.round(rng.standard_normal((4, nframes * batches * 8)) * scale), -32768, 32767).astype(np.int16)
    data, fs = pkg.make_adcdac_frames(raw, batches, seq0=seq0)
    buf = np.frombuffer(data, dtype=np.uint8).copy().reshape(nframes, fs)
    lsb = np.float32(4.096) * np.float32(2.5) / np.float32(32768)
    traces = [raw[c].astype(np.float32) * lsb for c in range(2)]
    traces += [(raw[c].view(np.uint16) ^ np.uint16(0x8000)).view(np.int16).astype(np.float32) * lsb for c in (2, 3)]
    for c in range(4):  # the oracle's own decode of a few frames, held against the vectorised one above
        for f in (0, nframes // 2, nframes - 1):
            st, _, nb, tr = ora.adcdac_decode(buf[f].tobytes())
            assert st == 0 and nb == batches and np.array_equal(tr[c], traces[c][f * batches * 8:(f + 1) * batches * 8])
    return buf, fs, traces


@pytest.mark.parametrize("n,batches,detrend", [(2048, 22, "none"), (4096, 22, "none"), (4096, 7, "mean"), (4096, 1, "span"),
                                               (8192, 31, "midpoint"), (16384, 22, "none"), (4096, 13, "none"),
                                               (256, 22, "none"), (512, 22, "mean"), (512, 3, "midpoint"), (1024, 22, "none"),
                                               (1024, 9, "span"),
                                               # sizes without a fused kernel: every call is decoded into the stage-0 streams (adcdac_kernel)
                                               (64, 5, "mean"), (128, 22, "none"), (1200, 22, "none")])
def test_frames_in_place(pkg, ora, gpu_required, n, batches, detrend):
    """Uneven calls (long enough to be read in place, and short ones that are decoded), a read-out in between, the u32
    sequence wrapping, a gap: every stage of the four cascades against the oracle on the decoded traces."""
    import torch
    per_frame = batches * 8
    nframes = (160 * n) // per_frame + 37
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=n + batches, seq0=0xFFFFFFF0)
    gap_at = nframes // 3
    buf[gap_at:, 4:8] = (buf[gap_at:, 4:8].copy().view("<u4") + np.uint32(5 * batches)).view(np.uint8)  # 5 frames lost
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    g = pkg.PsdCascadeBank(n, 4)
    g.set_detrend(pkg.Detrend[detrend.upper()])
    cuts = [0, (40 * n) // per_frame + 3, (40 * n) // per_frame + 5, (97 * n) // per_frame, nframes]  # one 2-frame call
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert g.process_adcdac_frames_device(d.data_ptr() + a * fs, fs, b - a) == b - a
        if a == cuts[1]:
            g.num_stages(0)  # a read-out in mid-stream
    assert g.loss() == {"received": nframes * batches, "dropped": 5 * batches}
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c]], n, detrend=detrend, channel=c,
                             what=f"in-place frames N={n} B={batches} {detrend} {pkg.ADCDAC_TRACES[c]}")
    g.close()


def test_frames_in_place_coalesced_and_mixed(pkg, ora, gpu_required):
    """Calls that share rounds (held until four calls share a round: coalesce = 4), then the same channels fed host
    samples and an f32 device span behind the frames: one stream per trace whatever the source."""
    import torch
    n, batches = 4096, 22
    per_frame = batches * 8
    nframes = (300 * n) // per_frame
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=5)
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    g = pkg.PsdCascadeBank(n, 4)
    g.configure(coalesce=4)
    step = nframes // 6
    for a in range(0, 6 * step, step):
        assert g.process_adcdac_frames_device(d.data_ptr() + a * fs, fs, step) == step
    used = 6 * step * per_frame
    extra = [pkg.noise_host(50 * n + 24, seed=40 + c) for c in range(4)]
    dx = [torch.from_numpy(e[20 * n:]).cuda() for e in extra]
    for c in range(4):
        g.process(c, extra[c][:20 * n])
        g.process_device(c, dx[c].data_ptr(), extra[c].size - 20 * n)
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c][:used], extra[c]], n, channel=c, what=f"coalesced frames + f32, trace {c}")
    g.close()


def test_frames_in_place_errors(pkg, ora, gpu_required):
    """A bad frame in the middle of a long buffer: the frames before it are ingested (in place), the error is the
    reference's (src/de/frame.rs:27-30, src/de/data.rs:23-24), later calls continue the streams."""
    import torch
    n, batches = 4096, 22
    per_frame = batches * 8
    nframes = (60 * n) // per_frame
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=9)
    for pos, (byte, val, code) in enumerate([(0, 0x00, pkg.ERR_FRAME_HEADER), (2, 9, pkg.ERR_FRAME_FORMAT),
                                             (3, batches + 1, pkg.ERR_FRAME_SIZE)]):
        bad = buf.copy()
        at = nframes // 2 + pos
        bad[at, byte] = val
        db = torch.from_numpy(bad.reshape(-1)).cuda()
        g = pkg.PsdCascadeBank(n, 4)
        with pytest.raises(pkg.FrameError) as e:
            g.process_adcdac_frames_device(db.data_ptr(), fs, nframes)
        assert e.value.code == code
        assert g.loss()["received"] == at * batches
        # the caller skips the bad frame and goes on, as the reference's loop does (src/bin/psd.rs:184)
        assert g.process_adcdac_frames_device(db.data_ptr() + (at + 1) * fs, fs, nframes - at - 1) == nframes - at - 1
        for c in range(4):
            x = np.concatenate([traces[c][:at * per_frame], traces[c][(at + 1) * per_frame:]])
            check_against_oracle(pkg, ora, g, [x], n, channel=c, what=f"bad frame {code}, trace {c}")
        g.close()


@pytest.mark.parametrize("n,batches", [(1024, 22), (4096, 7), (256, 31)])
def test_contiguous_frame_calls_merge_into_one_run(pkg, ora, gpu_required, n, batches):
    """PSDC_OPT_MERGE for frames: a run of AdcDac frames that continues the held run in memory (a capture ring handed over piece by
    piece: calls of one frame up to a few hundred) extends the four traces' spans instead of adding spans -- bit-identical accumulators
    and pending samples to the same buffer handed over in ONE call, Loss counted per call all the same, every trace against the oracle."""
    import torch
    nframes = max(600, (40 * n) // (8 * batches))
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=n + batches, seq0=0xFFFFFF00)
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    torch.cuda.synchronize()
    one, many = pkg.PsdCascadeBank(n, 4), pkg.PsdCascadeBank(n, 4)
    assert one.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
    rng = np.random.default_rng(n)
    pos = 0
    first = -(-4 * (n + 288) // (8 * batches)) + 1  # (the first call long enough to be read in place: nothing to extend yet)
    while pos < nframes:
        m = first if pos == 0 else int(min(nframes - pos, rng.choice([1, rng.integers(1, 20), rng.integers(20, 300)])))
        assert many.process_adcdac_frames_device(d.data_ptr() + pos * fs, fs, m) == m
        pos += m
    assert one.loss() == many.loss() == {"received": nframes * batches, "dropped": 0}
    for c in range(4):
        ns = one.num_stages(c)
        assert many.num_stages(c) == ns
        for k in range(ns):
            assert many.stage_info(c, k) == one.stage_info(c, k)
            assert np.array_equal(many.stage_spectrum(c, k).view(np.uint32), one.stage_spectrum(c, k).view(np.uint32)), (c, k)
            assert np.array_equal(many.stage_buf(c, k).view(np.uint32), one.stage_buf(c, k).view(np.uint32)), (c, k)
        check_against_oracle(pkg, ora, many, [traces[c]], n, channel=c, what=f"merged frame calls, trace {c}")
    one.close()
    many.close()


@pytest.mark.parametrize("n,batches", [(1024, 22), (256, 9)])
def test_merged_frame_runs_stop_at_the_hold_cap(pkg, ora, gpu_required, monkeypatch, n, batches):
    """... and a merged run stops growing where its traces would hold more than the cap (2^29 samples a trace; here 2^15,
    `PSDC_DBG_HOLD_LOG2` read when the handle is made): the next call starts a new run of the four traces behind seams, and rounds go
    out as the cap fills -- a capture ring handed over in calls of a few frames, every trace against the oracle, Loss exact."""
    import torch
    per_frame = 8 * batches
    nframes = (5 << 15) // per_frame + 37  # five caps' worth of samples a trace
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=3 * n + batches, seq0=17)
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    torch.cuda.synchronize()
    monkeypatch.setenv("PSDC_DBG_HOLD_LOG2", "15")
    g = pkg.PsdCascadeBank(n, 4)
    monkeypatch.delenv("PSDC_DBG_HOLD_LOG2")
    g.configure(profile=True)
    rng = np.random.default_rng(n + batches)
    pos = 0
    first = -(-4 * (n + 288) // per_frame) + 1
    while pos < nframes:
        m = first if pos == 0 else int(min(nframes - pos, rng.integers(1, 60)))
        assert g.process_adcdac_frames_device(d.data_ptr() + pos * fs, fs, m) == m
        pos += m
    g.sync()
    assert g.profile_read()["launches"] >= 4  # (one round would be a cap that was not in force)
    assert g.loss() == {"received": nframes * batches, "dropped": 0}
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c]], n, channel=c, what=f"merged frame runs under a 2^15 cap, trace {c}")
    g.close()


def test_headers_may_change_once_the_call_has_returned(pkg, ora, gpu_required):
    """include/psdcascade.h, psdc_process_adcdac_frames_device, Lifetime: the verdict launch is the only reader of the 8 header
    bytes and has completed when the call returns -- the payload is read later (held spans share rounds; the tail of a span is
    read by the next round's first launch).  Every call's headers are overwritten with garbage right after the call returns,
    while its spans are still held: Loss and every spectrum must be what the untouched stream gives."""
    import torch
    n, batches, nframes = 1024, 22, 4000
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=77)
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    hdr = d.view(nframes, fs)[:, :8]
    torch.cuda.synchronize()
    g = pkg.PsdCascadeBank(n, 4)
    g.configure(coalesce=4)
    cuts = [0, 700, 1500, 2100, 3000, 3600, nframes]
    for a, b in zip(cuts[:-1], cuts[1:]):
        assert g.process_adcdac_frames_device(d.data_ptr() + a * fs, fs, b - a) == b - a
        hdr[a:b] = 0xEE  # (torch's stream; the library's streams never read these bytes again)
        torch.cuda.synchronize()
    g.sync()
    assert g.loss() == {"received": nframes * batches, "dropped": 0}
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c]], n, channel=c, what=f"headers rewritten, trace {c}")
    g.close()


@pytest.mark.timeout(600)
def test_config3_frames_in_place_full_size(pkg, ora, gpu_required):
    """BASELINE config 3 with the frames resident in HBM, at full size (2^24 samples per trace, four calls): pure 1e-5
    against the f64 oracle on every stage with four averages, and bit-identical accumulators whatever the call split
    that keeps the same rounds (two handles fed the same calls)."""
    import torch
    n, batches = 4096, 22
    nframes = -(-(1 << 24) // (8 * batches))
    per = nframes * 8 * batches
    lsb = np.float32(4.096) * np.float32(2.5) / np.float32(32768)  # f32 arithmetic, as src/de/data.rs:31 (the f64 quotient rounds one ulp lower)
    words = np.stack([np.clip(np.round(pkg.noise_host(per, 0x7654321 + c).astype(np.float64) * 4096), -32768, 32767)
                      .astype(np.int16) for c in range(4)])
    wire = words.copy()
    wire[2:] = (wire[2:].view(np.uint16) ^ np.uint16(0x8000)).view(np.int16)  # DAC words are offset-binary on the wire
    data, fs = pkg.make_adcdac_frames(wire, batches, seq0=0xFFFFF000)
    d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    g, g2 = pkg.PsdCascadeBank(n, 4), pkg.PsdCascadeBank(n, 4)
    q = nframes // 4
    for h in (g, g2):
        for i in range(4):
            a, b = i * q, (nframes if i == 3 else (i + 1) * q)
            assert h.process_adcdac_frames_device(d.data_ptr() + a * fs, fs, b - a) == b - a
        h.sync()
    assert g.loss() == {"received": nframes * batches, "dropped": 0}
    for c in range(4):
        xc = words[c].astype(np.float32) * lsb
        w = check_against_oracle(pkg, ora, g, [xc], n, channel=c, what=f"config 3 in place {pkg.ADCDAC_TRACES[c]}", pure_min_count=4)
        print(f"config 3 (frames in HBM, read in place) {pkg.ADCDAC_TRACES[c]}: worst relative error {w:.3g}")
    g.close()
    g2.close()


@pytest.mark.timeout(900)
def test_one_call_longer_than_a_frame_span(pkg, ora, gpu_required):
    """One call of more than 2^26 samples per trace: the in-place path cuts it into spans of at most FSPAN_MAX_SAMPLES (the
    kernels' cell arithmetic is exact below 2^24 cells), whose seams are stream seams like any other.  ADC0 and DAC1
    against the f64 oracle (pure 1e-5 on stages with four averages); every trace's counters against the closed form."""
    import torch
    n, batches = 4096, 22
    per_frame = 8 * batches
    nframes = (1 << 26) // per_frame + 20011  # 401 312 frames, 70.6 M samples per trace: two spans
    per = nframes * per_frame
    assert per > (1 << 26)
    lsb = np.float32(4.096) * np.float32(2.5) / np.float32(32768)  # f32 arithmetic, as src/de/data.rs:31 (the f64 quotient rounds one ulp lower)
    words = np.stack([np.clip(np.round(pkg.noise_host(per, 0x1234567 + c).astype(np.float64) * 4096), -32768, 32767)
                      .astype(np.int16) for c in range(4)])
    wire = words.copy()
    wire[2:] = (wire[2:].view(np.uint16) ^ np.uint16(0x8000)).view(np.int16)
    data, fs = pkg.make_adcdac_frames(wire, batches, seq0=17)
    d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
    del data, wire
    g = pkg.PsdCascadeBank(n, 4)
    assert g.process_adcdac_frames_device(d.data_ptr(), fs, nframes) == nframes
    g.sync()
    assert g.loss() == {"received": nframes * batches, "dropped": 0}
    infos = [[g.stage_info(c, k) for k in range(g.num_stages(c))] for c in range(4)]
    assert all(i == infos[0] for i in infos)  # same stream length -> same counters on every trace
    for c in (0, 3):
        xc = words[c].astype(np.float32) * lsb
        w = check_against_oracle(pkg, ora, g, [xc], n, channel=c, what=f"70.6 M-sample call, {pkg.ADCDAC_TRACES[c]}",
                                 pure_min_count=4)
        print(f"one call of 2^26 + samples per trace, {pkg.ADCDAC_TRACES[c]}: worst relative error {w:.3g}")
    g.close()


def test_frames_device_rectangular_window(pkg, ora, gpu_required):
    """Window::rectangular() (overlap 0: the generic two-pass kernels): device-resident frames are decoded into the stage-0
    streams instead of being read in place -- same traces, same Loss, same spectra."""
    import torch
    n, batches = 1024, 22
    nframes = (120 * n) // (8 * batches) + 9
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=31)
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    g = pkg.PsdCascadeBank(n, 4, window=pkg.Window.RECTANGULAR)
    half = nframes // 2
    assert g.process_adcdac_frames_device(d.data_ptr(), fs, half) == half
    assert g.process_adcdac_frames_device(d.data_ptr() + half * fs, fs, nframes - half) == nframes - half
    assert g.loss() == {"received": nframes * batches, "dropped": 0}
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c]], n, channel=c, window="rect", what=f"rectangular window, frames in HBM, trace {c}")
    g.close()


@pytest.mark.parametrize("n,detrend,avg", [(512, "none", (7, 900)), (1024, "mean", (40, 3000)), (2048, "span", (3, 50)),
                                           (4096, "none", (25, 2000)), (8192, "midpoint", (5, 100)), (16384, "mean", (2, 30))])
def test_frames_in_place_with_finite_averaging(pkg, ora, gpu_required, n, detrend, avg):
    """AvgOpts { limit, count } (src/psd.rs:360-376, per-stage rule :447-450) on frames read in place: the EWMA + FRAMES variants of
    the team, three-pass and four-pass kernels, against the oracle on the decoded traces."""
    import torch
    batches = 22
    nframes = (220 * n) // (8 * batches) + 5
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=3 * n + avg[0])
    d = torch.from_numpy(buf.reshape(-1)).cuda()
    g = pkg.PsdCascadeBank(n, 4)
    g.set_detrend(pkg.Detrend[detrend.upper()])
    a = pkg.AvgOpts(*avg)
    g.set_avg(a)
    third = nframes // 3
    for lo, hi in ((0, third), (third, 2 * third + 1), (2 * third + 1, nframes)):
        assert g.process_adcdac_frames_device(d.data_ptr() + lo * fs, fs, hi - lo) == hi - lo
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c]], n, detrend=detrend, avg=a, channel=c,
                             what=f"frames in place N={n} {detrend} avg={avg} trace {c}")
    g.close()


@pytest.mark.parametrize("n", [1024, 4096, 8192])
@pytest.mark.parametrize("shift", [1, 2, 4, 7])
def test_frames_at_an_unaligned_base(pkg, ora, gpu_required, n, shift):
    """The ABI takes a plain `const uint8_t *`: a frame buffer whose base is not a multiple of 8 (an odd offset into a
    capture buffer) must give the same traces.  The in-place kernels read wire words with 8-, 4- and 2-byte loads at
    offsets aligned relative to the base only, so such a base takes the byte-wise decode kernel (round-3 advisor finding:
    nothing checked the base, and every earlier test passed `data_ptr() + k * frame_size`, always 8-aligned)."""
    import torch
    batches = 22
    per_frame = batches * 8
    nframes = (60 * n) // per_frame + 11
    buf, fs, traces = make_frames(pkg, ora, nframes, batches, seed=7 * n + shift, seq0=5)
    d = torch.zeros(buf.size + 16, dtype=torch.uint8, device="cuda")
    assert d.data_ptr() % 8 == 0
    d[shift:shift + buf.size] = torch.from_numpy(buf.reshape(-1)).cuda()
    g = pkg.PsdCascadeBank(n, 4)
    cut = nframes // 2 + 1
    assert g.process_adcdac_frames_device(d.data_ptr() + shift, fs, cut) == cut
    assert g.process_adcdac_frames_device(d.data_ptr() + shift + cut * fs, fs, nframes - cut) == nframes - cut
    assert g.loss() == {"received": nframes * batches, "dropped": 0}
    for c in range(4):
        check_against_oracle(pkg, ora, g, [traces[c]], n, channel=c,
                             what=f"frames at base + {shift}, N={n} {pkg.ADCDAC_TRACES[c]}")
    g.close()
