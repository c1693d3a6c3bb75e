"""BASELINE config 3: 4-channel dual-iir (AdcDac) frame stream, N=4096, 50 % overlap, one GPU.
Frames live in host memory (as read from a frame file / UDP); rate = samples of all 4 traces per second."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
n, batches = 4096, 22
nframes = 95325  # ~2^24 samples per channel
rng = np.random.default_rng(1)
raw = np.clip(np.round(rng.standard_normal((4, nframes * batches * 8)) * 4096), -32768, 32767).astype(np.int16)
data, fs = pkg.make_adcdac_frames(raw, batches)
g = pkg.PsdCascadeBank(n, 4)
g.process_adcdac_frames(data, fs)
g.sync()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    g.process_adcdac_frames(data, fs)
g.sync()
dt = time.perf_counter() - t0
samples = reps * raw.size
print(f"frames path (host memory): {samples / dt / 1e6:.0f} MS/s over 4 traces ({len(data) * reps / dt / 1e9:.2f} GB/s of frame bytes), "
      f"{g.num_stages(0)} stages, loss {g.loss()}")
# the same frames resident in HBM (psdc_process_adcdac_frames_device): BASELINE config 3 with the input already on the GPU
import torch
d = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
torch.cuda.synchronize()
g2 = pkg.PsdCascadeBank(n, 4)
for _ in range(3):
    g2.process_adcdac_frames_device(d.data_ptr(), fs, nframes)
g2.sync()
reps = 40
t0 = time.perf_counter()
for _ in range(reps):
    g2.process_adcdac_frames_device(d.data_ptr(), fs, nframes)
g2.sync()
dt = time.perf_counter() - t0
samples = reps * raw.size
print(f"frames path (device memory): {samples / dt / 1e6:.0f} MS/s over 4 traces ({len(data) * reps / dt / 1e9:.2f} GB/s of frame bytes = "
      f"{len(data) * reps / dt / 8e12:.3f} of the HBM roofline on the algorithmic bytes), {g2.num_stages(0)} stages, loss {g2.loss()}")
# the other three payload formats through psdc_process_frames (host memory; one sample per batch and trace): frames of 60 / 18 / 25
# batches with pseudo-random payload words, ~2^22 samples per trace a call
for fmt, bb, nb, name in ((4, 24, 60, "Mpll"), (3, 80, 18, "ThermostatEem"), (2, 56, 25, "Fls")):
    nfr = (1 << 22) // nb
    fs2 = 8 + bb * nb
    fr = np.zeros((nfr, fs2), dtype=np.uint8)
    fr[:, 0], fr[:, 1], fr[:, 2], fr[:, 3] = 0x7B, 0x05, fmt, nb
    fr[:, 4:8] = (np.arange(nfr, dtype=np.uint64) * nb).astype("<u4").view(np.uint8).reshape(nfr, 4)
    if fmt == 3:
        fr[:, 8:] = rng.standard_normal((nfr, bb * nb // 4)).astype("<f4").view(np.uint8).reshape(nfr, bb * nb)
    else:
        fr[:, 8:] = rng.integers(-(1 << 20), 1 << 20, size=(nfr, bb * nb // 4)).astype("<i4").view(np.uint8).reshape(nfr, bb * nb)
    buf = fr.tobytes()
    ntr = 3 if fmt == 4 else 4
    g3 = pkg.PsdCascadeBank(1024, ntr)
    g3.process_frames(buf, fs2)
    g3.sync()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        g3.process_frames(buf, fs2)
    g3.sync()
    dt = time.perf_counter() - t0
    print(f"{name} frames (host memory, N = 1024): {reps * nfr * nb * ntr / dt / 1e6:.0f} MS/s over {ntr} traces ({len(buf) * reps / dt / 1e9:.2f} GB/s of frame bytes)")
    g3.close()
    dv = torch.from_numpy(np.frombuffer(buf, dtype=np.uint8).copy()).cuda()
    torch.cuda.synchronize()
    g4 = pkg.PsdCascadeBank(1024, ntr)
    g4.process_frames_device(dv.data_ptr(), fs2, nfr)
    g4.sync()
    reps = 40
    t0 = time.perf_counter()
    for _ in range(reps):
        g4.process_frames_device(dv.data_ptr(), fs2, nfr)
    g4.sync()
    dt = time.perf_counter() - t0
    print(f"{name} frames (device memory, N = 1024): {reps * nfr * nb * ntr / dt / 1e6:.0f} MS/s over {ntr} traces ({len(buf) * reps / dt / 1e9:.2f} GB/s of frame bytes)")
    g4.close()
