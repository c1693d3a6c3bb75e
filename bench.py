#!/usr/bin/env python3
"""Throughput of the cascaded PSD hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 (default): BASELINE.json configs[1] -- 1-channel raw f32, PsdCascade N=1024, >= 6 stages.  A step =
`--passes` (16) consecutive passes of the cascade over one 2^26-sample batch of synthetic raw-f32 samples
already resident in HBM, i.e. 2^30 stream samples per step (the stream simply continues from pass to pass).
N > 1: BASELINE.json configs[3] -- raw f32 channels sharded 8 per GPU (64 channels at N = 8), N=1024,
2^24 samples per channel per pass, `--passes` (8) passes per step; one process per GPU, no data-path
collective, and the final per-stage spectra of all channels gathered to rank 0 over RCCL inside the
timed region (weak scaling: the per-GPU work is fixed).
Started without a launcher (`WORLD_SIZE` unset) and with --gpus N > 1 the script starts the N ranks itself
(`python -m torch.distributed.run`), from a parent that never touches the GPU, and relays rank 0's line.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBPS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP32_VALU_PEAK_TFLOPS = 157.3  # PACKED f32 FMA: 256 CUs x 4 SIMDs x 32 lane-results/clk x 2 flop x 2.4 GHz
# What the fused kernels are built from is scalar (non-packed) f32 VALU -- the build disables packed f32 ops (csrc/Makefile NOPK:
# a packed op is two passes through the SIMD and costs register pairing).  tools/probes/valu_rate.cpp measures the rate the
# SIMDs sustain for dependent-free v_fma_f32 at four wavefronts per SIMD: ~22 lane-results per clock and SIMD (26 for
# v_pk_fma_f32, of the 32 the packed peak assumes).  The scalar-FMA ceiling at the nominal 2.4 GHz:
SCALAR_FMA_LANES_PER_CLK_SIMD = 22.0
FP32_SCALAR_FMA_PEAK_TFLOPS = 256 * 4 * SCALAR_FMA_LANES_PER_CLK_SIMD * 2 * 2.4e9 / 1e12  # = 108.1
BINDING = ("fp32 VALU issue + LDS issue, at the package power limit (SQ counters: VALU active ~0.38, LDS-issue stall ~0.23 of wave "
           "time, ~1350 W at ~2.15 GHz); HBM is NOT the binding resource -- `frac` prices the kernel against HBM because "
           "BASELINE.json's metric asks for that")
# what binds the dominant kernel (the field a script reads); `achieved` / `peak` / `frac` stay priced against HBM (`priced_against`), the
# resource BASELINE.json's metric names
BOUND = "valu+lds issue (power-limited)"
ALG_BYTES_PER_SAMPLE = 4.0   # each raw f32 sample crosses HBM once
ALG_FLOP_PER_SAMPLE = {1024: 78.0, 4096: 89.0, 16384: 100.0}


class DeviceState:
    """Engine clock and socket power of THIS process's GPU, sampled from the amdgpu hwmon files (sysfs reads: no child process,
    nothing on the GPU) by a thread for as long as the `with` block runs.  The card is found through the PCI bus id HIP
    reports; None fields where the box does not expose the files."""

    def __init__(self, torch, device, period_s=0.005):
        import glob
        self.period, self.rows, self._stop, self._th = period_s, [], False, None
        self.fp = self.ff = self.cap = None
        try:
            pr = torch.cuda.get_device_properties(device)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            hw = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
            if hw:
                for name in ("power1_input", "power1_average"):
                    if os.path.exists(os.path.join(hw[0], name)):
                        self.fp = os.path.join(hw[0], name)
                        break
                if os.path.exists(os.path.join(hw[0], "freq1_input")):
                    self.ff = os.path.join(hw[0], "freq1_input")
                try:
                    self.cap = int(open(os.path.join(hw[0], "power1_cap")).read()) / 1e6
                except (OSError, ValueError):
                    pass
                self.src = f"{hw[0]} ({bdf})"
            else:
                self.src = f"no hwmon under /sys/bus/pci/devices/{bdf}"
        except Exception as e:  # noqa: BLE001 -- a diagnostic, never the reason a bench fails
            self.src = f"unavailable: {e}"

    def _read(self, path):
        try:
            return int(open(path).read())
        except (OSError, ValueError, TypeError):
            return None

    def _run(self):
        while not self._stop:
            self.rows.append((self._read(self.ff), self._read(self.fp)))
            time.sleep(self.period)

    def __enter__(self):
        if self.fp or self.ff:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def __exit__(self, *a):
        self._stop = True
        if self._th:
            self._th.join()

    def summary(self):
        f = [r[0] / 1e6 for r in self.rows if r[0]]
        w = [r[1] / 1e6 for r in self.rows if r[1]]
        return {"sclk_mhz_mean": sum(f) / len(f) if f else None, "sclk_mhz_min": min(f) if f else None,
                "sclk_mhz_max": max(f) if f else None, "sclk_mhz_peak_nominal": 2400,
                "socket_power_w_mean": sum(w) / len(w) if w else None, "socket_power_w_max": max(w) if w else None,
                "socket_power_cap_w": self.cap, "samples": len(self.rows), "period_ms": self.period * 1e3, "source": self.src,
                "note": "sampled during the timed region (hwmon freq1_input = sclk, power1_input = socket power)"}


def cpu_baseline(n, seconds=12.0, threads=1):
    """Time the CPU oracle (C restatement, f32) on this host's cores: process() calls of 65536 samples like the
    reference's `insn` bench (src/psd.rs:554-559), one cascade per thread (min(channels, cores)
    threads, one channel per thread).  The library is REBUILT here with this host's -march=native first.  `value` is
    the build with the radix-4 FFT whose loops gcc vectorises (a fairer stand-in for rustfft's SIMD butterflies); the
    scalar radix-2 FFT every parity test sees and N = 512 (the size the reference's own figure is quoted for) ride along."""
    import threading
    ora = entry.load_oracle()
    try:
        ora.build(force=True)  # gcc -O3 -march=native on THIS host (the shipped .so was built elsewhere)
        rebuilt = True
    except Exception:
        rebuilt = False
    pkg = entry.load_package()
    xs = [pkg.noise_host(1 << 16, 0x7654321 + t) for t in range(threads)]

    def leg(nn, fast, secs):
        done = [0] * threads

        def work(t):
            c = ora.PsdCascade(nn, "f32")
            if fast:
                c.set_fast_fft()
            c.process(xs[t])  # warm
            t0 = time.perf_counter()
            k = 0
            while time.perf_counter() - t0 < secs:
                for _ in range(8):
                    c.process(xs[t])
                k += 8
            done[t] = k

        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]  # ctypes releases the GIL in the C calls
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        return sum(done) * (1 << 16) / dt / 1e6, sum(done), dt

    v_fast, calls, dt = leg(n, True, seconds * 0.5)
    v_r2, _, _ = leg(n, False, seconds * 0.25)
    v_512, _, _ = leg(512, True, seconds * 0.25)
    model = "?"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": v_fast, "unit": "MS/s", "cores": threads, "kind": "port",
            "scalar_radix2_fft_value": v_r2, "n512_value": v_512,
            "host": f"{model}, {os.cpu_count()} logical cores visible",
            "rebuilt_on_this_host": rebuilt,
            "sample": f"{calls} process() calls of 65536 samples over {threads} thread(s) (one cascade each) in {dt:.1f} s, N={n}; "
                      f"C restatement of src/psd.rs (oracle/, f32, gcc -O3 -march=native), not the Rust crate: `value` with a "
                      f"radix-4 Stockham FFT gcc vectorises, `scalar_radix2_fft_value` with the scalar radix-2 FFT of the parity "
                      f"tests, `n512_value` = N=512 for the reference's own figure (>200 MS/s/core, N=512, README.md:11, src/psd.rs:550)"}


def host_fed_rate(pkg, n, device, seconds=2.0):
    """PCIe-inclusive ingest: psdc_process() on host memory (copy into pinned staging, H2D, kernels).
    Reported beside the headline, never as `value`."""
    x = pkg.noise_host(1 << 24, 0x7654321)
    bank = pkg.PsdCascadeBank(n, 1, device=device)
    bank.process(0, x)
    bank.sync()
    t0 = time.perf_counter()
    done = 0
    while time.perf_counter() - t0 < seconds:
        bank.process(0, x)
        done += x.size
    bank.sync()
    dt = time.perf_counter() - t0
    bank.close()
    return {"value": done / dt / 1e6, "unit": "MS/s",
            "note": "host numpy buffer -> psdc_process (copy to pinned staging split over <= 4 host threads, "
                    "hipMemcpyAsync, kernels); link-bound"}


def host_fed_small(n, device, seconds=0.5):
    """The boundary at the reference's own call granularity (src/source.rs:150-157: 512 samples per Source::get; src/psd.rs:554-559:
    65536 per call in its `insn` bench), measured from C++ with no Python in the loop: tests/host/smallcall_probe feeds psdc_process
    on HOST memory in calls of 512 / 4096 / 65536 / 2^22 samples and reports MS/s to the drain and ns per call.  A child process
    (this one keeps its own GPU context; nothing of the headline is running any more)."""
    exe = os.path.join(ROOT, "tests", "host", "smallcall_probe")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "host"), "smallcall_probe"], capture_output=True, text=True)
        if r.returncode != 0:
            return {"error": "tests/host/smallcall_probe did not build: " + r.stderr[-300:]}
    r = subprocess.run([exe, str(n), str(seconds), str(device)], capture_output=True, text=True, timeout=120)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        return {"error": f"smallcall_probe rc={r.returncode}: {r.stderr[-300:]}"}
    d = json.loads(line)
    d["note"] = ("psdc_process(host memory) from C++ in calls of 512 / 4096 / 65536 / 2^22 samples (the reference's Source::get, frame and "
                 "`insn` granularities): staging copy, upload and the whole cascade; PCIe-inclusive, never `value`")
    return d


def device_fed_calls(n, device, seconds=0.3):
    """psdc_process_device at call sizes 2^26 ... 2^16 from C++ (tests/host/devcall_probe): "contiguous" = consecutive pieces of one
    buffer (each call continues the last one in memory and extends the held span: PSDC_OPT_MERGE), "scattered" = the same pieces
    in an order in which no call continues its predecessor (every call a span of its own).  A child process, after the timed region."""
    exe = os.path.join(ROOT, "tests", "host", "devcall_probe")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "host"), "devcall_probe"], capture_output=True, text=True)
        if r.returncode != 0:
            return {"error": "tests/host/devcall_probe did not build: " + r.stderr[-300:]}
    r = subprocess.run([exe, str(n), str(seconds), str(device)], capture_output=True, text=True, timeout=180)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
    if r.returncode != 0 or line is None:
        return {"error": f"devcall_probe rc={r.returncode}: {r.stderr[-300:]}"}
    d = json.loads(line)
    d["note"] = ("psdc_process_device from C++, one 2^26-sample device buffer handed over in calls of 2^26 ... 2^16 samples; MS/s to the "
                 "drain and host ns per call; side leg, never `value`")
    return d


def side_leg_raw(pkg, torch, n, log2_batch, device, seconds, buf=None, window=None):
    """A short untimed-by-`value` leg of the same cascade at another shape: 1 channel raw f32, 2^log2_batch samples
    resident in HBM, passes until `seconds` have gone by.  Returns value + kernel-only roofline like the headline's."""
    T = 1 << log2_batch
    if buf is None or buf.numel() != T:
        buf = torch.empty(T, dtype=torch.float32, device="cuda")
        pkg.fill_noise_device(buf.data_ptr(), T, seed=0x7654321, device=device)
        torch.cuda.synchronize()
    bank = pkg.PsdCascadeBank(n, 1, device=device) if window is None else pkg.PsdCascadeBank(n, 1, window=window, device=device)
    for _ in range(4):
        bank.process_device(0, buf.data_ptr(), T)
    bank.sync()
    bank.configure(profile=True)
    t0 = time.perf_counter()
    passes = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            bank.process_device(0, buf.data_ptr(), T)
        passes += 8
    ns = bank.num_stages(0)
    bank.psd(0)  # read-out inside the timed region, like the headline
    dt = time.perf_counter() - t0
    prof = bank.profile_read()
    bank.close()
    kern_s = prof["kernel_ms"] * 1e-3
    ach = ALG_BYTES_PER_SAMPLE * prof["stage0_samples"] / kern_s / 1e9 if kern_s > 0 else 0.0
    return {"value": passes * T / dt / 1e6, "unit": "MS/s",
            "workload": f"1-channel raw f32, PsdCascade N={n}{'' if window is None else ', Window::rectangular() (overlap 0)'}, {passes} passes over 2^{log2_batch} samples "
                        f"resident in HBM ({T * 4 >> 20} MiB), {ns} stages",
            "roofline": {"bound": BOUND, "priced_against": "hbm", "binding": BINDING, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                         "end_to_end_frac": ALG_BYTES_PER_SAMPLE * passes * T / dt / 1e9 / HBM_PEAK_GBPS,
                         "launches": prof["launches"], "avg_launch_ms": prof["kernel_ms"] / max(1, prof["launches"])}}


class FrameReplay:
    """A frame stream replayed from two copies of one buffer with the sequence numbers moving on: a real stream's `seq` advances
    by `batches` per frame (src/de/frame.rs:5-9), and Loss::update (src/loss.rs:11-26) counts a call whose first `seq` is not the
    last call's next one as ~2^32 dropped batches -- which replaying ONE buffer did (round-3 record: `dropped` 5e13).  Call k reads
    buffer k % 2; right after it has been handed over, the `seq` words of that buffer are moved on by two calls' worth on torch's
    stream (4 bytes per header; the payload the in-place kernels may still be reading is not touched -- header bytes are the
    verdict launch's alone and that has completed when the call returns: include/psdcascade.h, Lifetime), long before call k + 2
    waits for that event."""

    def __init__(self, torch, frames_u8, frame_size, n_frames, batches):
        assert frame_size % 4 == 0  # 8 + 64 x batches: the buffer is also an [n_frames, frame_size / 4] array of 32-bit words
        self.torch, self.fs, self.nf, self.batches = torch, frame_size, n_frames, batches
        self.buf = [torch.from_numpy(frames_u8).cuda() for _ in range(2)]
        # word 1 of every frame is its `seq` (src/de/frame.rs:5-9): ONE strided in-place add moves a whole buffer on
        # (int32 arithmetic wraps like the u32 on the wire)
        self.seq = [b.view(torch.int32).view(n_frames, frame_size // 4)[:, 1] for b in self.buf]
        self.ev = [None, None]
        self.k = 0
        self._advance(1, n_frames * batches)
        torch.cuda.synchronize()

    def _advance(self, b, by):
        self.seq[b].add_(by)
        e = self.torch.cuda.Event()
        e.record()
        self.ev[b] = e

    def feed(self, bank):
        b = self.k & 1
        if self.ev[b] is not None:
            self.ev[b].synchronize()  # (recorded two calls ago)
        bank.process_adcdac_frames_device(self.buf[b].data_ptr(), self.fs, self.nf)
        self._advance(b, 2 * self.nf * self.batches)
        self.k += 1


def side_leg_frames(pkg, torch, device, seconds, n=4096, batches=22, log2_per_trace=24):
    """BASELINE configs[2]: 4-trace AdcDac ("dual-iir") frame stream, N=4096, frames resident in HBM
    (psdc_process_adcdac_frames_device).  Algorithmic bytes = n_frames x frame_size."""
    nframes = (1 << log2_per_trace) // (batches * 8)
    rng = np.random.default_rng(1)
    raw = np.clip(np.round(rng.standard_normal((4, nframes * batches * 8), dtype=np.float32) * 4096), -32768, 32767).astype(np.int16)
    data, fs = pkg.make_adcdac_frames(raw, batches)
    rep = FrameReplay(torch, np.frombuffer(data, dtype=np.uint8).copy(), fs, nframes, batches)
    bank = pkg.PsdCascadeBank(n, 4, device=device)
    for _ in range(3):
        rep.feed(bank)
    bank.sync()
    bank.configure(profile=True)
    t0 = time.perf_counter()
    calls = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(4):
            rep.feed(bank)
        calls += 4
    ns = bank.num_stages(0)
    for c in range(4):
        bank.psd(c)
    dt = time.perf_counter() - t0
    prof = bank.profile_read()
    loss = bank.loss()
    bank.close()
    kern_s = prof["kernel_ms"] * 1e-3
    nbytes = float(calls) * nframes * fs
    ach = nbytes / kern_s / 1e9 if kern_s > 0 else 0.0
    return {"value": calls * raw.size / dt / 1e6, "unit": "MS/s (samples of the four traces)",
            "workload": f"BASELINE configs[2]: 4-trace AdcDac frames ({batches} batches, {fs} B) resident in HBM, PsdCascade N={n}, "
                        f"{calls} calls of {nframes} frames (2^{log2_per_trace} samples per trace) replayed from two buffers with `seq` "
                        f"moving on, {ns} stages",
            "loss": loss,
            "roofline": {"bound": BOUND, "priced_against": "hbm", "binding": BINDING, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                         "end_to_end_frac": nbytes / dt / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_sample": fs / (batches * 8.0 * 4),
                         "launches": prof["launches"], "avg_launch_ms": prof["kernel_ms"] / max(1, prof["launches"]),
                         "note": "achieved = n_frames x frame_size / time of the dominant (fused) launches; end_to_end_frac over wall time"}}


def measured_traffic(kernel, n, channels, samples, window="hann"):
    """HBM bytes per one-span dominant launch from the latest committed PMC passes for this workload shape
    (profiles/*_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE of `bench.py --coalesce 1 --passes 1` under
    rocprofv3 --pmc); None if this shape was not measured."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("kernel") != kernel:
            continue
        if (d.get("fft_size") or 1024) != n or (d.get("channels") or 1) != channels:
            continue
        if (d.get("samples_per_launch") or (1 << 26)) != samples * channels:
            continue
        # the window decides the kernel variant (overlap 0 runs the two-segment form: other traffic): a record is of the window its
        # workload string names (the rectangular-window passes r04*_rect1024 matched the headline's shape otherwise)
        if ("rectangular" in (d.get("workload") or "")) != (window == "rectangular"):
            continue
        best = d
    return best


def launch_ranks(args, argv):
    """--gpus N > 1 without a launcher: start N ranks (one per GPU) under torch.distributed.run from this
    process, which has not initialised the GPU (no torch.cuda / HIP call so far), relay their output and
    exit with the launcher's code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    if args.dry_run_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(cmd, env=env)
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--passes", type=int, default=None,
                    help="passes of the cascade over the resident batch per step (default: 16 at 1 channel x 2^26, "
                         "8 at 8 channels x 2^24: 2^30 stream samples per GPU per step); the stream continues across passes")
    ap.add_argument("--clock-warm-ms", type=float, default=300.0,
                    help="untimed run of the same kernels on a scratch cascade before the warm-up steps, so that short "
                         "timed regions do not measure the GPU's clock ramp")
    ap.add_argument("--n", type=int, default=None, help="FFT size N (default 1024; 4096 with --workload frames)")
    ap.add_argument("--workload", default="raw", choices=["raw", "frames"],
                    help="raw: f32 streams resident in HBM (the headline).  frames: BASELINE configs[2] as the MAIN leg -- 4-trace AdcDac "
                         "frames (22 batches) resident in HBM, psdc_process_adcdac_frames_device, 2^log2-batch samples per trace per call "
                         "(default 24); for the profiling passes of tools/gpu.sh")
    ap.add_argument("--log2-batch", type=int, default=None,
                    help="samples per channel per pass = 2^this (default 26 at --gpus 1, 24 at --gpus > 1)")
    ap.add_argument("--channels-per-gpu", type=int, default=None, help="default 1 at --gpus 1, 8 at --gpus > 1 (config 4)")
    ap.add_argument("--detrend", default="none")
    ap.add_argument("--window", default="hann", choices=["hann", "rectangular", "hamming"],
                    help="hann (the configs'), rectangular (overlap 0: the fused kernels, two disjoint segments per transform) or a caller-built Hamming table with overlap N/2")
    ap.add_argument("--coalesce", type=int, default=None,
                    help="PSDC_OPT_COALESCE: in-place spans of a channel that share a round (library default 8; a handle of one channel 16, and more of spans shorter than 2^24 samples)")
    ap.add_argument("--eager", action="store_true", help="PSDC_OPT_EAGER: held spans go out when the device is seen idle (timing-dependent rounds; A/B aid)")
    ap.add_argument("--min-pairs", type=int, default=None, help="PSDC_OPT_MIN_PAIRS (library default 32 x teams per workgroup)")
    ap.add_argument("--avg", default=None, help="finite averaging 'limit,count' (AvgOpts, src/psd.rs:360-376); default: plain sum")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short side legs after the headline (configs[2] frames, configs[4] N=16384, 2^28-sample footprint)")
    ap.add_argument("--side-seconds", type=float, default=1.5, help="wall time of each side leg")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsal)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--dry-run-launch", action="store_true", help="print the rank launch command instead of running it")
    args = ap.parse_args()

    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world is None and (args.gpus > 1 or args.dry_run_launch):
        # no launcher around us: be the launcher (nothing has touched the GPU in this process)
        argv = [a for a in sys.argv[1:] if a != "--dry-run-launch"]
        raise SystemExit(launch_ranks(args, argv))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank number as {args.gpus} GPUs")
    multi = args.gpus > 1
    frames = args.workload == "frames"
    if frames and multi:
        raise SystemExit("--workload frames is a one-GPU leg")
    if args.n is None:
        args.n = 4096 if frames else 1024
    if frames:
        args.channels_per_gpu = 4
        if args.log2_batch is None:
            args.log2_batch = 24
    if args.channels_per_gpu is None:
        args.channels_per_gpu = 8 if multi else 1
    if args.log2_batch is None:
        args.log2_batch = 24 if multi else 26
    if args.passes is None:  # 2^30 stream samples per GPU per step
        args.passes = max(1, (1 << 30) // (args.channels_per_gpu << args.log2_batch))

    import torch
    torch.set_num_threads(1)  # no CPU tensor math here; keep torch's thread pool out of the timed loop

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the PSD path has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but only {torch.cuda.device_count()} GPUs visible "
                         f"(use --single-device --backend gloo to rehearse on one GPU)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("PSD_BENCH_FORCE_DIST"):  # (forced: a 1-rank group, to run the RCCL calls on one GPU)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    pkg = entry.load_package()
    n, C = args.n, args.channels_per_gpu
    T = 1 << args.log2_batch
    def window_arg():
        if args.window == "hann":
            return pkg.Window.HANN
        if args.window == "rectangular":
            return pkg.Window.RECTANGULAR
        i = np.arange(n, dtype=np.float64)  # a Window<N> built by the caller (src/psd.rs:12-20), overlap N/2
        w = (0.54 - 0.46 * np.cos(2 * np.pi * i / n)).astype(np.float32)
        m1, m2 = float(np.mean(w.astype(np.float64))), float(np.mean(w.astype(np.float64) ** 2))
        return pkg.WindowTable(w, np.float32(m1 * m1).item(), np.float32(m2 / (m1 * m1)).item(), n // 2)

    bank = pkg.PsdCascadeBank(n, C, device=local_rank, window=window_arg())
    bank.set_detrend(pkg.Detrend[args.detrend.upper()])
    if args.coalesce is not None:
        bank.configure(coalesce=args.coalesce)
    if args.eager:
        bank.configure(eager=True)
    if args.min_pairs is not None:
        bank.configure(min_pairs=args.min_pairs)
    if args.avg:
        lim, cnt = (int(v) for v in args.avg.split(","))
        bank.set_avg(pkg.AvgOpts(lim, cnt))
    # synthetic raw-f32 streams, generated on the device: channel c of rank r uses seed 0x7654321 + global channel
    bufs = []
    FR_BATCHES = 22
    if frames:  # 4 traces of T samples as AdcDac frames of 22 batches (1416 B), resident in HBM
        fr_n = T // (FR_BATCHES * 8)
        T = fr_n * FR_BATCHES * 8
        rng = np.random.default_rng(1)
        raw = np.clip(np.round(rng.standard_normal((4, T), dtype=np.float32) * 4096), -32768, 32767).astype(np.int16)
        fr_data, fr_size = pkg.make_adcdac_frames(raw, FR_BATCHES)
        fr_bytes = np.frombuffer(fr_data, dtype=np.uint8).copy()
        replays = {}
        del raw, fr_data
    else:
        for c in range(C):
            d = torch.empty(T, dtype=torch.float32, device="cuda")
            pkg.fill_noise_device(d.data_ptr(), T, seed=0x7654321 + rank * C + c, device=local_rank)
            bufs.append(d)
    torch.cuda.synchronize()

    P = args.passes

    def feed(b):
        if frames:  # (one replay per cascade: the scratch cascade of the clock warm-up has a stream of its own)
            if id(b) not in replays:
                replays[id(b)] = FrameReplay(torch, fr_bytes, fr_size, fr_n, FR_BATCHES)
            replays[id(b)].feed(b)
        else:
            for c in range(C):
                b.process_device(c, bufs[c].data_ptr(), T)

    def step():
        for _ in range(P):
            feed(bank)

    def barrier():
        if dist is not None:
            dist.barrier()
        bank.sync()
        torch.cuda.synchronize()

    from stabilizer_stream_amd import shard
    if args.clock_warm_ms > 0:  # a scratch cascade: the measured one sees exactly W + K steps
        scratch = pkg.PsdCascadeBank(n, C, device=local_rank, window=window_arg())
        scratch.set_detrend(pkg.Detrend[args.detrend.upper()])
        if args.coalesce is not None:
            scratch.configure(coalesce=args.coalesce)
        if args.eager:
            scratch.configure(eager=True)
        if args.min_pairs is not None:
            scratch.configure(min_pairs=args.min_pairs)
        scratch.configure(profile=True)  # rocprofv3 --stats sees these launches too: counted in *_whole_process
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < args.clock_warm_ms:
            for _ in range(64):
                feed(scratch)
            scratch.sync()
        prof_scratch = scratch.profile_read()
        scratch.close()
    else:
        prof_scratch = None
    if not os.environ.get("PSD_BENCH_NO_PROFILE"):  # (debug: what the event stamping itself costs)
        bank.configure(profile=True)  # HIP events around every dominant-kernel launch, on the library's stream
    for _ in range(args.warmup):
        step()
    if args.warmup or dist is not None:
        # first-use costs of the read-out path belong to the warm-up: pinned read-out buffers, and for N > 1
        # the gather's point-to-point connections, which RCCL sets up on the first call that uses them
        wrec = shard.pack_readout(bank, C, n, pkg)
        if dist is not None:
            shard.gather_readout(dist, wrec, device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
    barrier()
    prof0 = bank.profile_read()  # launches of the warm-up (kept: rocprofv3 --stats sees them too)
    dstate = DeviceState(torch, local_rank)
    dstate.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue_s = time.perf_counter() - t0  # host time to enqueue all steps (GPU still running)
    if os.environ.get("PSD_BENCH_DEBUG"):
        bank.sync()
        t_sync = time.perf_counter() - t0
    # read-out: every rank's raw per-stage accumulators + counters to rank 0 in ONE RCCL gather,
    # then the host stitch (PsdCascade::psd) per channel on rank 0
    ns = bank.num_stages(0)
    if os.environ.get("PSD_BENCH_DEBUG"):
        t_drain = time.perf_counter() - t0
        bank.read_channel(0)  # debug only: the C-ABI part of the read-out alone
        t_rc = time.perf_counter() - t0
    rec = shard.pack_readout(bank, C, n, pkg)
    if os.environ.get("PSD_BENCH_DEBUG"):
        t_pack = time.perf_counter() - t0
    if dist is not None:
        recs = shard.gather_readout(dist, rec, device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
    else:
        recs = [rec]
    merged = shard.stitch_gathered(pkg, recs, [C] * world) if rank == 0 else None
    if os.environ.get("PSD_BENCH_DEBUG"):
        t_read = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    dstate.__exit__()
    if os.environ.get("PSD_BENCH_DEBUG"):
        print(f"[debug] enqueue {host_enqueue_s*1e3:.2f} ms, +sync {t_sync*1e3:.2f} ms, +drain {t_drain*1e3:.2f} ms, +read_channel {t_rc*1e3:.2f} ms, "
              f"+pack {t_pack*1e3:.2f} ms, +readout {t_read*1e3:.2f} ms, "
              f"total {dt*1e3:.2f} ms", file=sys.stderr)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    prof_all = bank.profile_read()
    prof = {k: prof_all[k] - prof0[k] for k in prof_all}  # the timed region alone
    if prof_scratch is not None:
        prof_all = {k: prof_all[k] + prof_scratch[k] for k in prof_all}
    if rank == 0:
        total_samples = float(args.steps) * P * T * C * world
        msps = total_samples / dt / 1e6
        psd, breaks = merged[0]
        assert len(merged) == C * world
        reached = sum(1 for b in breaks if b.include)
        kern_s = prof["kernel_ms"] * 1e-3
        # algorithmic bytes per stage-0 sample: 4 for raw f32; frame bytes / samples for AdcDac frames
        alg_bps = (fr_size / (FR_BATCHES * 8.0 * 4.0)) if frames else ALG_BYTES_PER_SAMPLE
        ach = alg_bps * prof["stage0_samples"] / kern_s / 1e9 if kern_s > 0 else 0.0
        flop = ALG_FLOP_PER_SAMPLE.get(n, 5 * np.log2(n) + 18)
        kname = ("fused_kernel" if n in (256, 512, 1024) else "bigfused3_kernel" if n in (2048, 4096) else
                 "bigfused_kernel" if n in (8192, 16384) else "welch_kernel")
        tr = measured_traffic(kname, n, C, T, "rectangular" if args.window == "rectangular" else "hann")
        out = {
            "metric": "MS/s ingested (PsdCascade N=%d, %s)" % (n, "AdcDac frames, samples of the four traces" if frames else "raw f32"),
            "value": msps, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms_per_step": host_enqueue_s / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"BASELINE configs[2]: 4-trace AdcDac frames ({FR_BATCHES} batches, {fr_size} B) resident in HBM, read in place "
                                    f"(psdc_process_adcdac_frames_device): " if frames else "") +
                                   f"{'BASELINE configs[3]' if multi else 'BASELINE configs[1]'}: {C * world}-channel raw f32 stream "
                                   f"({C} per GPU), PsdCascade N={n}, {'Hann' if args.window == 'hann' else args.window}, detrend {args.detrend}, a step = {P} passes over "
                                   f"2^{args.log2_batch} samples/channel resident in HBM (the stream continues across passes), "
                                   f"{ns} stages instantiated ({reached} with count>=1)",
                       "fft_size": n, "channels": C * world, "channels_per_gpu": C,
                       "samples_per_pass_per_channel": T, "passes_per_step": P,
                       "samples_per_step_per_channel": T * P,
                       "algorithmic_bytes_per_sample": alg_bps,
                       "stages": ns, "parallelism": f"channel-shard x{world}"},
            "roofline": {"bound": BOUND, "priced_against": "hbm", "binding": BINDING, "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBPS,
                         "device_state": dstate.summary(),
                         # launches hold 1..PSDC_OPT_COALESCE spans of 2^log2_batch samples: both per-launch
                         # figures are for the AVERAGE launch of the timed region; the PMC traffic is measured
                         # per one-span launch (--coalesce 1) and scaled by the spans per launch
                         # NOT measured by this run: the PMC passes on record for this shape (profiles/*_traffic.json), scaled
                         "traffic_from_profiles": (tr["hbm_bytes_per_launch"] * prof["stage0_samples"] / max(1, prof["launches"]) / (T * C)
                                                   if tr else None),
                         "traffic": (tr["hbm_bytes_per_launch"] * prof["stage0_samples"] / max(1, prof["launches"]) / (T * C)
                                     if tr else None),  # = traffic_from_profiles (the contract's field name)
                         "traffic_over_algorithmic": (tr["hbm_bytes_per_launch"] / tr["algorithmic_bytes_per_launch"]) if tr else None,
                         # what the memory side actually moves: counter traffic / kernel time / peak (= frac x traffic_over_algorithmic)
                         "counter_traffic_frac": (ach / HBM_PEAK_GBPS * tr["hbm_bytes_per_launch"] / tr["algorithmic_bytes_per_launch"]) if tr else None,
                         "traffic_source": (tr["round"] + " PMC passes (one-span launches), profiles/") if tr else None,
                         "algorithmic_bytes_per_launch": alg_bps * prof["stage0_samples"] / max(1, prof["launches"]),
                         "spans_per_launch": prof["stage0_samples"] / max(1, prof["launches"]) / (T * C),
                         "kernel": kname, "launches": prof["launches"],
                         "avg_launch_ms": prof["kernel_ms"] / max(1, prof["launches"]),
                         "avg_launch_ms_whole_process": prof_all["kernel_ms"] / max(1, prof_all["launches"]),
                         "launches_whole_process": prof_all["launches"],
                         "algorithmic_bytes_per_sample": alg_bps},
            "compute_roofline": {"bound": "fp32_valu", "achieved": flop * msps * 1e6 / 1e12 / world,
                                 "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": flop * msps * 1e6 / 1e12 / world / FP32_VALU_PEAK_TFLOPS,
                                 "peak_is": "packed f32 FMA (v_pk_fma_f32) at 2.4 GHz; the kernels issue scalar f32 VALU",
                                 "scalar_fma_peak": FP32_SCALAR_FMA_PEAK_TFLOPS,
                                 "frac_of_scalar_fma_peak": flop * msps * 1e6 / 1e12 / world / FP32_SCALAR_FMA_PEAK_TFLOPS,
                                 "scalar_fma_peak_is": f"{SCALAR_FMA_LANES_PER_CLK_SIMD:g} lane-results/clk/SIMD sustained by v_fma_f32 at four "
                                                       "wavefronts per SIMD (tools/probes/valu_rate.cpp) x 1024 SIMDs x 2 flop x 2.4 GHz; "
                                                       "algorithmic flop count, so instruction overheads (DPP moves, address arithmetic, "
                                                       "non-fused adds: ~820 VALU instructions per pair and wavefront for ~670 flop-instructions) "
                                                       "and the sub-nominal clock both sit between this fraction and the VALU's busy fraction",
                                 "algorithmic_flop_per_sample": flop},
        }
        bank.close()
        bank = None
        if world == 1 and not args.no_other_configs and n == 1024 and C == 1 and not frames:
            # short legs of the other single-GPU configs, after (and outside) the headline's timed region
            del bufs
            torch.cuda.empty_cache()
            oc = {}
            oc["cfg3_frames_device"] = side_leg_frames(pkg, torch, local_rank, args.side_seconds)
            oc["cfg5_n16384"] = side_leg_raw(pkg, torch, 16384, 26, local_rank, args.side_seconds)
            oc["cfg3_size_n4096_raw"] = side_leg_raw(pkg, torch, 4096, 26, local_rank, args.side_seconds)
            # Window::rectangular() (src/psd.rs:24-32) at the headline's size: the fused kernels with two disjoint segments per transform
            oc["rectangular_n1024"] = side_leg_raw(pkg, torch, 1024, 26, local_rank, args.side_seconds, window=pkg.Window.RECTANGULAR)
            out["other_configs"] = oc
            # the headline shape beyond the 256 MiB Infinity Cache (FETCH_SIZE counts fabric requests, MALL hits included)
            out["hbm_honest"] = side_leg_raw(pkg, torch, 1024, 28, local_rank, args.side_seconds)
        if not args.no_cpu_baseline:
            if world == 1 and not frames:
                out["host_fed"] = host_fed_rate(pkg, n, local_rank)
                out["host_fed_small"] = host_fed_small(n, local_rank)
                out["device_fed_calls"] = device_fed_calls(n, local_rank)
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            out["cpu_baseline"] = cpu_baseline(n, args.cpu_seconds, threads=max(1, min(C * world, cores)))
        print(json.dumps(out))
    if bank is not None:
        bank.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
