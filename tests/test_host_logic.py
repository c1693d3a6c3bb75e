"""CPU-only tests of the host side: C-ABI exports, bookkeeping planner, stitch, helpers,
and that the product path fails loudly (no CPU fallback) when there is no GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import has_gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol(pkg):
    hdr = open(os.path.join(ROOT, "include", "psdcascade.h")).read()
    declared = sorted(set(re.findall(r"\b(psdc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 25
    L = pkg.lib()
    for name in declared:
        assert hasattr(L, name), f"libpsdcascade.so does not export {name}"
    assert sorted(pkg.EXPORTS) == declared
    assert L.psdc_abi_version() == 3
    out = subprocess.run(["nm", "-D", "--defined-only", pkg.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (psdc_[a-z0-9_]+)", out))
    assert exported == set(declared)


def test_library_does_not_link_the_oracle(pkg):
    out = subprocess.run(["ldd", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", pkg.LIB_PATH], capture_output=True, text=True).stdout
    assert "ora_" not in syms
    for f in os.listdir(os.path.join(ROOT, "stabilizer-stream_amd", "csrc")):
        if not f.endswith((".h", ".hip", ".cpp", "Makefile")):
            continue
        src = open(os.path.join(ROOT, "stabilizer-stream_amd", "csrc", f), errors="ignore").read()
        assert "oracle/" not in src.replace("oracle/hbf_taps_oracle.h, then", "") and "ora_" not in src, f
    init = open(os.path.join(ROOT, "stabilizer-stream_amd", "__init__.py")).read()
    assert "oracle" not in init.lower()


@pytest.mark.skipif(has_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(pkg):
    with pytest.raises(pkg.PsdError) as e:
        pkg.PsdCascade(1024)
    assert e.value.code == pkg.ERR_DEVICE and "no CPU fallback" in str(e.value)
    with pytest.raises(pkg.PsdError):
        pkg.hbf_dec8(np.zeros(64, dtype=np.float32))


def test_host_programs(pkg):
    """fft_emul runs the device FFT code lane by lane; plan_check simulates the reference loop."""
    host = os.path.join(ROOT, "tests", "host")
    subprocess.run(["make", "-C", host], check=True, stdout=subprocess.DEVNULL)
    for prog in ("fft_emul", "plan_check"):
        r = subprocess.run([os.path.join(host, prog)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:]
    # the N=1024 frame exchange must be bank-conflict free
    out = subprocess.run([os.path.join(host, "fft_emul")], capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("N= 1024 rot=1")][0]
    m = re.search(r"read cycles (\d+) \(ideal (\d+)\)\s+write cycles (\d+) \(ideal (\d+)\)", line)
    assert m.group(1) == m.group(2) and m.group(3) == m.group(4), line


@pytest.mark.timeout(900)
def test_round_planner_on_the_cpu_under_sanitizers():
    """tests/host/round_plan_check: the library's host runtime (csrc/runtime.cpp, planner.cpp, frames_ingest.cpp, readout.cpp unchanged -- the round planner
    advance_round, staging, frame ingest, read-outs) linked with a host model of the HIP runtime and of the kernels
    (tests/host/sim/), built with -fsanitize=address,undefined.  The GPU fuzz campaigns' feeds are replayed; every address
    the planner hands to a kernel is touched as the real kernel touches it, and every segment and every decimator output
    of every stream must be produced exactly once (src/psd.rs:196-269).  tools/planner_mutations.sh shows that seeded
    planner bugs are caught."""
    host = os.path.join(ROOT, "tests", "host")
    subprocess.run(["make", "-C", host, "round_plan_check"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0")
    r = subprocess.run([os.path.join(host, "round_plan_check"), "1", "30"], capture_output=True, text=True, timeout=850, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "every segment and every decimator output exactly once" in r.stdout


def test_cpp_source_mirror(pkg, ora, tmp_path):
    """cpp/source.hpp, the C++ mirror of the reference's file-backed Source (src/source.rs:135-157): get() at the
    reference's granularity (512 raw samples / one frame per call, --repeat wrap), AdcDac decode on the host
    (src/de/data.rs:11-82), Loss counting (src/loss.rs:11-26), de::Error text -- tests/host/source_check.cpp, CPU only."""
    host = os.path.join(ROOT, "tests", "host")
    subprocess.run(["make", "-C", host, "source_check"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(host, "source_check"), str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]
    assert "3112 with --repeat" in r.stdout and "6 dropped" in r.stdout
    # Fls / ThermostatEem / Mpll (src/de/data.rs:84-212): the C++ decode of the program's pseudo-random frames against the oracle's,
    # bit for bit (the frames are rebuilt here with the same generator)
    import struct
    seen = 0
    for fmt, bb in ((2, 56), (3, 80), (4, 24)):
        lcg = (12345 * fmt) & 0xFFFFFFFF
        for k in range(2):
            pay = bytearray()
            for _ in range(bb * 5):
                lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
                pay.append(lcg >> 24)
            fr = bytes([0x7B, 0x05, fmt, 5]) + struct.pack("<I", 100 * fmt + 5 * k) + bytes(pay)
            st, f, seq, bat, tr = ora.frame_decode(fr)
            assert (st, f, seq, bat) == (0, fmt, 100 * fmt + 5 * k, 5)
            for c, (name, v) in enumerate(tr):
                want = f"fmt {fmt} frame {k} trace {c} [{name}]:" + "".join(f" {u:08x}" for u in v.view(np.uint32))
                assert want in r.stdout, want
                seen += 1
    assert seen == 2 * (4 + 4 + 3)


def test_cpp_mirror_builds_against_the_abi(pkg):
    """cpp/psd_cascade.hpp (the header-only C++ mirror of PsdCascade / Break / MergeOpts) compiles
    and links against include/psdcascade.h + libpsdcascade.so: tests/host/cpp_mirror_check.cpp is the
    reference's stream_test flow written with it (run on the GPU by test_gpu_parity)."""
    host = os.path.join(ROOT, "tests", "host")
    subprocess.run(["make", "-C", host, "cpp_mirror_check"], check=True, stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(host, "cpp_mirror_check"))


@pytest.mark.parametrize("n,window", [(16, "hann"), (64, "hann"), (512, "hann"), (1024, "hann"),
                                      (128, "rect"), (4096, "hann")])
def test_plan_counts_match_the_oracle(pkg, ora, n, window):
    rng = np.random.default_rng(n)
    wk = pkg.Window.HANN if window == "hann" else pkg.Window.RECTANGULAR
    for total in [0, 1, n - 1, n, n + 1, 3 * n, 17 * n + 5, 300 * n + 77, int(rng.integers(1, 700 * n))]:
        o = ora.PsdCascade(n, "f32", window=window)
        o.process(np.zeros(total, dtype=np.float32))
        plan = pkg.plan_counts(n, total, wk)
        assert len(plan) == o.num_stages, (n, total)
        for k, (recv, segs, pend) in enumerate(plan):
            info = o.stage_info(k)
            assert (segs, pend) == (info["count"], info["pending"]), (n, total, k)
    assert pkg.hbf_response_length(3) == ora.hbf_response_length(3) == 35


@pytest.mark.parametrize("n", [64, 1024])
def test_stitch_matches_the_oracle(pkg, ora, n):
    """psdc_stitch (host part of PsdCascade::psd, src/psd.rs:479-543) on the oracle's f32 stage data."""
    x = np.random.default_rng(5).standard_normal(700 * n).astype(np.float32)
    o = ora.PsdCascade(n, "f32")
    o.process(x)
    ns = o.num_stages
    infos = [o.stage_info(k) for k in range(ns)]
    spectra = np.stack([o.stage_spectrum(k) for k in range(ns)])
    for opts in (pkg.MergeOpts(), pkg.MergeOpts(True, 0, True), pkg.MergeOpts(False, 3, False),
                 pkg.MergeOpts(True, 1, False), pkg.MergeOpts(False, 10 ** 6, True)):
        p, br = pkg.stitch(n, [i["count"] for i in infos], [i["avg"] for i in infos],
                           [i["pending"] for i in infos], spectra, opts)
        pr, brr, cbr = o.psd(opts.keep_overlap, opts.min_count, opts.keep_transition_band)
        assert np.array_equal(p, pr, equal_nan=True)  # same f32 ops in the same order (count 0 with min_count 0 gives NaN, like the reference)
        assert len(br) == len(brr)
        for b, r in zip(br, brr):
            assert (b.start, b.include, b.count, b.avg, b.bins.start, b.bins.stop, b.fft_size,
                    b.decimation, b.pending, b.processed) == (
                r["start"], bool(r["include"]), r["count"], r["avg"], r["bins_start"], r["bins_end"],
                r["fft_size"], r["decimation"], r["pending"], r["processed"])
        assert np.array_equal(pkg.Break.frequencies(br), o.frequencies(cbr))
        if br and br[0].include:
            assert br[-1].rbw() == pytest.approx(1.0 / n) and br[0].effective_fft_size() == n * br[0].decimation


def test_var_and_noise_helpers(pkg, ora):
    p, f = [1000.0, 100.0, 1.2, 3.4, 5.6], [0.0, 1.0, 3.0, 6.0, 9.0]
    assert abs(pkg.var_eval(p, f, 2.7) - 0.13478442) < 1e-6  # src/var.rs:52-60
    assert pkg.var_eval(p, f, 2.7) == ora.var_eval(p, f, 2.7)
    assert pkg.var_eval(p, f, 0.3, x_exp=-4, sinx_exp=6, clip=1.0) == ora.var_eval(p, f, 0.3, -4, 6, 1.0)
    x = pkg.noise_host(1 << 16, 0x7654321)
    assert x.dtype == np.float32 and abs(float(x.mean())) < 10 / 256 and abs(float((x * x).mean()) - 1) < 10 / 256
    assert np.array_equal(pkg.noise_host(100, 5, 50), pkg.noise_host(150, 5)[50:])


def test_adcdac_frame_builder_roundtrip(pkg, ora):
    raw = np.random.default_rng(1).integers(-32768, 32768, size=(4, 8 * 5 * 7)).astype(np.int16)
    data, fs = pkg.make_adcdac_frames(raw, 5, seq0=0xFFFFFFFE)  # seq wraps (src/loss.rs wrapping_sub)
    assert fs == 8 + 64 * 5 and len(data) == 7 * fs
    got = [[] for _ in range(4)]
    for i in range(7):
        st, seq, nb, tr = ora.adcdac_decode(data[i * fs:(i + 1) * fs])
        assert st == 0 and nb == 5 and seq == (0xFFFFFFFE + 5 * i) % 2 ** 32
        for c in range(4):
            got[c].append(tr[c])
    lsb = np.float32(4.096) * np.float32(2.5) / np.float32(32768)
    assert np.array_equal(np.concatenate(got[0]), raw[0].astype(np.float32) * lsb)
    dac = ((raw[3].astype(np.int32) + 65536) % 65536 ^ 0x8000).astype(np.uint16).view(np.int16)
    assert np.array_equal(np.concatenate(got[3]), dac.astype(np.float32) * lsb)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with no launcher around it must start N ranks itself (from a parent that has
    not touched the GPU) and must never report a different world size as N GPUs."""
    import json
    import sys
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "7", "--dry-run-launch"], capture_output=True,
                       text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(bench) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7"]  # the ranks see the same arguments
    # a launcher that gives another world size than --gpus: refuse, non-zero
    r = subprocess.run([sys.executable, bench, "--gpus", "8"], capture_output=True, text=True,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    src = open(bench).read()
    assert src.index("launch_ranks(args, argv)") < src.index("import torch\n")  # launched before torch is imported


def test_header_is_plain_c(tmp_path):
    """include/psdcascade.h is the drop-in boundary for a C / Rust-FFI / cgo caller: it must compile as strict C99 and
    link against the library with a C compiler."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "psdcascade.h"\n'
                   "int main(void) { psdc_break b; psdc_stage_stat s; psdc_loss l; psdc_profile p; psdc_stage *st = 0;\n"
                   "  (void)b; (void)s; (void)l; (void)p; (void)st;\n"
                   "  return psdc_abi_version() == PSDC_ABI_VERSION && psdc_hbf_response_length(3) == 35 ? 0 : 1; }\n")
    exe = tmp_path / "hdr"
    lib_dir = os.path.join(ROOT, "stabilizer-stream_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(exe), "-L", lib_dir, "-lpsdcascade", "-L/opt/rocm/lib", "-lamdhip64",
                    f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    assert subprocess.run([str(exe)]).returncode == 0  # pure host entry points: no GPU needed


def test_window_tables_match_the_reference_construction(pkg, ora):
    """psdc_window_table = Window::hann() / Window::rectangular() (src/psd.rs:24-55) exactly as the f32 oracle
    restates them, weights and constants; nenbw * power = mean(w^2) (what gain() relies on, src/psd.rs:279-283)."""
    for n in (16, 512, 1024, 16384):
        for kind, name in ((pkg.Window.HANN, "hann"), (pkg.Window.RECTANGULAR, "rect")):
            t = pkg.WindowTable._kind(n, kind)
            w, p, e, ov = ora.window(n, name, "f32")
            assert np.array_equal(t.win, w) and (t.power, t.nenbw, t.overlap) == (p, e, ov)
            assert abs(t.nenbw * t.power - float(np.mean(t.win.astype(np.float64) ** 2))) < 1e-6
    with pytest.raises(pkg.PsdError):
        pkg.WindowTable._kind(512, 7)


def test_create_window_rejects_bad_tables_before_touching_a_device(pkg):
    """(N - overlap) % 8 != 0 panics in the reference at the first decimation (src/psd.rs:246-247), overlap >= N
    underflows `N - overlap`: both are PSDC_ERR_ARG at construction, GPU or not."""
    n = 256
    t = pkg.WindowTable.hann(n)
    for bad in (n, n + 8, n - 4, 3):
        with pytest.raises(pkg.PsdError) as e:
            pkg.PsdCascadeBank(n, window=pkg.WindowTable(t.win, t.power, t.nenbw, bad))
        assert "overlap" in str(e.value)
        with pytest.raises(pkg.PsdError) as e:
            pkg.Psd(n, pkg.WindowTable(t.win, t.power, t.nenbw, bad))
        assert "overlap" in str(e.value)
    with pytest.raises(pkg.PsdError) as e:
        pkg.Psd.new(n // 2, t)
    assert "fft.len()" in str(e.value)  # assert_eq!(N, fft.len()) src/psd.rs:139


def test_packed_record_roundtrip_and_64_bit_counts(pkg, ora):
    """psdc_pack_init / psdc_pack_channel / psdc_unpack_stitch (pure host): a record built from the oracle's stage
    data stitches bit-identically to psdc_stitch on the same data; with a count past 2^32 the record's 64-bit count
    normalises like single-GPU psd() (gain() of the 64-bit count) while the u32 view saturates."""
    n = 64
    x = np.random.default_rng(11).standard_normal(900 * n).astype(np.float32)
    chans = []
    for seed in range(3):
        o = ora.PsdCascade(n, "f32")
        o.process(np.roll(x, 17 * seed))
        infos = [o.stage_info(k) for k in range(o.num_stages)]
        chans.append(([i["count"] for i in infos], [i["avg"] for i in infos], [i["pending"] for i in infos],
                      np.stack([o.stage_spectrum(k) for k in range(o.num_stages)])))
    rec = pkg.pack_record(n, chans, rows=5)  # padded to the largest shard of a gather
    assert rec.size == pkg.readout_bytes(n, 5)
    for c, (counts, avgs, pend, sp) in enumerate(chans):
        assert pkg.unpack_info(rec, c) == (n, 5, len(counts))
        for opts in (pkg.MergeOpts(), pkg.MergeOpts(True, 0, True), pkg.MergeOpts(False, 3, False)):
            p, br = pkg.stitch(n, counts, avgs, pend, sp, opts)
            q, bq = pkg.unpack_stitch(rec, c, opts)
            assert np.array_equal(p, q, equal_nan=True) and br == bq
    assert pkg.unpack_info(rec, 4) == (n, 5, 0) and pkg.unpack_stitch(rec, 4)[0].size == 0  # padding channels are empty
    # psdc_pack_pad: the unpadded record widened as bytes is the padded record, bit for bit (what shard.pack_readout does
    # with a bank's own psdc_pack_readout record when shards are uneven)
    rec3 = pkg.pack_record(n, chans)
    assert pkg.unpack_info(rec3, 0)[1] == 3 and np.array_equal(pkg.pack_pad(rec3, 5), rec) and pkg.pack_pad(rec3, 3) is not None
    with pytest.raises(pkg.PsdError):
        pkg.pack_pad(rec, 3)  # cannot drop channels
    assert pkg.unpack_info(pkg.pack_pad(pkg.pack_record(n, []), 2), 1) == (n, 2, 0)  # a rank that owns no channel
    with pytest.raises(pkg.PsdError):
        pkg.unpack_stitch(rec[:100], 0)
    # past 2^32 segments
    counts, avgs, pend, sp = chans[0]
    big = [counts[0] + (1 << 33)] + list(counts[1:])
    rec2 = pkg.pack_record(n, [(big, avgs, pend, sp)])
    q, bq = pkg.unpack_stitch(rec2, 0)
    p64, b64 = pkg.stitch(n, big, avgs, pend, sp)  # routes through psdc_stitch_window (64-bit counts)
    assert np.array_equal(q, p64) and bq == b64
    top = bq[-1]  # stage 0 is the last break (highest rate)
    assert top.count == 0xFFFFFFFF  # the u32 the ABI reports saturates (the reference's wraps)
    sl = slice(top.start, top.start + len(top.bins))
    expect = sp[0][top.bins.start:top.bins.stop] / (np.float32((n // 2) * big[0]) * np.float32(1.5) * np.float32(0.25))
    assert np.allclose(q[sl], expect, rtol=1e-6)
    psat, _ = pkg.stitch(n, [0xFFFFFFFF] + list(counts[1:]), avgs, pend, sp)  # what a u32 gather would have given
    assert not np.allclose(psat[sl], q[sl], rtol=1e-3)


def test_packed_record_rejects_hostile_headers(pkg):
    """A packed read-out may arrive over any transport: psdc_unpack_* must hold every header field to the range the
    library can produce before it enters a size computation (the round-3 advisor's case: n = 2^31 with
    n_channels = 2^28 wraps `n_channels * bytes_per_channel` in 64 bits, the length test passes and the channel
    pointer leaves the buffer), never read out of bounds and never let an exception cross the ABI."""
    import struct
    n = 64
    sp = np.ones((2, n // 2 + 1), np.float32)
    rec = pkg.pack_record(n, [([9, 1], [0xFFFFFFFF] * 2, [3, 5], sp)], rows=2)
    assert pkg.unpack_info(rec, 0) == (n, 2, 2)

    def with_header(**kw):
        f = dict(zip(("magic", "version", "n", "n_channels", "power", "nenbw", "overlap", "kind"),
                     struct.unpack("<IIIIffII", rec[:32].tobytes())))
        f.update(kw)
        b = rec.copy()
        b[:32] = np.frombuffer(struct.pack("<IIIIffII", *f.values()), np.uint8)
        return b

    hostile = [
        with_header(n=1 << 31, n_channels=1 << 28),   # product wraps to a small number
        with_header(n=1 << 31, n_channels=2),
        with_header(n=0x80000040, n_channels=1 << 27),
        with_header(n=1 << 20),                        # beyond any FFT size the library has
        with_header(n=1),
        with_header(n_channels=0xFFFFFFFF),
        with_header(n_channels=4097),
        with_header(n_channels=3),                     # one more channel than the bytes hold
        with_header(overlap=n), with_header(overlap=0xFFFFFFFF),
        with_header(power=0.0), with_header(nenbw=float("nan")),
        with_header(magic=0x12345678), with_header(version=99),
        rec[:31], rec[:32], rec[: rec.size - 1],
    ]
    for b in hostile:
        for ch in (0, 1, 7, 0xFFFFFFFF):
            with pytest.raises(pkg.PsdError):
                pkg.unpack_info(b, ch)
            with pytest.raises(pkg.PsdError):
                pkg.unpack_stitch(b, ch)
    # a stage count beyond the slots of a channel is refused too (it indexes the stage table)
    b = rec.copy()
    b[32:36] = np.frombuffer(struct.pack("<I", 17), np.uint8)
    with pytest.raises(pkg.PsdError):
        pkg.unpack_stitch(b, 0)
    # the intact record still reads
    assert pkg.unpack_stitch(rec, 0)[0].size > 0
    for bad in ((1 << 31, 1 << 28), (64, 4097), (1 << 20, 1)):
        with pytest.raises(pkg.PsdError):
            pkg.pack_record(bad[0], [], rows=bad[1])


@pytest.mark.timeout(900)
def test_kernels_pass_the_machine_verifier():
    """`make verify` (csrc/Makefile): every kernel TU through the gfx950 backend with LLVM's machine verifier on after
    each pass.  It is the check that names the miscompile round 3 met in one kernel variant (si-form-memory-clauses left
    a truncated live subrange; the allocator then reused the VGPR of a live sample -- DESIGN.md section 4): a restructured
    kernel that trips it again fails HERE, on the CPU, instead of as wrong spectra in one template variant on the GPU."""
    csrc = os.path.join(ROOT, "stabilizer-stream_amd", "csrc")
    r = subprocess.run(["make", "-j8", "-C", csrc, "verify"], capture_output=True, text=True, timeout=880)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "every kernel TU clean" in r.stdout


def test_generic_kernels_contain_no_calls_and_no_packed_ops_where_scalar(tmp_path):
    """csrc/kernels.hip builds welch_kernel (every size) and the chirp-z kernel (M <= 256, M >= 8192) all-scalar through a per-kernel
    target attribute.  A lambda or `__syncthreads()` called directly in such a kernel does not carry the attribute and is then NOT
    inlined: the first form of this change had 42 real calls with scratch traffic inside the pair loop (N = 128: 61 GS/s instead of
    300).  Compiled to gfx950 assembly here (no GPU needed): no `s_swappc` anywhere in the TU, no `v_pk_*` f32 arithmetic in the
    kernels that are meant to be scalar, and packed ops present in the chirp-z kernels that keep them."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "stabilizer-stream_amd", "csrc")
    out = tmp_path / "kernels.s"
    r = subprocess.run([hipcc, "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                        os.path.join(csrc, "kernels.hip"), "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert "s_swappc" not in asm and "s_call" not in asm
    bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(_ZN4psdk\w+):.*?\n(.*?)s_endpgm", asm, re.M | re.S)}
    pk = lambda name: len(re.findall(r"v_pk_(add|mul|fma)_f32", bodies[name]))
    scalar = [k for k in bodies if "welch_kernelILi" in k or "welch_bluestein_kernel_scalar" in k]
    packed = [k for k in bodies if "welch_bluestein_kernelILi" in k]
    assert len(scalar) >= 22 + 10 and len(packed) == 10, (len(scalar), len(packed))
    for k in scalar:
        assert pk(k) == 0, (k, pk(k))
    assert all(pk(k) > 50 for k in packed if any(f"ILi{m}E" in k for m in (512, 1024, 2048, 4096)))


def test_bench_traffic_record_is_of_the_benched_window():
    """bench.py's `roofline.traffic` is taken from the committed PMC passes of the SAME shape (profiles/*_traffic.json): the record of
    the rectangular-window kernel (same kernel family, size, samples) must not stand in for the headline's, nor the other way round."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    hann = b.measured_traffic("fused_kernel", 1024, 1, 1 << 26)
    rect = b.measured_traffic("fused_kernel", 1024, 1, 1 << 26, "rectangular")
    assert hann and "Hann" in hann["workload"] and "rectangular" not in hann["workload"]
    assert rect and "rectangular" in rect["workload"]
    assert hann["ratio"] > rect["ratio"]  # (1.35 against 1.20: half the transforms, the same streams)


def test_binding_constants_match_the_header(pkg):
    """The Python mirror's option / window / detrend / status constants are the header's (include/psdcascade.h): a binding that drifts
    would configure the wrong thing silently (PSDC_OPT_EAGER = 5 and PSDC_OPT_MERGE = 6 are round 5's)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "psdcascade.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"^#define (PSDC_[A-Z_0-9]+) \(?(-?\d+)\)?", hdr, re.M)}
    for name, value in (("PSDC_OPT_QUANTUM", pkg.OPT_QUANTUM), ("PSDC_OPT_PROFILE", pkg.OPT_PROFILE), ("PSDC_OPT_COALESCE", pkg.OPT_COALESCE),
                        ("PSDC_OPT_MIN_PAIRS", pkg.OPT_MIN_PAIRS), ("PSDC_OPT_EAGER", pkg.OPT_EAGER), ("PSDC_OPT_MERGE", pkg.OPT_MERGE)):
        assert defs[name] == value, name
    assert len({defs[k] for k in defs if k.startswith("PSDC_OPT_")}) == len([k for k in defs if k.startswith("PSDC_OPT_")])  # no two options share a number
    assert defs["PSDC_WINDOW_HANN"] == int(pkg.Window.HANN) and defs["PSDC_WINDOW_RECTANGULAR"] == int(pkg.Window.RECTANGULAR)
    for d in ("NONE", "MIDPOINT", "SPAN", "MEAN"):
        assert defs[f"PSDC_DETREND_{d}"] == int(pkg.Detrend[d])
    assert defs["PSDC_ABI_VERSION"] == pkg.lib().psdc_abi_version()
