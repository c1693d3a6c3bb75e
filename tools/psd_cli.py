#!/usr/bin/env python3
"""The receiver loop of the reference's `psd` binary (src/bin/psd.rs:158-221) without its GUI: a file-backed `Source` feeds one
`PsdCascade<512>` per trace (`:174-182`), and when the source is exhausted every trace is read out as `Cmd::Send` does (`:191-216`):
`psd(&merge_opts)`, `Break::frequencies`, `Trace::plot` (integrated RMS and plot points, `:125-157`).  Same option names and defaults
as `SourceOpts` (file-backed subset, src/source.rs:16-48) and `AcqOpts` (src/bin/psd.rs:31-72).
usage: tools/psd_cli.py (--file FRAMES [--frame-size N] | --raw RAW) [AcqOpts ...] [--max-bytes B] [--csv DIR]
Prints one line per trace: name, stages, averages of the top stage, bins, integrated RMS; --csv writes DIR/<trace>.csv (the plot points)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-f", "--file")
    ap.add_argument("--frame-size", type=int, default=8 + 30 * 2 * 6 * 4)   # src/source.rs:31
    ap.add_argument("--repeat", action="store_true")
    ap.add_argument("-r", "--raw")
    ap.add_argument("-d", "--detrend", default="mean", choices=["none", "midpoint", "span", "mean"])  # src/bin/psd.rs:34-35
    ap.add_argument("--fs", type=float, default=1.0)
    ap.add_argument("--avg-max", type=int, default=1000)
    ap.add_argument("--avg-min", type=int, default=1)
    ap.add_argument("-a", "--avg", type=int, default=0xFFFFFFFF)
    ap.add_argument("--keep-overlap", action="store_true")
    ap.add_argument("--keep-transition-band", action="store_true")
    ap.add_argument("--integrate", action="store_true")
    ap.add_argument("--integral-start", type=float, default=1e-6)
    ap.add_argument("--integral-end", type=float, default=0.5)
    ap.add_argument("--max-bytes", type=int, default=None, help="stop after this many input bytes (needed with --repeat)")
    ap.add_argument("--csv", default=None, help="directory for the plot points of every trace")
    a = ap.parse_args(argv)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    from stabilizer_stream_amd import source
    integral_start, integral_end = a.integral_start * a.fs, a.integral_end * a.fs  # src/bin/psd.rs:161-162
    src = source.Source(source.SourceOpts(file=a.file, frame_size=a.frame_size, repeat=a.repeat, raw=a.raw), pkg)
    if a.raw:
        names = ["raw"]
    else:  # the labels of Payload::traces for the file's format (its first frame's header, src/de/frame.rs:25-37)
        with open(a.file, "rb") as f:
            head = f.read(8)
        if len(head) < 8 or head[0] != 0x7B or head[1] != 0x05 or not 1 <= head[2] <= 4:
            raise SystemExit("source: Invalid frame header")
        names = list(pkg.TRACE_NAMES[pkg.Format(head[2])])
    bank = pkg.PsdCascadeBank(1 << 9, len(names))  # PsdCascade::<{ 1 << 9 }> (src/bin/psd.rs:176)
    bank.set_detrend(pkg.Detrend[a.detrend.upper()])
    bank.set_avg(pkg.AvgOpts(limit=max(0, a.avg_max - 1), count=max(0, a.avg - 1)))  # AcqOpts::avg_opts (:74-79)
    merge = pkg.MergeOpts(keep_overlap=a.keep_overlap, min_count=a.avg_min, keep_transition_band=a.keep_transition_band)  # :81-87
    total = 0
    while a.max_bytes is None or total < a.max_bytes:
        got = src.feed(bank, max_bytes=(64 << 20) if a.max_bytes is None else min(64 << 20, max(a.frame_size, a.max_bytes - total)))
        if got == 0:
            break
        total += got
    src.close()
    if a.csv:
        os.makedirs(a.csv, exist_ok=True)
    for i, name in enumerate(names):
        if bank.num_stages(i) == 0:
            print(f"{name}: no samples")
            continue
        psd, breaks = bank.psd(i, merge)
        freqs = pkg.Break.frequencies(breaks)
        rms, xy = pkg.trace_plot(psd, freqs, fs=a.fs, integrate=a.integrate, integral_start=integral_start, integral_end=integral_end)
        top = bank.stage_info(i, 0)["count"]
        print(f"{name}: stages {bank.num_stages(i)} top-stage averages {top} bins {psd.size} breaks {len(breaks)} rms {rms:.9g}")
        if a.csv:
            safe = "".join(ch if ch.isalnum() else "_" for ch in name)
            with open(os.path.join(a.csv, safe + ".csv"), "w") as f:
                f.writelines(f"{x:.9g},{y:.9g}\n" for x, y in xy)
    loss = bank.loss()
    if not a.raw:
        tot = loss["received"] + loss["dropped"]
        print(f"loss: {loss['dropped']} of {tot} batches ({(loss['dropped'] / tot if loss['received'] else 0.0):.3e})")  # Loss::analyze
    bank.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
