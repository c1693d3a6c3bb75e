#!/usr/bin/env python3
"""The reference's `stream_to_raw` binary (src/bin/stream_to_raw.rs) over this package's `Source`: one trace of a frame file (any of
the four payload formats) or of a raw file, written to stdout as native-endian f32 -- the format `--raw` reads back (src/source.rs:148-157).
usage: tools/stream_to_raw.py (--file FRAMES [--frame-size N] | --raw RAW) [--repeat] [--trace I] > out.raw
Host-side only (the decode of one frame per `get()`, as in the reference); no GPU."""
import argparse
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-f", "--file")
    ap.add_argument("--frame-size", type=int, default=8 + 30 * 2 * 6 * 4)  # src/source.rs:31
    ap.add_argument("--repeat", action="store_true")
    ap.add_argument("-r", "--raw")
    ap.add_argument("-t", "--trace", type=int, default=0)  # src/bin/stream_to_raw.rs:12-13
    ap.add_argument("--max-gets", type=int, default=None, help="stop after this many get() calls (the reference runs until an error)")
    a = ap.parse_args(argv)
    spec = importlib.util.spec_from_file_location("psdc_source", os.path.join(ROOT, "stabilizer-stream_amd", "source.py"))
    source = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(source)
    s = source.Source(source.SourceOpts(file=a.file, frame_size=a.frame_size, repeat=a.repeat, raw=a.raw))
    out = sys.stdout.buffer
    k = 0
    try:
        while a.max_gets is None or k < a.max_gets:
            t = s.get()[a.trace][1]  # &source.get()?[trace].1 (src/bin/stream_to_raw.rs:24): an index past the traces panics there
            if a.raw and s.eof:
                break  # (the reference spins on an exhausted raw file without --repeat, writing nothing)
            out.write(t.astype("<f4").tobytes())
            k += 1
    except EOFError:
        pass  # read_exact: UnexpectedEof ends the reference's loop with an error (src/source.rs:137-146)
    out.flush()
    return 0


if __name__ == "__main__":
    sys.exit(main())
