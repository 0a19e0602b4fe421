#!/usr/bin/env python3
"""BASELINE configs[0] end to end: `gen sine | hilbert` on 2^24 float32 samples through
unix pipes -- the drop-in blocks (blocks/bin, GPU) beside the reference's own blocks
(oracle/_ref, CPU, built from /root/reference in the build container) on the same box.

    python tests/pipe_end_to_end.py [--log2n 24] [--procs P]

(kept under tests/ because it runs the reference binaries of oracle/_ref as the checker and
comparator; pytest does not collect it -- tests/test_blocks_gpu.py holds the small version.)

Prints one JSON object: wall time and Msamples/s of each pipeline (PCIe- and pipe-
inclusive: this is NOT bench.py's `value`), the reference `hilbert` alone from a
page-cached file on 1 and on P cores (independent processes: the reference's only form
of parallelism), and the parity of the two output streams (headers byte-exact, samples
within 1e-5 of the output RMS).
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root
BIN = os.path.join(ROOT, "blocks", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")
HDR = struct.Struct("<HHIQ")
GEN_ARGS = "freq=100k waveform=sine sample_rate=1M mode=real"


def sh(cmd):
    t0 = time.perf_counter()
    subprocess.run(cmd, shell=True, check=True, executable="/bin/bash")
    return time.perf_counter() - t0


def parse(path):
    """-> (headers as a list of tuples, all samples as one float32 array)"""
    buf = np.fromfile(path, np.uint8)
    hdrs, parts, off = [], [], 0
    while off + 16 <= len(buf):
        h = HDR.unpack_from(buf, off)
        hdrs.append(h)
        parts.append(buf[off + 16: off + 16 + h[2]])
        off += 16 + h[2]
    return hdrs, np.concatenate(parts).view(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=24)
    ap.add_argument("--procs", type=int, default=8)
    args = ap.parse_args()
    n = 1 << args.log2n
    B = 2044
    packets = -(-n // B)
    nbytes = packets * (16 + 4 * B)
    out = {"config": "gen %s | head -c %d | hilbert  (%d packets of %d samples)" % (GEN_ARGS, nbytes, packets, B),
           "samples": packets * B}
    have_ref = os.path.exists(os.path.join(REF, "hilbert"))
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        src, o_gpu, o_ref = (os.path.join(d, x) for x in ("in.bin", "gpu.bin", "ref.bin"))
        # -- ours: literal pipeline, then hilbert alone from the page cache
        sh("%s/gen %s | head -c %d > /dev/null" % (BIN, GEN_ARGS, 16 + 4 * B))   # first GPU touch (image page-in) not timed
        t = sh("%s/gen %s | head -c %d | %s/hilbert > %s" % (BIN, GEN_ARGS, nbytes, BIN, o_gpu))
        out["gpu_pipeline"] = {"seconds": t, "Msamples_per_s": packets * B / t / 1e6}
        sh("%s/gen %s | head -c %d > %s" % (BIN, GEN_ARGS, nbytes, src))
        t = min(sh("%s/hilbert < %s > /dev/null" % (BIN, src)) for _ in range(3))
        out["gpu_hilbert_from_file"] = {"seconds": t, "Msamples_per_s": packets * B / t / 1e6,
                                        "note": "process start + HIP init + H2D/D2H included"}
        if have_ref:
            t = sh("%s/gen %s | head -c %d | %s/hilbert > %s" % (REF, GEN_ARGS, nbytes, REF, o_ref))
            out["reference_pipeline"] = {"seconds": t, "Msamples_per_s": packets * B / t / 1e6, "cores": 2,
                                         "build": "g++ -std=gnu++14 -O3 -ffast-math -DNDEBUG (oracle/Makefile)"}
            t = sh("%s/hilbert < %s > /dev/null" % (REF, src))
            out["reference_hilbert_from_file"] = {"seconds": t, "Msamples_per_s": packets * B / t / 1e6, "cores": 1}
            P = args.procs
            t = sh(" & ".join("%s/hilbert < %s > /dev/null" % (REF, src) for _ in range(P)) + " & wait")
            out["reference_hilbert_%d_processes" % P] = {"seconds": t, "Msamples_per_s": P * packets * B / t / 1e6,
                                                         "cores": P}
            hg, sg = parse(o_gpu)
            hr, sr = parse(o_ref)
            rms = float(np.sqrt(np.mean(sr.astype(np.float64) ** 2)))
            out["parity"] = {"headers_equal": hg == hr, "packets": len(hr), "samples": int(len(sr)),
                             "expected_samples": packets * B - 63,
                             "max_abs_err_over_rms": float(np.max(np.abs(sg.astype(np.float64) - sr)) / rms)
                             if len(sg) == len(sr) else None}
        else:
            out["reference_pipeline"] = None
    print(json.dumps(out, indent=1))
    if have_ref:
        p = out["parity"]
        ok = p["headers_equal"] and p["samples"] == p["expected_samples"] and p["max_abs_err_over_rms"] is not None \
            and p["max_abs_err_over_rms"] <= 1e-5
        sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
