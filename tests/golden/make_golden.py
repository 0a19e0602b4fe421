#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Every fixture is data: inputs (seeded numpy or the reference `gen`) and the
bytes the reference binaries in oracle/_ref/ produced for them.  The command
line behind each fixture is recorded in manifest.json.  Nothing of the
reference's source text is stored.
"""
import json
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref")
HDR = struct.Struct("<HHIQ")
manifest = {}


def run(argv, stdin=b""):
    p = subprocess.run([os.path.join(REF, argv[0])] + argv[1:], input=stdin,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.stdout, p.stderr, p.returncode


def gen(args, nbytes):
    """`gen` never exits: bound it with head -c."""
    cmd = "%s %s | head -c %d" % (os.path.join(REF, "gen"), " ".join(args), nbytes)
    return subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout


def pkt(pid, content, payload, duration=0):
    return HDR.pack(pid, content, len(payload), duration) + payload


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)


def u8(b):
    return np.frombuffer(b, np.uint8)


# ---- (i) tap vectors ---------------------------------------------------------
taps = {}
for n in (2, 3, 4, 31, 127, 128, 255, 4095):
    out, _, _ = run(["ref_harness", "taps", "hilbert", str(n)])
    taps["hilbert_%d" % n] = np.frombuffer(out, np.float32)
designs = [("lowpass", 127, 0.1, 0.0), ("lowpass", 128, 0.25, 0.0), ("highpass", 63, 0.2, 0.0),
           ("bandpass", 4095, 0.1, 0.2), ("bandpass", 255, 0.05, 0.15), ("bandstop", 65, 0.1, 0.3)]
for kind, n, f1, f2 in designs:
    out, _, _ = run(["ref_harness", "taps", kind, str(n), repr(f1), repr(f2)])
    taps["%s_%d_%g_%g" % (kind, n, f1, f2)] = np.frombuffer(out, np.float32)
# the same designs under the other windows, and the windows themselves
for wname in ("rectangular", "hann", "hamming", "blackman_harris", "blackman"):
    for n in (9, 64, 127, 1000):
        out, _, _ = run(["ref_harness", "window", wname, str(n)])
        taps["window_%s_%d" % (wname, n)] = np.frombuffer(out, np.float32)
    if wname == "blackman":
        continue
    for kind, n, f1, f2 in (("lowpass", 127, 0.1, 0.0), ("highpass", 63, 0.2, 0.0), ("bandpass", 255, 0.05, 0.15),
                            ("bandstop", 65, 0.1, 0.3)):
        out, _, _ = run(["ref_harness", "taps", kind, str(n), repr(f1), repr(f2), wname])
        taps["%s_%d_%g_%g_%s" % (kind, n, f1, f2, wname)] = np.frombuffer(out, np.float32)
# the Kaiser window (one parameter, beta) and designs under it
for n, beta in ((9, 3.0), (64, 0.5), (127, 8.6), (1000, 14.0)):
    out, _, _ = run(["ref_harness", "window", "kaiser", str(n), repr(beta)])
    taps["window_kaiser%g_%d" % (beta, n)] = np.frombuffer(out, np.float32)
for kind, n, f1, f2, beta in (("lowpass", 127, 0.1, 0.0, 8.6), ("highpass", 63, 0.2, 0.0, 3.0),
                              ("bandpass", 255, 0.05, 0.15, 5.0), ("bandstop", 65, 0.1, 0.3, 0.5)):
    out, _, _ = run(["ref_harness", "taps", kind, str(n), repr(f1), repr(f2), "kaiser", repr(beta)])
    taps["%s_%d_%g_%g_kaiser%g" % (kind, n, f1, f2, beta)] = np.frombuffer(out, np.float32)
save("taps.npz", **taps)
manifest["taps.npz"] = ("ref_harness taps hilbert N ; ref_harness taps KIND N F1 F2 [WINDOW] (key = KIND_N_F1_F2[_WINDOW]) ; "
                        "ref_harness window NAME N [BETA] (key = window_NAME_N; kaiserBETA for the Kaiser window)")

# ---- (ii) gen ----------------------------------------------------------------
gens = {
    "real_sine_100k_1M": ["freq=100k", "waveform=sine", "sample_rate=1M", "mode=real"],
    "real_cosine_1k_48000_amp05_phi30": ["freq=1k", "sample_rate=48000", "mode=real", "amp=0.5", "phi=30"],
    "complex_cosine_100k_1M": ["freq=100k", "sample_rate=1M"],
    "complex_sine_1k_48000_amp05_phi30": ["freq=1k", "waveform=sine", "sample_rate=48000", "amp=0.5", "phi=30"],
    "real_sine_stream5_7k_250k": ["7k", "hz", "sine", "sample_rate=250k", "mode=real", "stream=5"],
    "real_square_100k_1M": ["freq=100k", "waveform=square", "sample_rate=1M", "mode=real"],
    "real_triangle_1k_48000_amp05_phi30": ["freq=1k", "waveform=triangle", "sample_rate=48000", "mode=real", "amp=0.5", "phi=30"],
    "real_sawtooth_7k_250k": ["freq=7k", "waveform=sawtooth", "sample_rate=250k", "mode=real"],
    "complex_square_100k_1M": ["freq=100k", "waveform=square", "sample_rate=1M"],
    "complex_triangle_1k_48000_amp05_phi30_taps31": ["freq=1k", "waveform=triangle", "sample_rate=48000", "amp=0.5", "phi=30", "hilbert_taps=31"],
    "complex_sawtooth_neg7k_250k": ["freq=-7k", "waveform=sawtooth", "sample_rate=250k"],
}
g = {}
for k, args in gens.items():
    g[k] = u8(gen(args, 3 * (16 + 8176)))
    manifest["gen.npz:" + k] = "gen " + " ".join(args) + " | head -c 24576"
save("gen.npz", **g)

# ---- (iii) hilbert block in/out streams ----------------------------------------
rng = np.random.default_rng(0x5D12)


def real_stream(counts, pid=0, dur=1000):
    return b"".join(pkt(pid, 6, rng.uniform(-1, 1, c).astype(np.float32).tobytes(), dur * c) for c in counts)


def cplx_stream(counts, pid=0, dur=1000):
    return b"".join(pkt(pid, 7, rng.uniform(-1, 1, 2 * c).astype(np.float32).tobytes(), dur * c) for c in counts)


hil = {}


def hilbert_case(name, stream, args):
    out, _, rc = run(["hilbert"] + args, stream)
    hil[name + "_in"] = u8(stream)
    hil[name + "_out"] = u8(out)
    manifest["hilbert.npz:" + name] = "hilbert " + " ".join(args) + " (rc %d)" % rc


hilbert_case("real_127", real_stream([500, 37, 700, 1, 300]), [])
hilbert_case("complex_127", cplx_stream([400, 64, 333]), [])
hilbert_case("real_128", real_stream([300, 300, 300]), ["0", "128"])
hilbert_case("real_255_nodelay", real_stream([300, 200, 400]), ["taps=255", "delay=false"])
hilbert_case("complex_31_stream3", cplx_stream([100, 100], pid=3), ["3", "31"])
# first packet shorter than the group delay: swallowed entirely
hilbert_case("real_4095_swallow", real_stream([2044, 2044, 1000]), ["taps=4095"])
# sine from the reference generator through default hilbert (config C1 in small)
hilbert_case("gen_sine_127", gen(gens["real_sine_100k_1M"], 4 * (16 + 8176)), [])
# truncated final packet (header promises 400 samples, 250 present)
trunc = real_stream([300, 400])
hilbert_case("real_127_truncated", trunc[:len(trunc) - 150 * 4], [])
# foreign packets interleaved (other id, other content); the reference emits a
# twin of each foreign packet (missing `continue`, blocks/hilbert.cpp:56-58)
foreign = (real_stream([200]) + pkt(9, 6, rng.uniform(-1, 1, 50).astype(np.float32).tobytes(), 5)
           + pkt(0, 1, b"hello world.....") + real_stream([150]) + pkt(2, 7, bytes(64)) + real_stream([90]))
hilbert_case("real_127_foreign", foreign, [])
save("hilbert.npz", **hil)

# ---- (iv) stream-filter truth table -------------------------------------------
pkts = []
tag = 0
for pid in (0, 1, 2):
    for content in (0, 6, 7):
        pkts.append(pkt(pid, content, struct.pack("<Q", tag)))
        tag += 1
sf_stream = b"".join(pkts)
sf_cases = [["pass"], ["drop"], ["pass", "1"], ["pass", "0,2"], ["pass", "1", "signal"],
            ["pass", "content=signal,complex_signal"], ["drop", "1"], ["drop", "0,2", "binary"],
            ["drop", "content=complex_signal"], ["mode=PASS", "stream=2", "content=Binary,Signal"],
            ["drop", "stream=1,2", "content=signal"]]
sf = {}
for args in sf_cases:
    out, err, rc = run(["stream-filter"] + args, sf_stream)
    tags = []
    p = 0
    while p < len(out):
        _, _, size, _ = HDR.unpack_from(out, p)
        tags.append(struct.unpack_from("<Q", out, p + 16)[0])
        p += 16 + size
    sf[" ".join(args)] = tags
with open(os.path.join(HERE, "stream_filter.json"), "w") as f:
    json.dump({"packets": [[pid, c] for pid in (0, 1, 2) for c in (0, 6, 7)], "forwarded": sf}, f, indent=1)
manifest["stream_filter.json"] = "stream-filter ARGS on 9 packets (id x content), payload = u64 tag; value = tags forwarded"

# ---- CLI error/usage text -----------------------------------------------------
cli = {}
for name, args in [("hilbert_help", ["hilbert", "help"]), ("gen_help", ["gen", "help"]),
                   ("stream_filter_help", ["stream-filter", "help"]),
                   ("stream_filter_nomode", ["stream-filter"]), ("gen_missing", ["gen"]),
                   ("hilbert_badid", ["hilbert", "70000"]), ("hilbert_badtaps", ["hilbert", "0", "abc"]),
                   ("stream_filter_badmode", ["stream-filter", "frob"])]:
    out, err, rc = run(args, b"")
    cli[name] = {"argv": args, "stderr": err.decode().replace(REF + "/", "./"), "rc": rc}
with open(os.path.join(HERE, "cli.json"), "w") as f:
    json.dump(cli, f, indent=1)
manifest["cli.json"] = "stderr + exit status of the reference blocks for usage/error invocations (argv[0] rewritten to ./NAME)"

# ---- (v) generic FIR through kfr::fir ------------------------------------------
fir = {}


def fir_case(name, kind, n, f1, f2, cplx, count, block):
    tp = taps["%s_%d_%g_%g" % (kind, n, f1, f2)]
    path = "/tmp/_golden_taps.bin"
    tp.tofile(path)
    x = rng.uniform(-1, 1, count * (2 if cplx else 1)).astype(np.float32)
    out, _, _ = run(["ref_harness", "fir", "complex" if cplx else "real", path, str(block)], x.tobytes())
    fir[name + "_x"] = x
    fir[name + "_y"] = np.frombuffer(out, np.float32)
    fir[name + "_taps"] = tp
    manifest["fir.npz:" + name] = "ref_harness fir %s <%s %d %g %g> %d" % (
        "complex" if cplx else "real", kind, n, f1, f2, block)


fir_case("lowpass127_complex", "lowpass", 127, 0.1, 0.0, True, 4096, 1000)
fir_case("lowpass127_real", "lowpass", 127, 0.1, 0.0, False, 4096, 333)
fir_case("bandpass4095_real", "bandpass", 4095, 0.1, 0.2, False, 8192, 2044)
fir_case("bandpass255_complex", "bandpass", 255, 0.05, 0.15, True, 2048, 100)
save("fir.npz", **fir)

# ---- (vi) host-only blocks: wrap, unwrap, inspect, throttle -----------------------
def run_file(argv, data):
    """stdin = a regular (seekable) file"""
    path = "/tmp/_golden_stdin.bin"
    with open(path, "wb") as f:
        f.write(data)
    with open(path, "rb") as f:
        p = subprocess.run([os.path.join(REF, argv[0])] + argv[1:], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return p.stdout, p.stderr, p.returncode


util, util_txt = {}, {}
urng = np.random.default_rng(0x0B10C5)
raw = urng.integers(0, 256, 30000, dtype=np.uint8).tobytes()
util["raw"] = u8(raw)
wrap_cases = {
    "wrap_file_count": (["signal", "3", "element_size=4", "element_count=1000", "sample_rate=1M"], raw[:10000], True),
    "wrap_pipe_count": (["signal", "3", "element_size=4", "element_count=1000", "sample_rate=1M"], raw[:10000], False),
    "wrap_pipe_exact": (["signal", "3", "element_size=4", "element_count=1000", "sample_rate=1M"], raw[:8000], False),
    "wrap_file_exact": (["signal", "3", "element_size=4", "element_count=1000", "sample_rate=1M"], raw[:8000], True),
    "wrap_file_optimal": (["complex_signal", "element_size=8", "sample_rate=48000"], raw, True),
    "wrap_file_duration": (["element_size=4", "duration=1000000", "sample_rate=1M"], raw[:10000], True),
    "wrap_file_norate": (["string", "65535", "element_size=1"], raw[:20000], True),
}
for k, (args, data, seekable) in wrap_cases.items():
    out, err, rc = (run_file if seekable else run)(["wrap"] + args, data)
    util[k] = u8(out)
    util_txt[k] = {"argv": args, "input_bytes": len(data), "seekable": seekable, "stderr": err.decode(), "rc": rc}
pstream = b"".join(pkt(pid, c, raw[1000 * i: 1000 * i + n], d) for i, (pid, c, n, d) in enumerate(
    [(0, 6, 400, 1000), (1, 7, 800, 2000), (2, 0, 13, 0), (1, 6, 0, 5), (0, 1, 7, 3000), (1, 7, 160, 4000)]))
util["pstream"] = u8(pstream)
for k, argv in {"unwrap_all": ["unwrap"], "unwrap_1": ["unwrap", "1"], "unwrap_7": ["unwrap", "stream=7"],
                "inspect_all": ["inspect"], "inspect_1": ["inspect", "1"], "inspect_1_pass": ["inspect", "1", "pass"],
                "inspect_1_pass_all": ["inspect", "1", "pass_all"], "inspect_all_pass_all": ["inspect", "pass_all=true"],
                "throttle_all": ["throttle"], "throttle_1": ["throttle", "1"]}.items():
    out, err, rc = run(argv, pstream)
    util[k] = u8(out)
    util_txt[k] = {"argv": argv, "stderr": err.decode(), "rc": rc}
for k, argv in {"wrap_help": ["wrap", "help"], "unwrap_help": ["unwrap", "help"], "inspect_help": ["inspect", "help"],
                "throttle_help": ["throttle", "help"], "wrap_missing": ["wrap"], "wrap_zero_size": ["wrap", "element_size=0"],
                "wrap_zero_count": ["wrap", "element_size=4", "element_count=0"],
                "wrap_short_duration": ["wrap", "element_size=4", "duration=10", "sample_rate=1M"],
                "wrap_mismatch": ["wrap", "element_size=4", "element_count=10", "duration=7", "sample_rate=1M"],
                "wrap_badid": ["wrap", "signal", "70000", "element_size=4"], "unwrap_badid": ["unwrap", "70000"]}.items():
    out, err, rc = run(argv, b"")
    util_txt[k] = {"argv": argv, "stderr": err.decode().replace(REF + "/", "./"), "rc": rc, "stdout_bytes": len(out)}
save("utility.npz", **util)
with open(os.path.join(HERE, "utility.json"), "w") as f:
    json.dump(util_txt, f, indent=1)
manifest["utility.npz / utility.json"] = "wrap / unwrap / inspect / throttle of the reference on the inputs stored beside the outputs (argv in utility.json)"

with open(os.path.join(HERE, "manifest.json"), "w") as f:
    json.dump(manifest, f, indent=1)
print("golden fixtures written:", sorted(os.listdir(HERE)))
