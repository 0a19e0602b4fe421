"""CPU-only tests of the product's host side: the C ABI library loads and exports
every symbol include/sdr_amd.h declares, the tap designers match the reference's
golden taps, the blocks' argv grammar / usage / error text equals the reference's
(tests/golden/cli.json, captured from the reference binaries), stream-filter's
routing equals the reference's truth table, and compute entry points fail loudly
without a GPU (no CPU fallback)."""
import ctypes
import json
import os
import re
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import sdr_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "blocks", "bin")
HDR = struct.Struct("<HHIQ")


def _has_gpu():
    return sdr_amd.device_count() > 0


def test_capi_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "sdr_amd.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = set(re.findall(r"\b(sdr_[a-z_0-9]+)\s*\(", text))
    assert len(names) >= 20
    L = ctypes.CDLL(sdr_amd.lib_path())
    for n in sorted(names):
        assert hasattr(L, n), "libsdr_amd.so does not export %s" % n


def test_version_and_error_text():
    assert "gfx950" in sdr_amd.version()
    with pytest.raises(sdr_amd.SdrError) as e:
        sdr_amd.hilbert(4)
    assert e.value.code == -1 and "5 taps" in str(e.value)


def test_tap_designers_match_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "taps.npz"))
    for key in g.files:
        ref = g[key]
        parts = key.split("_")
        if parts[0] == "hilbert":
            n = int(parts[1])
            if n < 5:
                continue
            got = sdr_amd.hilbert(n)
            assert np.array_equal(got == 0, ref == 0), key
        elif parts[0] == "window":
            continue        # (the windows themselves are pinned on the oracle; the library exports only designs)
        else:
            window = "_".join(parts[4:]) or "blackman"
            beta = 0.5
            if window.startswith("kaiser"):
                window, beta = "kaiser", float(window[len("kaiser"):])
            got = sdr_amd.fir_design(parts[0], int(parts[1]), float(parts[2]), float(parts[3]), window, beta)
            # the library's taps are the oracle's, bit for bit (same fp32 recipe, same libm)
            assert np.array_equal(got, O.fir_design(parts[0], int(parts[1]), float(parts[2]), float(parts[3]), window, beta)), key
        # (other windows: KFR's own fp32 cosine is off by up to 1e-6 near 3 pi / 2)
        tol = 5e-7 if len(parts) <= 4 else 1e-6
        assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)), key
    with pytest.raises(sdr_amd.SdrError):
        sdr_amd._capi._check(sdr_amd.lib().sdr_fir_design_window(0, 9, 0.5, sdr_amd._capi._fp(np.zeros(8, np.float32)), 8, 0.1, 0.0))


def test_optimal_block_size():
    L = sdr_amd.lib()
    L.sdr_optimal_block_size.restype = ctypes.c_size_t
    L.sdr_optimal_block_size.argtypes = [ctypes.c_uint64, ctypes.c_uint64]
    assert L.sdr_optimal_block_size(4, 1000000) == 2044
    assert L.sdr_optimal_block_size(8, 1000000) == 1022
    assert L.sdr_optimal_block_size(4, 48000) == 2040
    assert L.sdr_optimal_block_size(4, 0) == 2044


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    with pytest.raises(sdr_amd.SdrError) as e:
        sdr_amd.FirState(sdr_amd.hilbert(127))
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    p = subprocess.run([os.path.join(BIN, "hilbert")], input=b"", capture_output=True)
    assert p.returncode == 255 and b"no HIP device" in p.stderr


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing that ships -- the Python package, the HIP
    library's sources, the blocks, the public header -- may import, link, open or name it."""
    import glob
    import re
    files = []
    for pat in ("sdr_amd/**/*.py", "sdr_amd/csrc/*", "blocks/*.cpp", "blocks/host/*", "blocks/Makefile", "include/*.h"):
        files += [f for f in glob.glob(os.path.join(ROOT, pat), recursive=True) if os.path.isfile(f)]
    assert len(files) > 15
    for f in files:
        text = open(f, errors="replace").read()
        assert not re.search(r"oracle|sdro_|_ref/", text), f
    # and the built library links nothing of it
    out = subprocess.run(["ldd", sdr_amd._capi.lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out


def run_block(name, args, stdin=b""):
    # argv[0] as the reference fixtures have it: ./NAME
    return subprocess.run(["./" + name] + args, input=stdin, capture_output=True, cwd=BIN)


def test_cli_usage_and_errors_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "cli.json")) as f:
        cases = json.load(f)
    for name, c in cases.items():
        p = run_block(c["argv"][0], c["argv"][1:])
        got = p.stderr.decode()
        if name == "stream_filter_help":
            got = got.split("\n")[0] + "\n"   # the extension line follows the reference's usage line
        assert got == c["stderr"], name
        assert p.returncode == c["rc"], name


def test_cli_number_grammar():
    # SI suffixes, fixed point, hex; errors as the reference words them
    for arg, msg in [("taps=1.5", "unsigned integer value expected"), ("taps=1.k", "digits expected after point"),
                     ("taps=12q", "invalid unit 'q'"), ("taps=-3", "unsigned integer value expected"),
                     ("taps=0x1k", "unsigned integer value expected"), ("delay=maybe", "boolean value expected")]:
        p = run_block("hilbert", ["0", arg])
        assert p.returncode == 255 and p.stderr.decode() == "error: %s: %s\n" % (arg.split("=")[0], msg), arg
    p = run_block("gen", ["freq=1x", "sample_rate=1M"])
    assert p.stderr.decode() == "error: freq: invalid unit 'x'\n"
    # 70k > 65535 is parsed (SI suffix) and then rejected as a stream id
    p = run_block("hilbert", ["70k"])
    assert p.stderr.decode() == "error: gen: 70000 is not a valid stream id\n"
    p = run_block("hilbert", ["0x10000"])
    assert p.stderr.decode() == "error: gen: 65536 is not a valid stream id\n"
    # unknown key=value pairs are ignored, extra positionals too
    p = run_block("stream-filter", ["pass", "1", "signal", "extra", "bogus=1"], b"")
    assert p.returncode == 0 and p.stderr == b""


def test_stream_filter_truth_table(golden_dir):
    with open(os.path.join(golden_dir, "stream_filter.json")) as f:
        g = json.load(f)
    stream = b"".join(HDR.pack(pid, c, 8, 0) + struct.pack("<Q", i) for i, (pid, c) in enumerate(g["packets"]))
    for argline, tags in g["forwarded"].items():
        p = run_block("stream-filter", argline.split(), stream)
        assert p.returncode == 0, argline
        out, got, pos = p.stdout, [], 0
        while pos < len(out):
            pid, c, size, dur = HDR.unpack_from(out, pos)
            tag = struct.unpack_from("<Q", out, pos + 16)[0]
            assert (pid, c) == tuple(g["packets"][tag]) and size == 8
            got.append(tag)
            pos += 16 + size
        assert got == tags, argline


def test_stream_filter_truncated_and_empty_input():
    # clean EOF on a short header; a truncated payload is forwarded as far as it came
    p = run_block("stream-filter", ["pass"], b"\x01\x02\x03")
    assert p.returncode == 0 and p.stdout == b""
    pkt = HDR.pack(1, 0, 100, 7) + b"x" * 40
    p = run_block("stream-filter", ["pass"], pkt)
    assert p.stdout == pkt


def test_stream_filter_routing_in_dribbles_and_bulk():
    """The routing loop parses headers in place in whatever pieces the pipe delivers: a stream
    fed 7 bytes at a time (headers split across reads), one fed in bulk with payloads of every
    size from empty to larger than a read, and a truncated tail; against the predicate of
    the oracle's restatement of blocks/stream_filter.cpp:59-86."""
    import random
    rng = random.Random(5)
    sizes = [0, 1, 15, 16, 17, 8176, 70000, 3, 0, 0, 5000000, 8, 4096] + [rng.randrange(0, 3000) for _ in range(300)]
    packets = []
    for i, n in enumerate(sizes):
        pid, content = rng.randrange(0, 6), rng.randrange(0, 10)
        packets.append((pid, content, HDR.pack(pid, content, n, i) + bytes([i % 251]) * n))
    stream = b"".join(p[2] for p in packets)
    for args, mode, ids, contents in (
            (["pass", "1,3"], "pass", {1, 3}, None), (["drop", "2", "signal,binary"], "drop", {2}, {6, 0}),
            (["pass", "content=complex_signal"], "pass", None, {7}), (["drop"], "drop", None, None), (["pass"], "pass", None, None)):
        want = b"".join(p[2] for p in packets if O.stream_filter_forward(mode, ids, contents, p[0], p[1]))
        got = run_block("stream-filter", args, stream)
        assert got.returncode == 0 and got.stdout == want, args
        # the same cut short inside a payload, and inside a header
        for cut in (len(stream) - 700, len(packets[0][2]) + 5):
            want_cut = b""
            pos = 0
            for p in packets:
                part = stream[pos:min(cut, pos + len(p[2]))]
                pos += len(p[2])
                if len(part) >= 16 and O.stream_filter_forward(mode, ids, contents, p[0], p[1]):
                    want_cut += part
                if pos >= cut:
                    break
            assert run_block("stream-filter", args, stream[:cut]).stdout == want_cut, (args, cut)
    # 7 bytes at a time through a real pipe
    small = b"".join(p[2] for p in packets[13:60])
    want = b"".join(p[2] for p in packets[13:60] if O.stream_filter_forward("pass", {1, 3}, None, p[0], p[1]))
    proc = subprocess.Popen(["./stream-filter", "pass", "1,3"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, cwd=BIN)
    import threading
    out = []
    t = threading.Thread(target=lambda: out.append(proc.stdout.read()))
    t.start()
    for i in range(0, len(small), 7):
        proc.stdin.write(small[i:i + 7])
        proc.stdin.flush()
    proc.stdin.close()
    t.join()
    assert proc.wait() == 0 and out[0] == want


@pytest.mark.parametrize("stdin_kind,stdout_kind", [("pipe", "pipe"), ("file", "pipe"), ("pipe", "file"), ("file", "file")])
def test_stream_filter_zero_copy_pass_and_drop(tmp_path, stdin_kind, stdout_kind):
    """Large payloads never enter user space (the reference's Source::pass / drop,
    lib/sdr/stream.cpp:219-296): what lies beyond the bytes already read is spliced when an end is a
    pipe, sent with sendfile from a regular file, and a dropped payload is seeked over or spliced
    into /dev/null.  Output bytes are checked against the routing predicate; the mechanism against
    the block's own byte counters (SDR_AMD_ROUTE_STATS)."""
    import re
    rng = np.random.default_rng(11)
    sizes = [8176, 6 << 20, 100, 9 << 20, 8176, 3 << 20, 12 << 20, 40]
    ids = [1, 1, 2, 2, 1, 2, 1, 1]
    pk = [HDR.pack(i, 6, n, k) + rng.integers(0, 256, n, dtype=np.uint8).tobytes() for k, (i, n) in enumerate(zip(ids, sizes))]
    stream = b"".join(pk)
    want = b"".join(p for p, i in zip(pk, ids) if i == 1)
    src = tmp_path / "in.bin"
    src.write_bytes(stream)
    dst = tmp_path / "out.bin"
    env = dict(os.environ, SDR_AMD_ROUTE_STATS="1")
    fin = open(src, "rb") if stdin_kind == "file" else subprocess.PIPE
    fout = open(dst, "wb") if stdout_kind == "file" else subprocess.PIPE
    proc = subprocess.Popen([os.path.join(BIN, "stream-filter"), "pass", "1"], stdin=fin, stdout=fout, stderr=subprocess.PIPE, env=env)
    import threading
    got = []
    readers = [threading.Thread(target=lambda: got.append(proc.stdout.read()))] if stdout_kind == "pipe" else []
    err = []
    readers.append(threading.Thread(target=lambda: err.append(proc.stderr.read())))
    for t in readers:
        t.start()
    if stdin_kind == "pipe":
        proc.stdin.write(stream)
        proc.stdin.close()
    assert proc.wait(timeout=120) == 0
    for t in readers:
        t.join()
    if stdout_kind == "file":
        fout.close()
        out = dst.read_bytes()
    else:
        out = got[0]
    if stdin_kind == "file":
        fin.close()
    assert out == want
    m = re.search(r"route: spliced=(\d+) sendfile=(\d+) copied=(\d+) skipped=(\d+)", err[0].decode())
    assert m, err[0]
    spliced, sent, copied, skipped = map(int, m.groups())
    big_pass = (6 << 20) + (12 << 20)          # forwarded payload bytes in packets larger than a read
    big_drop = (9 << 20) + (3 << 20)
    assert copied == 0
    if stdin_kind == "pipe" or stdout_kind == "pipe":
        assert spliced >= big_pass - (5 << 20) * 2 and sent == 0   # all but what the first reads already held
    else:
        assert sent >= big_pass - (5 << 20) * 2 and spliced == 0
    assert skipped >= big_drop - (5 << 20) * 2


def test_phase_chain_closed_form_is_the_fp32_recurrence():
    """sdr_capi.hip gen_next_table / sdr_gen_advance: when the block phase and its increment are whole
    multiples of 2^-23 below 1, the serial fp32 recurrence phi' = fract(phi + incr) (gen.cpp:185-227) never
    rounds and equals ((P + k I) mod 2^23) / 2^23 bit for bit -- the form the host then uses."""
    rng = np.random.default_rng(23)
    f = np.float32
    cases = [(0, 0), (0, 1), (1, (1 << 23) - 1), ((1 << 23) - 1, (1 << 23) - 1), (12345, 1 << 22), (0, 3355456)]
    cases += [(int(rng.integers(0, 1 << 23)), int(rng.integers(0, 1 << 23))) for _ in range(60)]
    n = 3000
    for P, I in cases:
        phi, incr = f(P) / f(1 << 23), f(I) / f(1 << 23)
        serial = np.empty(n, np.float32)
        p = phi
        for b in range(n):
            serial[b] = p
            s = f(p + incr)
            p = f(s - f(1)) if s >= 1 else s
        closed = ((P + np.arange(n, dtype=np.uint64) * I) & 0x7FFFFF).astype(np.float32) * f(1.0 / (1 << 23))
        assert np.array_equal(serial.view(np.uint32), closed.view(np.uint32)), (P, I)
