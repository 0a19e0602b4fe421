"""GPU tests of the drop-in blocks (blocks/bin/*) against the byte streams the
reference's own binaries produced for the same inputs (tests/golden/hilbert.npz,
gen.npz; made by tests/golden/make_golden.py) and against the oracle.

Headers (id, content, size, duration) must be byte-exact; samples within 1e-5 of
the RMS of the expected stream (north_star)."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "blocks", "bin")
TOL = 1e-5


def run_block(name, args, stdin=b"", env=None):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([os.path.join(BIN, name)] + args, input=stdin, capture_output=True, env=e, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


def samples(pkts):
    return np.concatenate([np.frombuffer(p[4], np.float32) for p in pkts]) if pkts else np.zeros(0, np.float32)


def rel_err(a, b):
    rms = np.sqrt(np.mean(b.astype(np.float64) ** 2))
    return float(np.max(np.abs(a.astype(np.float64) - b)) / rms)


HILBERT_CASES = {
    "real_127": [], "complex_127": [], "real_128": ["0", "128"],
    "real_255_nodelay": ["taps=255", "delay=false"], "complex_31_stream3": ["3", "31"],
    "real_4095_swallow": ["taps=4095"], "gen_sine_127": [], "real_127_truncated": [],
}


@pytest.mark.parametrize("batch", ["8388608", "65536"])
@pytest.mark.parametrize("name", sorted(HILBERT_CASES))
def test_hilbert_block_matches_reference_stream(golden_dir, name, batch):
    g = np.load(os.path.join(golden_dir, "hilbert.npz"))
    out = run_block("hilbert", HILBERT_CASES[name], bytes(g[name + "_in"]), {"SDR_AMD_BATCH_BYTES": batch})
    got, ref = O.parse_packets(out), O.parse_packets(bytes(g[name + "_out"]))
    assert [p[:4] for p in got] == [p[:4] for p in ref]      # framing byte-exact
    assert all(len(a[4]) == len(b[4]) for a, b in zip(got, ref))
    assert rel_err(samples(got), samples(ref)) <= TOL


@pytest.mark.parametrize("name", ["real_127", "complex_127", "real_4095_swallow"])
def test_hilbert_block_per_output_numerics(golden_dir, name):
    """SDR_AMD_NUMERICS=per-output keeps the block on the direct-form kernels (kfr::fir's own error
    character, include/sdr_amd.h: sdr_fir_create_numerics): the same stream as the reference's."""
    g = np.load(os.path.join(golden_dir, "hilbert.npz"))
    out = run_block("hilbert", HILBERT_CASES[name], bytes(g[name + "_in"]), {"SDR_AMD_NUMERICS": "per-output"})
    got, ref = O.parse_packets(out), O.parse_packets(bytes(g[name + "_out"]))
    assert [p[:4] for p in got] == [p[:4] for p in ref]
    assert rel_err(samples(got), samples(ref)) <= TOL
    e = dict(os.environ, SDR_AMD_NUMERICS="exact")
    p = subprocess.run([os.path.join(BIN, "hilbert")], input=bytes(g[name + "_in"]), capture_output=True, env=e, timeout=120)
    assert p.returncode != 0 and b"SDR_AMD_NUMERICS must be fast or per-output" in p.stderr


def test_hilbert_block_foreign_packets_once(golden_dir):
    """Interleaving kept, foreign packets forwarded byte-exact and ONCE (the
    reference's fall-through emits a garbage twin after each: SURVEY.md 8b Q1)."""
    g = np.load(os.path.join(golden_dir, "hilbert.npz"))
    inp = bytes(g["real_127_foreign_in"])
    out = run_block("hilbert", [], inp, {"SDR_AMD_BATCH_BYTES": "65536"})
    got = O.parse_packets(out)
    ref, it = [], iter(O.parse_packets(bytes(g["real_127_foreign_out"])))
    for p in it:
        ref.append(p)
        if not (p[0] == 0 and p[1] in (6, 7)):
            next(it)  # drop the reference's twin
    assert [p[:4] for p in got] == [p[:4] for p in ref]
    for a, b in zip(got, ref):
        if a[0] == 0 and a[1] == 6:
            assert np.max(np.abs(np.frombuffer(a[4], np.float32) - np.frombuffer(b[4], np.float32))) <= 2e-6
        else:
            assert a[4] == b[4]


def test_hilbert_block_mixed_content_shares_history():
    """One delay line per stream id whatever the content (real packets are widened
    to complex before kfr::fir, blocks/hilbert.cpp:60-68): a stream that switches
    Signal -> ComplexSignal -> Signal continues its history."""
    rng = np.random.default_rng(2)
    a = rng.uniform(-1, 1, 300).astype(np.float32)
    b = rng.uniform(-1, 1, 2 * 200).astype(np.float32)
    c = rng.uniform(-1, 1, 250).astype(np.float32)
    inp = O.pack_packet(0, 6, a.tobytes(), 300) + O.pack_packet(0, 7, b.tobytes(), 200) + O.pack_packet(0, 6, c.tobytes(), 250)
    got = O.parse_packets(run_block("hilbert", ["delay=false"], inp))
    want = O.parse_packets(O.hilbert_block(inp, delay=False))
    assert [p[:4] for p in got] == [p[:4] for p in want]
    assert rel_err(samples(got), samples(want)) <= TOL


def test_hilbert_block_packet_larger_than_a_batch():
    """The wire format allows payloads up to 4 GiB and the reference reads any packet whole:
    a packet bigger than the batch size gets a batch of its own, between ordinary packets and
    a forwarded one, with order and arithmetic unchanged."""
    rng = np.random.default_rng(5)
    sizes = [1000, 100000, 7, 30000]
    xs = [rng.uniform(-1, 1, n).astype(np.float32) for n in sizes]
    stream = b"".join([O.pack_packet(0, O.SIGNAL, xs[0].tobytes(), 1000),
                       O.pack_packet(9, 1, b"between", 5),
                       O.pack_packet(0, O.SIGNAL, xs[1].tobytes(), 100000),
                       O.pack_packet(0, O.SIGNAL, xs[2].tobytes(), 7),
                       O.pack_packet(0, O.SIGNAL, xs[3].tobytes(), 30000)])
    out = run_block("hilbert", [], stream, {"SDR_AMD_BATCH_BYTES": "65536"})
    got = O.parse_packets(out)
    ref = O.parse_packets(O.hilbert_block(stream))
    assert [p[:4] for p in got] == [p[:4] for p in ref]
    assert got[1][4] == b"between"
    a = samples([p for p in got if p[1] == O.SIGNAL])
    b = samples([p for p in ref if p[1] == O.SIGNAL])
    assert rel_err(a, b) <= TOL


def test_gen_block_matches_reference_packets(golden_dir):
    g = np.load(os.path.join(golden_dir, "gen.npz"))
    cases = {
        "real_sine_100k_1M": ["freq=100k", "waveform=sine", "sample_rate=1M", "mode=real"],
        "real_cosine_1k_48000_amp05_phi30": ["freq=1k", "sample_rate=48000", "mode=real", "amp=0.5", "phi=30"],
        "complex_cosine_100k_1M": ["freq=100k", "sample_rate=1M"],
        "complex_sine_1k_48000_amp05_phi30": ["freq=1k", "waveform=sine", "sample_rate=48000", "amp=0.5", "phi=30"],
        "real_sine_stream5_7k_250k": ["7k", "hz", "sine", "sample_rate=250k", "mode=real", "stream=5"],
    }
    for key, args in cases.items():
        ref = bytes(g[key])
        cmd = "%s %s | head -c %d" % (os.path.join(BIN, "gen"), " ".join(args), len(ref))
        out = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120).stdout
        assert len(out) == len(ref), key
        got, want = O.parse_packets(out), O.parse_packets(ref)
        assert [p[:4] for p in got] == [p[:4] for p in want], key
        # same fp32 recipe: real mode is the restated polynomial (<= 2 ulp from FMA-free
        # vs -ffast-math builds), complex mode accurate sincos of the same fp32 angle
        assert np.max(np.abs(samples(got) - samples(want))) <= 5e-7, key


def test_gen_block_other_waveforms_match_reference_packets(golden_dir):
    """Real square / triangle / sawtooth, and their complex forms (waveform + j * Hilbert
    FIR of the waveform: the reference's second FIR call site, blocks/gen.cpp:231-260),
    against the reference's own packets."""
    g = np.load(os.path.join(golden_dir, "gen.npz"))
    cases = {
        "real_square_100k_1M": ["freq=100k", "waveform=square", "sample_rate=1M", "mode=real"],
        "real_triangle_1k_48000_amp05_phi30": ["freq=1k", "waveform=triangle", "sample_rate=48000", "mode=real", "amp=0.5", "phi=30"],
        "real_sawtooth_7k_250k": ["freq=7k", "waveform=sawtooth", "sample_rate=250k", "mode=real"],
        "complex_square_100k_1M": ["freq=100k", "waveform=square", "sample_rate=1M"],
        "complex_triangle_1k_48000_amp05_phi30_taps31": ["freq=1k", "waveform=triangle", "sample_rate=48000", "amp=0.5", "phi=30", "hilbert_taps=31"],
        "complex_sawtooth_neg7k_250k": ["freq=-7k", "waveform=sawtooth", "sample_rate=250k"],
    }
    for key, args in cases.items():
        ref = bytes(g[key])
        cmd = "%s %s | head -c %d" % (os.path.join(BIN, "gen"), " ".join(args), len(ref))
        out = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120).stdout
        assert len(out) == len(ref), key
        got, want = O.parse_packets(out), O.parse_packets(ref)
        assert [p[:4] for p in got] == [p[:4] for p in want], key
        a, b = samples(got), samples(want)
        rms = np.sqrt(np.mean(b.astype(np.float64) ** 2))
        bad = np.abs(a - b) > 3e-6 * rms
        # a square wave's switching instant may move by one sample where fract(arg) sits
        # within an ulp of the threshold (the reference build contracts differently); in
        # complex mode such a flip also shows in the quadrature FIR output around it
        allowed = 0 if "square" not in key else (2 if key.startswith("real") else 2 * 4 * 127)
        assert np.count_nonzero(bad) <= allowed, (key, int(np.count_nonzero(bad)), float(np.abs(a - b).max() / rms))


def test_gen_block_frequency_control_input():
    """unit=stream (blocks/gen.cpp:61-81, 167-176, 187): Frequency packets of the named stream
    on stdin retune the generator at a packet boundary; the generator ends when the control
    input ends.  Every output packet must continue the phase recurrence with either the
    old or the new frequency, old before new -- and, like the reference, the switch happens on
    the FIRST packet generated after the control packet was read: the test lets the generator
    block on a full output pipe, sends the control packet, waits, and then counts the
    old-frequency packets that were already in the pipe (its capacity) plus the one blocked
    in write()."""
    import fcntl
    import time
    B, rate = 2044, 1000000
    pkt = 16 + 4 * B
    proc = subprocess.Popen([os.path.join(BIN, "gen"), "7", "stream", "sine", "sample_rate=1M", "mode=real", "amp=0.5", "phi=90"],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    proc.stdin.write(O.pack_packet(3, 3, np.float32(5e4).tobytes(), 0))      # foreign stream id: ignored
    proc.stdin.write(O.pack_packet(7, 6, np.zeros(4, np.float32).tobytes(), 0))  # not a frequency packet: ignored
    proc.stdin.flush()
    first = proc.stdout.read(8 * pkt)                                         # cps = 0 until told otherwise
    time.sleep(1.0)                                                           # the generator fills the pipe and blocks
    cap = fcntl.fcntl(proc.stdout.fileno(), 1032)                             # F_GETPIPE_SZ (the block widened it)
    proc.stdin.write(O.pack_packet(7, 3, np.float32(125e3).tobytes(), 0))     # Frequency, Hz
    proc.stdin.flush()
    time.sleep(1.0)                                                           # the control thread has read it by now
    more = proc.stdout.read(cap + 64 * pkt)
    proc.stdin.close()                                                        # control input ends: gen ends
    t0 = time.time()
    rest = b""
    while time.time() - t0 < 60:
        chunk = proc.stdout.read(1 << 20)
        if not chunk:
            break
        rest += chunk
    assert proc.wait(timeout=60) == 0
    pk = O.parse_packets(first + more + rest)
    pk = [p for p in pk if len(p[4]) == p[2]]
    assert len(pk) >= 72 and all(p[0] == 0 and p[1] == O.SIGNAL and p[2] == 4 * B for p in pk)
    f_new = float(np.float32(125e3) / np.float32(rate))
    phi, cps, switched = float(np.float32(90.0) / np.float32(360.0)), 0.0, None
    for k, p in enumerate(pk):
        got = np.frombuffer(p[4], np.float32)
        ref, phi_next = O.gen_real(1, B, 0.5, cps, phi, 1)
        if switched is None and np.max(np.abs(got - ref)) > 1e-6:
            cps, switched = f_new, k
            ref, phi_next = O.gen_real(1, B, 0.5, cps, phi, 1)
        assert np.max(np.abs(got - ref)) <= 1e-6, k
        phi = phi_next
    assert switched is not None and k >= 8
    # old-frequency packets after the 8 read up front: what the full pipe held, plus the packet
    # that was blocked in write() (generated before the control packet arrived) -- not one more
    in_pipe = -(-cap // pkt)
    assert in_pipe <= switched - 8 <= in_pipe + 1, (switched, cap)


def test_gen_pipe_hilbert_equals_reference_chain(golden_dir):
    """Config C1 in small: `gen sine | hilbert` through a real unix pipe, compared
    with the reference's `gen | hilbert` output for the first 4 packets."""
    g = np.load(os.path.join(golden_dir, "hilbert.npz"))
    ref = O.parse_packets(bytes(g["gen_sine_127_out"]))
    nbytes = 4 * (16 + 8176)
    cmd = "%s freq=100k waveform=sine sample_rate=1M mode=real | head -c %d | %s" % (
        os.path.join(BIN, "gen"), nbytes, os.path.join(BIN, "hilbert"))
    out = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120).stdout
    got = O.parse_packets(out)
    assert [p[:4] for p in got] == [p[:4] for p in ref]
    assert rel_err(samples(got), samples(ref)) <= TOL


def test_config1_full_size_pipe():
    """BASELINE configs[0] at its full size under the driver: `gen sine | head -c | hilbert` on
    2^24 float32 samples through real unix pipes (8208 whole packets of 2044 samples and a
    truncated 8209th), against the oracle's generator + hilbert block loop (C, a few seconds):
    headers byte-exact, N_out = 2^24 - 63, samples within 1e-5 of the output RMS.  No reference
    binary is needed here (profiles/r01_c1_pipe_2p24.json holds the run beside the reference's
    own blocks)."""
    n, B = 1 << 24, 2044
    whole, rest = divmod(n, B)
    nbytes = whole * (16 + 4 * B) + 16 + 4 * rest
    cmd = "%s freq=100k waveform=sine sample_rate=1M mode=real | head -c %d | %s" % (
        os.path.join(BIN, "gen"), nbytes, os.path.join(BIN, "hilbert"))
    out = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300).stdout
    cps = float(np.float32(100e3) / np.float32(1e6))      # convert_freq, include/sdr/options.hpp:120-141
    x, _ = O.gen_real(whole + 1, B, 1.0, cps, 0.0, 1)
    dur = B * 1000000000 // 1000000
    stream = b"".join(O.pack_packet(0, O.SIGNAL, x[k * B:(k + 1) * B].tobytes(), dur) for k in range(whole + 1))[:nbytes]
    want = O.parse_packets(O.hilbert_block(stream))
    got = O.parse_packets(out)
    assert len(got) == len(want) == whole + 1
    assert [p[:4] for p in got] == [p[:4] for p in want]
    sg, sw = samples(got), samples(want)
    assert len(sg) == len(sw) == n - 63
    assert rel_err(sg, sw) <= TOL


def test_stream_filter_fir_extension(golden_dir):
    """fir= extension: forwarded sample packets are filtered with kfr-style taps;
    compared with the reference's own kfr::fir output (tests/golden/fir.npz)."""
    g = np.load(os.path.join(golden_dir, "fir.npz"))
    x, y = g["lowpass127_complex_x"], g["lowpass127_complex_y"]
    cuts = [0, 1000, 1001, 3000, 4096]
    inp = b""
    for i in range(len(cuts) - 1):
        inp += O.pack_packet(4, 7, x[2 * cuts[i]:2 * cuts[i + 1]].tobytes(), 10)
        inp += O.pack_packet(9, 7, b"\0" * 16)          # dropped by the id filter
        inp += O.pack_packet(4, 1, b"note")              # forwarded untouched
    out = run_block("stream-filter", ["pass", "4", "fir=lowpass", "taps=127", "cutoff=0.1"], inp,
                    {"SDR_AMD_BATCH_BYTES": "65536"})
    got = O.parse_packets(out)
    assert [p[1] for p in got] == [7, 1] * 4 and all(p[0] == 4 for p in got)
    sig = samples([p for p in got if p[1] == 7])
    assert rel_err(sig, y) <= TOL


@pytest.mark.parametrize("kind,args,window", [
    ("lowpass", ["cutoff=0.1"], "hann"), ("bandpass", ["cutoff=0.05", "cutoff2=0.15"], "hamming"),
    ("highpass", ["cutoff=0.2"], "blackman_harris"), ("bandstop", ["cutoff=0.1", "cutoff2=0.3"], "rectangular"),
    ("lowpass", ["cutoff=0.12", "beta=8.6"], "kaiser")])
def test_stream_filter_fir_extension_windows(kind, args, window):
    """fir= under the other kfr windows (window=...): the block's output against the oracle's
    FIR run with the oracle's design for that window (both pinned to the reference's own taps
    in tests/golden/taps.npz)."""
    rng = np.random.default_rng(31)
    n, ntaps = 9000, 63
    x = rng.uniform(-1, 1, n).astype(np.float32)
    f = {"cutoff": 0.0, "cutoff2": 0.0, "beta": 0.5}
    f.update({a.split("=")[0]: float(a.split("=")[1]) for a in args})
    taps = O.fir_design(kind, ntaps, f["cutoff"], f["cutoff2"], window, f["beta"])
    inp = b"".join(O.pack_packet(0, O.SIGNAL, x[i:i + 1500].tobytes(), 1500) for i in range(0, n, 1500))
    out = run_block("stream-filter", ["pass", "fir=" + kind, "taps=%d" % ntaps, "window=" + window] + args, inp)
    got = O.parse_packets(out)
    assert [p[:3] for p in got] == [(0, O.SIGNAL, 6000)] * 6
    assert rel_err(samples(got), O.fir(taps, x)) <= TOL
    p = subprocess.run([os.path.join(BIN, "stream-filter"), "pass", "fir=lowpass", "cutoff=0.1", "window=gaussian"], input=b"",
                       capture_output=True)
    assert p.returncode == 255 and b"window" in p.stderr
