"""Pins the CPU oracle (oracle/sdr_oracle.c) to golden vectors produced by the
reference itself (tests/golden/make_golden.py ran the reference's own blocks
and a KFR harness compiled from /root/reference).  CPU only.

Tolerance: north_star's bar is max-abs-error <= 1e-5 of signal RMS (fp32); two
legal builds of the reference already differ by up to 7e-7 abs at T=127 and
3.3e-6 at T=4095, so the oracle (double accumulation)
is held to 2e-6*RMS at T<=255 and 6e-6*RMS at T=4095 against the reference.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O


def rel_err(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b).astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    rms = np.sqrt(np.mean(np.abs(b) ** 2))
    return np.max(np.abs(a - b)) / rms


@pytest.fixture(scope="module")
def taps(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "taps.npz")))


@pytest.mark.parametrize("n", [2, 3, 4, 31, 127, 128, 255, 4095])
def test_hilbert_taps(taps, n):
    ref = taps["hilbert_%d" % n]
    got = O.hilbert_taps(n)
    assert got.shape == ref.shape
    if n <= 4:
        # degenerate in the reference: the Blackman end points are exactly 0, so
        # gain = 0 and every tap is 0/0 = NaN.  The product
        # rejects taps < 5 instead; nothing to compare.
        assert np.isnan(ref).all()
        return
    # structure is exact: zeros where the reference has zeros
    assert np.array_equal(got == 0, ref == 0)
    assert np.max(np.abs(got - ref)) <= 2e-7 * np.max(np.abs(ref))


def split_window(name):
    """golden key suffix -> (window, parameter): 'kaiser8.6' -> ('kaiser', 8.6)"""
    if name.startswith("kaiser"):
        return "kaiser", float(name[len("kaiser"):])
    return name, 0.5


def test_fir_designs(taps):
    n_checked = 0
    for key, ref in taps.items():
        if key.startswith("hilbert") or key.startswith("window"):
            continue
        kind, n, f1, f2 = key.split("_")[:4]
        window, param = split_window("_".join(key.split("_")[4:]) or "blackman")
        got = O.fir_design(kind, int(n), float(f1), float(f2), window, param)
        # (Blackman designs to 5e-7; under the other windows KFR's own fp32 cosine shows:
        # it is off by up to 1e-6 near 3 pi / 2, the oracle uses libm's)
        tol = 5e-7 if window == "blackman" else 1e-6
        assert np.max(np.abs(got - ref)) <= tol * np.max(np.abs(ref)), key
        n_checked += 1
    assert n_checked == 6 + 4 * 4 + 4


def test_windows(taps):
    """The windows of the FIR designs (libs/kfr/dsp/window.hpp) against the reference's own."""
    n_checked = 0
    for key, ref in taps.items():
        if not key.startswith("window_"):
            continue
        name, n = key[len("window_"):].rsplit("_", 1)
        name, param = split_window(name)
        got = O.window(name, int(n), param)
        assert np.max(np.abs(got - ref)) <= 6e-7, key      # (KFR's fp32 cosine, see above)
        n_checked += 1
    assert n_checked == 5 * 4 + 4


def test_fir_vs_kfr(golden_dir):
    g = np.load(os.path.join(golden_dir, "fir.npz"))
    for name, cplx, tol in [("lowpass127_complex", True, 2e-6), ("lowpass127_real", False, 2e-6),
                            ("bandpass255_complex", True, 2e-6), ("bandpass4095_real", False, 6e-6)]:
        x, y, t = g[name + "_x"], g[name + "_y"], g[name + "_taps"]
        if cplx:
            x, y = x.view(np.complex64), y.view(np.complex64)
        for acc64, slack in ((True, 1.0), (False, 2.0)):
            f = O.Fir(t, cplx, acc64)
            # cut the stream differently from the reference run: results must
            # not depend on how the stream is split into calls
            parts = [f.process(x[:100]), f.process(x[100:101]), f.process(x[101:])]
            got = np.concatenate(parts)
            assert rel_err(got, y) <= tol * slack, (name, acc64)


def test_optimal_block_size():
    # probe values recorded in SURVEY.md 8a A6 / 8c
    assert O.optimal_block_size(4, 1000000) == 2044
    assert O.optimal_block_size(8, 1000000) == 1022
    assert O.optimal_block_size(4, 48000) == 2040
    assert O.optimal_block_size(4) == 2044


def _gen_payloads(buf):
    pk = O.parse_packets(bytes(buf))
    return pk


def test_gen_real(golden_dir):
    g = np.load(os.path.join(golden_dir, "gen.npz"))
    cases = [("real_sine_100k_1M", 1.0, 100e3, 1000000, 0.0, 1, 0),
             ("real_cosine_1k_48000_amp05_phi30", 0.5, 1e3, 48000, 30.0, 0, 0),
             ("real_sine_stream5_7k_250k", 1.0, 7e3, 250000, 0.0, 1, 5),
             ("real_square_100k_1M", 1.0, 100e3, 1000000, 0.0, 2, 0),
             ("real_triangle_1k_48000_amp05_phi30", 0.5, 1e3, 48000, 30.0, 3, 0),
             ("real_sawtooth_7k_250k", 1.0, 7e3, 250000, 0.0, 4, 0)]
    for key, amp, freq, sr, phase, wf, sid in cases:
        pk = _gen_payloads(g[key])
        B = O.optimal_block_size(4, sr)
        full = [p for p in pk if len(p[4]) == p[2]]
        assert all(p[0] == sid and p[1] == O.SIGNAL and p[2] == 4 * B for p in full)
        assert all(p[3] == B * 1000000000 // sr for p in full)
        ref = np.concatenate([np.frombuffer(p[4], np.float32) for p in full])
        cps = np.float32(np.float32(freq) / np.float32(sr))
        phi0 = np.float32(phase) / np.float32(360.0)
        phi0 = np.float32(phi0 - np.floor(phi0))
        got, _ = O.gen_real(len(full), B, amp, float(cps), float(phi0), wf)
        if wf == 2:
            # square: +-A exactly, except where fract(arg) sits within an ulp of a
            # switching point and -ffast-math contraction moves it across
            assert np.count_nonzero(got != ref) <= 2, key
            continue
        # the polynomial sine restated; <=2 ulp wiggle from -ffast-math contraction
        assert np.max(np.abs(got - ref)) <= 3e-7, key


def test_gen_complex(golden_dir):
    g = np.load(os.path.join(golden_dir, "gen.npz"))
    cases = [("complex_cosine_100k_1M", 1.0, 100e3, 1000000, 0.0, 0),
             ("complex_sine_1k_48000_amp05_phi30", 0.5, 1e3, 48000, 30.0, 1)]
    for key, amp, freq, sr, phase, wf in cases:
        pk = _gen_payloads(g[key])
        B = O.optimal_block_size(8, sr)
        full = [p for p in pk if len(p[4]) == p[2]]
        assert all(p[1] == O.COMPLEX_SIGNAL and p[2] == 8 * B for p in full)
        ref = np.concatenate([np.frombuffer(p[4], np.complex64) for p in full])
        cps = np.float32(np.float32(freq) / np.float32(sr))
        phi0 = np.float32(phase) / np.float32(360.0)
        phi0 = np.float32(phi0 - np.floor(phi0))
        got, _ = O.gen_complex(len(full), B, amp, float(cps), float(phi0), wf)
        assert np.max(np.abs(got - ref)) <= 5e-7, key


def test_gen_complex_quadrature_waveforms(golden_dir):
    """Complex square / triangle / sawtooth: the reference's second FIR call site
    (blocks/gen.cpp:231-260): waveform + (+-j) * Hilbert FIR of the advanced waveform."""
    g = np.load(os.path.join(golden_dir, "gen.npz"))
    cases = [("complex_square_100k_1M", 1.0, 100e3, 1000000, 0.0, 2, 127),
             ("complex_triangle_1k_48000_amp05_phi30_taps31", 0.5, 1e3, 48000, 30.0, 3, 31),
             ("complex_sawtooth_neg7k_250k", 1.0, -7e3, 250000, 0.0, 4, 127)]
    for key, amp, freq, sr, phase, wf, ht in cases:
        pk = _gen_payloads(g[key])
        B = O.optimal_block_size(8, sr)
        full = [p for p in pk if len(p[4]) == p[2]]
        assert len(full) >= 2 and all(p[1] == O.COMPLEX_SIGNAL and p[2] == 8 * B for p in full)
        ref = np.concatenate([np.frombuffer(p[4], np.complex64) for p in full])
        cps = np.float32(np.float32(freq) / np.float32(sr))
        phi0 = np.float32(phase) / np.float32(360.0)
        phi0 = np.float32(phi0 - np.floor(phi0))
        got, _ = O.gen_complex_quad(len(full), B, amp, float(cps), float(phi0), wf, ht)
        rms = np.sqrt(np.mean(np.abs(ref.astype(np.complex128)) ** 2))
        bad = np.abs(got - ref) > 2e-6 * rms
        # a square wave's switching instants can move by one sample where fract(arg) sits
        # within an ulp of 0 or 0.5 (-ffast-math contraction in the reference build); such a
        # flip changes that sample by 2A and the quadrature FIR output around it
        assert np.count_nonzero(bad) <= (4 * ht if wf == 2 else 0), (key, int(np.count_nonzero(bad)), float(np.abs(got - ref).max() / rms))


HILBERT_CASES = {
    "real_127": dict(), "complex_127": dict(), "real_128": dict(taps=128),
    "real_255_nodelay": dict(taps=255, delay=False), "complex_31_stream3": dict(stream_id=3, taps=31),
    "real_4095_swallow": dict(taps=4095), "gen_sine_127": dict(), "real_127_truncated": dict(),
}


@pytest.mark.parametrize("name", sorted(HILBERT_CASES))
def test_hilbert_block(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "hilbert.npz"))
    kw = HILBERT_CASES[name]
    got = O.parse_packets(O.hilbert_block(bytes(g[name + "_in"]), **kw))
    ref = O.parse_packets(bytes(g[name + "_out"]))
    assert len(got) == len(ref)
    tol = 6e-6 if kw.get("taps", 127) > 255 else 2e-6
    for a, b in zip(got, ref):
        assert a[:4] == b[:4]          # id, content, size, duration: exact
        dt = np.float32
        ya, yb = np.frombuffer(a[4], dt), np.frombuffer(b[4], dt)
        assert ya.shape == yb.shape
    ya = np.concatenate([np.frombuffer(a[4], np.float32) for a in got])
    yb = np.concatenate([np.frombuffer(b[4], np.float32) for b in ref])
    assert rel_err(ya, yb) <= tol


def test_hilbert_block_foreign_packets(golden_dir):
    """Reference quirk Q1 (blocks/hilbert.cpp:56-58, no `continue`): each foreign
    packet is followed by a twin with the same header and unspecified payload.
    The oracle's emulate_q1 mode reproduces the packet sequence; the default
    (intent) mode forwards each foreign packet exactly once."""
    g = np.load(os.path.join(golden_dir, "hilbert.npz"))
    inp = bytes(g["real_127_foreign_in"])
    ref = O.parse_packets(bytes(g["real_127_foreign_out"]))
    q1 = O.parse_packets(O.hilbert_block(inp, emulate_q1=True))
    assert [p[:4] for p in q1] == [p[:4] for p in ref]
    clean = O.parse_packets(O.hilbert_block(inp))
    foreign_in = [p for p in O.parse_packets(inp) if not (p[0] == 0 and p[1] in (6, 7))]
    assert len(ref) - len(clean) == len(foreign_in) == 3
    # the processed packets agree sample for sample, the forwarded ones byte for byte
    ref_wo_twins, seen = [], 0
    it = iter(ref)
    for p in it:
        ref_wo_twins.append(p)
        if not (p[0] == 0 and p[1] in (6, 7)):
            next(it)  # skip the twin
    assert len(ref_wo_twins) == len(clean)
    for a, b in zip(clean, ref_wo_twins):
        assert a[:4] == b[:4]
        if a[0] == 0 and a[1] == 6:
            assert np.max(np.abs(np.frombuffer(a[4], np.float32) - np.frombuffer(b[4], np.float32))) < 2e-6
        else:
            assert a[4] == b[4]


def test_stream_filter_truth_table(golden_dir):
    with open(os.path.join(golden_dir, "stream_filter.json")) as f:
        g = json.load(f)
    names = ["binary", "string", "time", "frequency", "wavelength", "sample_count", "signal",
             "complex_signal", "spectrum", "complex_spectrum"]
    for argline, tags in g["forwarded"].items():
        mode, ids, contents = None, None, None
        pos = 0
        for a in argline.split():
            if "=" in a:
                k, v = a.split("=", 1)
            else:
                k, v = ["mode", "stream", "content"][pos], a
                pos += 1
            if k == "mode":
                mode = v.lower()
            elif k == "stream":
                ids = [int(s) for s in v.split(",")]
            elif k == "content":
                contents = [names.index(s.lower()) for s in v.split(",")]
        got = [i for i, (pid, c) in enumerate(g["packets"])
               if O.stream_filter_forward(mode, ids, contents, pid, c)]
        assert got == tags, argline
