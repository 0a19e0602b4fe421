"""GPU parity tests: the HIP FIR path, called through the C ABI
(include/sdr_amd.h via sdr_amd._capi), against the CPU oracle and against the
golden vectors produced by the reference's own kfr::fir.

Tolerance (north_star): max-abs-error <= 1e-5 of the RMS of the expected
output, fp32.  The oracle accumulates in double, so the error measured here is
the HIP kernel's own fp32 rounding (~3e-7 at 127 taps).
"""
import os

import numpy as np
import pytest

import oracle_lib as O
import sdr_amd

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel_err(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b).astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    rms = np.sqrt(np.mean(np.abs(b) ** 2))
    return float(np.max(np.abs(a - b)) / rms)


def rand(rng, n, cplx):
    if cplx:
        return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    return rng.uniform(-1, 1, n).astype(np.float32)


def taps_for(kind, n):
    if kind == "hilbert":
        return sdr_amd.hilbert(n)
    if kind == "lowpass":
        return sdr_amd.fir_lowpass(n, 0.1)
    if kind == "bandpass":
        return sdr_amd.fir_bandpass(n, 0.1, 0.2)
    if kind == "random":
        return np.random.default_rng(n).uniform(-1, 1, n).astype(np.float32)
    raise ValueError(kind)


CASES = [
    ("lowpass", 127, True), ("lowpass", 127, False), ("lowpass", 128, True), ("lowpass", 255, False),
    ("lowpass", 31, True), ("random", 5, False), ("random", 1, True), ("random", 2, False), ("random", 16, True),
    ("lowpass", 1000, False),
    ("hilbert", 127, True), ("hilbert", 127, False), ("hilbert", 128, False), ("hilbert", 255, True),
    ("hilbert", 125, False), ("hilbert", 125, True), ("hilbert", 31, False), ("hilbert", 5, True),
    ("bandpass", 4095, False), ("bandpass", 4095, True),
]


@pytest.mark.parametrize("path", ["direct", "fft"])
@pytest.mark.parametrize("kind,ntaps,cplx", [("lowpass", 127, True), ("lowpass", 127, False), ("hilbert", 127, False),
                                             ("hilbert", 128, True), ("lowpass", 255, True), ("random", 16, False),
                                             ("random", 2, True), ("lowpass", 1000, False), ("bandpass", 4095, True)])
def test_both_kernel_families(path, kind, ntaps, cplx):
    """Every tap count runs on both the direct-form and the overlap-save FFT kernels, pinned
    through the library's internal create hook (sdr_amd/csrc/internal.h) -- the release library
    has no environment switch."""
    rng = np.random.default_rng(ntaps + 3 * cplx)
    taps = taps_for(kind, ntaps)
    n = 33333
    x = rand(rng, n, cplx)
    st = sdr_amd.FirState(taps, cplx, path=path)
    assert ("fft" in st.kernel_name) == (path == "fft"), st.kernel_name
    cuts = [0, 1, 4000, 4001, 12288, 20000, n]   # tile-sized, tiny and ragged calls
    y = np.concatenate([sdr_amd.fir(st, x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
    assert rel_err(y, O.fir(taps, x)) <= TOL


@pytest.mark.parametrize("kind,ntaps,cplx", [("lowpass", 127, True), ("lowpass", 127, False), ("hilbert", 127, False),
                                             ("hilbert", 127, True), ("hilbert", 255, False), ("hilbert", 255, True),
                                             ("lowpass", 65, True), ("lowpass", 49, False), ("lowpass", 33, True),
                                             ("random", 16, False), ("random", 2, True), ("random", 1, False),
                                             ("random", 97, True), ("random", 113, False), ("random", 129, True)])
def test_matrix_core_kernels(kind, ntaps, cplx):
    """fir_mfma (the Toeplitz form on the fp32 matrix cores): dense filters up to 129 taps and
    every-other-tap (Hilbert) designs up to 257, real and complex, stream cut at tile-sized, tiny and
    ragged places, two channels with a padded stride."""
    rng = np.random.default_rng(ntaps + 5 * cplx)
    taps = taps_for(kind, ntaps)
    n = 33333
    x = rand(rng, n, cplx)
    st = sdr_amd.FirState(taps, cplx, path="mfma")
    assert "mfma" in st.kernel_name, st.kernel_name
    cuts = [0, 1, 4000, 4001, 12288, 20000, n]
    y = np.concatenate([sdr_amd.fir(st, x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
    assert rel_err(y, O.fir(taps, x)) <= TOL
    x2 = np.stack([rand(rng, 9001, cplx) for _ in range(2)])
    st2 = sdr_amd.FirState(taps, cplx, channels=2, path="mfma")
    y2 = np.concatenate([sdr_amd.fir(st2, x2[:, :5000]), sdr_amd.fir(st2, x2[:, 5000:])], axis=1)
    for c in range(2):
        assert rel_err(y2[c], O.fir(taps, x2[c])) <= TOL, c


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("ntaps", [130, 193, 194, 300, 385, 386, 700, 769, 770])
def test_4096_point_tiles_halo_classes_in_between(ntaps, cplx):
    """Halos of 192, 384 and 768 samples (fir_fft.hip: fir_fft_plan) between the power-of-two classes, at their first
    and last tap counts and one past (the next class): a stream of many tiles cut at tile-sized, tiny and ragged
    places, then the same filter on three channels with a padded stride, and on a Hilbert-shaped design through
    the pair view (kept taps decide the class there)."""
    rng = np.random.default_rng(ntaps + 7 * cplx)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    n = 150001
    x = rand(rng, n, cplx)
    st = sdr_amd.FirState(taps, cplx, path="fft")
    assert "fir_fft4096" in st.kernel_name, st.kernel_name
    cuts = [0, 1, 4000, 4001, 12288, 70000, n]
    y = np.concatenate([sdr_amd.fir(st, x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
    assert rel_err(y, O.fir(taps, x)) <= TOL
    C, m = 3, 40001
    x2 = np.stack([rand(rng, m, cplx) for _ in range(C)])
    st2 = sdr_amd.FirState(taps, cplx, channels=C, path="fft")
    y2 = np.concatenate([sdr_amd.fir(st2, x2[:, :9000]), sdr_amd.fir(st2, x2[:, 9000:])], axis=1)
    for c in range(C):
        assert rel_err(y2[c], O.fir(taps, x2[c])) <= TOL, c
    if not cplx:
        h = sdr_amd.hilbert(2 * ntaps - 1)      # ntaps kept taps
        st3 = sdr_amd.FirState(h, False, path="fft")
        xr = rand(rng, n, False)
        y3 = np.concatenate([sdr_amd.fir(st3, xr[:50001]), sdr_amd.fir(st3, xr[50001:])])
        assert rel_err(y3, O.fir(h, xr)) <= TOL, st3.kernel_name


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("ntaps", [1026, 1537, 2049, 2050, 4095, 4097, 4098, 6143, 8191, 8192])
def test_16384_point_tiles_for_long_filters(ntaps, cplx):
    """fir_fft16k (one 16384-point tile holds a filter of 1026..8193 taps: a halo of two to eight rows of 1024 samples,
    one spectrum, no partitions; fir_fft16k.hip) pinned through the internal hook, complex samples and real ones (two stretches of the
    stream as re / im of one transform): a stream long enough for many big tiles, cut so that calls start and end
    inside, at and between tile boundaries (the 2048-sample tiles at both ends of a call go to the 4096-point
    partitioned form), then three channels with a padded stride; and the same stream through the 4096-point kernels
    alone agrees."""
    rng = np.random.default_rng(ntaps)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    n = 400001 if cplx else 700001
    x = rand(rng, n, cplx)
    st = sdr_amd.FirState(taps, cplx, path="fft16k")
    assert "fir_fft16k" in st.kernel_name and ("COMPLEX" if cplx else "REAL") in st.kernel_name, st.kernel_name
    cuts = [0, 5, 20000, 180000, 180001, 300000, n]
    y = np.concatenate([sdr_amd.fir(st, x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
    want = O.fir(taps, x)
    assert rel_err(y, want) <= TOL
    st4 = sdr_amd.FirState(taps, cplx, path="fft4k")
    assert "fir_fft4096" in st4.kernel_name
    assert rel_err(sdr_amd.fir(st4, x), want) <= TOL
    C, m = 3, 70001 if cplx else 140001
    x2 = np.stack([rand(rng, m, cplx) for _ in range(C)])
    st2 = sdr_amd.FirState(taps, cplx, channels=C, path="fft16k")
    y2 = np.concatenate([sdr_amd.fir(st2, x2[:, :41000]), sdr_amd.fir(st2, x2[:, 41000:])], axis=1)
    for c in range(C):
        assert rel_err(y2[c], O.fir(taps, x2[c])) <= TOL, c


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("ntaps", [8194, 9001, 16383, 16385, 24577, 40001])
def test_16384_point_tiles_partitioned(ntaps, cplx):
    """More than 8193 taps on the 16384-point kernel: partitions of 8192 taps, hop 8192 (fir_fft16k<.., LONG16>), the
    older windows re-read and re-transformed.  Pieces that start inside the filter's reach (edge tiles with several
    windows in the history), whole tiles, one sample; then the library's own choice on a long call agrees."""
    rng = np.random.default_rng(ntaps + 3 * cplx)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    P = 8192 * (1 if cplx else 2)
    sizes = [5000, 3 * P, 1, 150000 if cplx else 300000, P - 1, 2 * P + 1]
    x = rand(rng, sum(sizes), cplx)
    want = O.fir(taps, x)
    st = sdr_amd.FirState(taps, cplx, path="fft16k")
    assert "fir_fft16k" in st.kernel_name
    out, at = [], 0
    for m in sizes:
        out.append(sdr_amd.fir(st, x[at:at + m]))
        at += m
    assert rel_err(np.concatenate(out), want) <= TOL
    auto = sdr_amd.FirState(taps, cplx)
    assert "fir_fft16k" in auto.kernel_name and "short calls" in auto.kernel_name
    assert rel_err(sdr_amd.fir(auto, x), want) <= TOL


@pytest.mark.parametrize("ntaps,cplx,n", [(131072, True, 300000), (131072, False, 40000), (65537, False, 600000)])
def test_16384_point_tiles_up_to_the_longest_filter(ntaps, cplx, n):
    """Sixteen partitions of 8192 taps (the library's limit of 131072 taps) and nine: streams shorter and longer than the filter,
    in three calls."""
    rng = np.random.default_rng(ntaps + n)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps)).astype(np.float32)
    x = rand(rng, n, cplx)
    st = sdr_amd.FirState(taps, cplx, path="fft16k")
    cut = n // 3
    y = np.concatenate([sdr_amd.fir(st, x[:cut]), sdr_amd.fir(st, x[cut:cut + 5]), sdr_amd.fir(st, x[cut + 5:])])
    assert rel_err(y, O.fir(taps, x)) <= TOL


@pytest.mark.parametrize("cplx", [True, False])
@pytest.mark.parametrize("ntaps", [2049, 4095, 8191])
def test_16384_point_tiles_call_sizes(ntaps, cplx):
    """fir_fft16k takes the whole call in one launch: calls of one sample, of less than a tile, of exactly one, two
    and three tiles' worth of outputs, and one sample either side of those (the last tile is an edge tile exactly when
    it is partial), as consecutive pieces of one stream."""
    rng = np.random.default_rng(ntaps + cplx)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    P = (16384 - (2048 if ntaps - 1 <= 2048 else 4096 if ntaps - 1 <= 4096 else 8192)) * (1 if cplx else 2)
    sizes = [1, 100, P - 1, P, P + 1, 2 * P, 2 * P - 1, 3 * P, 3 * P + 1, 7]
    x = rand(rng, sum(sizes), cplx)
    st = sdr_amd.FirState(taps, cplx, path="fft16k")
    out, at = [], 0
    for m in sizes:
        out.append(sdr_amd.fir(st, x[at:at + m]))
        at += m
    assert rel_err(np.concatenate(out), O.fir(taps, x)) <= TOL


@pytest.mark.parametrize("ntaps", [8191, 16383])
def test_16384_point_tiles_on_the_complex_pair_view(ntaps):
    """A long Hilbert design on real samples: calls of even length take the complex view (x[2m], x[2m+1]) filtered by the
    kept taps -- 4096 or 8192 of them: the 16384-point kernel with the kept taps' spectrum; odd calls in between take
    the real kernels; both leave the same history."""
    rng = np.random.default_rng(ntaps)
    taps = sdr_amd.hilbert(ntaps)
    st = sdr_amd.FirState(taps, False, path="fft16k")
    assert "fir_fft16k<REAL as complex pairs>" in st.kernel_name, st.kernel_name
    sizes = [200000, 3, 150001, 400000, 65536]
    x = rand(rng, sum(sizes), False)
    out, at = [], 0
    for m in sizes:
        out.append(sdr_amd.fir(st, x[at:at + m]))
        at += m
    assert rel_err(np.concatenate(out), O.fir(taps, x)) <= TOL
    auto = sdr_amd.FirState(taps, False)
    assert "fir_fft16k<REAL as complex pairs> (short calls" in auto.kernel_name
    assert rel_err(sdr_amd.fir(auto, x), O.fir(taps, x)) <= TOL


def test_kernel_family_chosen_by_measured_crossover():
    """sdr_fir_create picks the family by the measured three-way crossover
    (profiles/r02_direct_vs_fft_vs_mfma.txt): direct form up to 16 kept taps on complex samples, the matrix-core
    kernel for 17..44 kept taps (every-other-tap designs: ..49; profiles/r03_mfma_vs_fft.txt) and from the first tap
    on real samples (profiles/r03_direct_vs_mfma.txt), overlap-save beyond (and for odd-phase Hilbert designs)."""
    def name(kind, ntaps, cplx):
        return sdr_amd.FirState(taps_for(kind, ntaps), cplx).kernel_name
    assert "fir_direct" in name("lowpass", 15, True) and "fir_direct" in name("hilbert", 31, True)
    assert "fir_mfma" in name("lowpass", 15, False) and "fir_mfma" in name("hilbert", 31, False) and "fir_mfma" in name("random", 2, False)
    assert "fir_mfma" in name("lowpass", 31, True) and "fir_mfma" in name("lowpass", 44, False)
    assert "fir_fft4096" in name("lowpass", 45, True) and "fir_fft4096" in name("lowpass", 49, False)
    assert "fir_mfma" in name("hilbert", 63, False) and "fir_mfma" in name("hilbert", 95, True)
    assert "fir_fft4096" in name("lowpass", 65, True) and "fir_fft4096" in name("hilbert", 127, False)
    assert "fir_fft4096" in name("lowpass", 127, True) and "fir_fft16k" not in name("bandpass", 1025, True)
    assert "fir_fft16k" in name("bandpass", 1026, True) and "fir_fft16k" in name("bandpass", 2049, False)   # halo 2048: from 1026 taps on
    # 2050..8192 taps: 16384-point tiles when the call fills the chip with them (profiles/r03_long_taps_16k.txt),
    # the 4096-point forms for short calls; an every-other-tap design on real samples takes them on its complex-pair
    # view when it keeps 1026 taps or more
    assert "fir_fft16k" in name("bandpass", 4095, True) and "fir_fft16k" in name("bandpass", 8191, True)
    assert "fir_fft16k<REAL>" in name("bandpass", 4095, False) and "fir_fft16k" in name("bandpass", 8195, True)
    assert "fir_fft16k<REAL as complex pairs>" in name("hilbert", 4095, False)    # 2048 kept taps: halo 2048 on the pair view
    assert "fir_fft16k" not in name("hilbert", 2047, False)                       # 1024 kept taps: 4096-point tiles
    assert "fir_mfma" not in name("hilbert", 61, False)      # kept taps on the odd phase


def test_matrix_core_kernel_refuses_what_it_cannot_do():
    with pytest.raises(sdr_amd.SdrError):
        sdr_amd.FirState(taps_for("lowpass", 255), True, path="mfma")     # 255 dense taps
    with pytest.raises(sdr_amd.SdrError):
        sdr_amd.FirState(taps_for("hilbert", 125), False, path="mfma")    # non-zero taps on the odd phase


@pytest.mark.parametrize("ntaps", [2049, 2050, 4096, 9001])
def test_partitioned_long_filters(ntaps):
    """More than 2049 taps: uniformly partitioned overlap-save (hop 2048)."""
    rng = np.random.default_rng(ntaps)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps)).astype(np.float32)
    x = rand(rng, 40000, True)
    st = sdr_amd.FirState(taps, True)
    assert "fft" in st.kernel_name
    y = np.concatenate([sdr_amd.fir(st, x[:17000]), sdr_amd.fir(st, x[17000:17001]), sdr_amd.fir(st, x[17001:])])
    assert rel_err(y, O.fir(taps, x)) <= TOL


@pytest.mark.parametrize("ntaps,n", [(40001, 30000), (131072, 12000)])
def test_very_long_filters(ntaps, n):
    """Tens of partitions, up to the library's limit of 64 (131072 taps): the history is longer
    than the calls that feed it."""
    rng = np.random.default_rng(ntaps)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps)).astype(np.float32)
    x = rand(rng, n, False)
    st = sdr_amd.FirState(taps, False)
    assert "fft" in st.kernel_name
    cut = n // 3
    y = np.concatenate([sdr_amd.fir(st, x[:cut]), sdr_amd.fir(st, x[cut:cut + 5]), sdr_amd.fir(st, x[cut + 5:])])
    assert rel_err(y, O.fir(taps, x)) <= TOL
    with pytest.raises(sdr_amd.SdrError):
        sdr_amd.FirState(np.ones(131073 + 2048, np.float32), False)   # 65 partitions: refused, not mis-filtered


@pytest.mark.parametrize("grid", [3, 7, 0])
@pytest.mark.parametrize("ntaps,cplx", [(4095, True), (4095, False), (2050, False), (3001, True)])
def test_two_partition_spectrum_cache(ntaps, cplx, grid):
    """2050..4096 taps: every workgroup walks a contiguous run of tiles and reuses the
    previous tile's spectrum.  A grid of 3 or 7 workgroups makes the runs long and makes
    them cross channel boundaries (where the cached spectrum must be rebuilt)."""
    rng = np.random.default_rng(ntaps + grid)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps)).astype(np.float32)
    C, n = 5, 50000
    x = np.stack([rand(rng, n, cplx) for _ in range(C)])
    st = sdr_amd.FirState(taps, cplx, channels=C, grid=grid)
    y = np.concatenate([sdr_amd.fir(st, x[:, :30001]), sdr_amd.fir(st, x[:, 30001:])], axis=1)
    for c in range(C):
        assert rel_err(y[c], O.fir(taps, x[c])) <= TOL, c


@pytest.mark.parametrize("grid", [1, 2, 5, 0])
@pytest.mark.parametrize("ntaps,cplx", [(127, True), (127, False), (1000, True), (6000, False)])
def test_ordered_tile_tickets(ntaps, cplx, grid):
    """Single-spectrum and uncached partitioned overlap-save: a workgroup's tiles past its first two
    come from ticket counters that the call's last workgroup to finish leaves zeroed for the next
    launch.  Small grids make every workgroup draw many tickets across channel boundaries (grid 1:
    fewer workgroups than counters -- the launch must fall back to grid-strided tiles rather than
    leave the tiles of an unowned counter undone); the call sequence has a launch without interior
    tiles between two big ones and ends with a reset + replay (same counters, same result)."""
    rng = np.random.default_rng(ntaps + 7 * grid + cplx)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps)).astype(np.float32)
    C, n = 3, 140000
    x = np.stack([rand(rng, n, cplx) for _ in range(C)])
    st = sdr_amd.FirState(taps, cplx, channels=C, path="fft", grid=grid)
    cuts = [0, 60000, 60300, 60301, 125000, n]
    y = np.concatenate([sdr_amd.fir(st, x[:, cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)], axis=1)
    for c in range(C):
        assert rel_err(y[c], O.fir(taps, x[c])) <= TOL, c
    st.reset()
    y2 = sdr_amd.fir(st, x[:, :60000])
    assert np.array_equal(y2, y[:, :60000])


@pytest.mark.parametrize("kind,ntaps,cplx", CASES)
def test_single_call_vs_oracle(kind, ntaps, cplx):
    rng = np.random.default_rng(ntaps * 2 + cplx)
    taps = taps_for(kind, ntaps)
    n = 20011  # ragged: not a multiple of any tile size
    x = rand(rng, n, cplx)
    st = sdr_amd.FirState(taps, cplx)
    if kind == "hilbert" and 8 <= ntaps < 96:
        assert st.tap_stride == 2, "short Hilbert taps must select the zero-skipping direct kernel"
    y = sdr_amd.fir(st, x)
    ref = O.fir(taps, x)
    assert y.shape == ref.shape and y.dtype == ref.dtype
    assert rel_err(y, ref) <= TOL


@pytest.mark.parametrize("kind,ntaps,cplx", [("lowpass", 127, True), ("hilbert", 127, False),
                                             ("hilbert", 128, True), ("lowpass", 255, False),
                                             ("bandpass", 4095, False)])
def test_stream_cut_into_calls(kind, ntaps, cplx):
    """History carry (kfr::fir_state persists across packets, fir.hpp:65-67):
    results must not depend on how the stream is cut; cuts shorter than the
    history, of one sample and empty calls included."""
    rng = np.random.default_rng(7)
    taps = taps_for(kind, ntaps)
    cuts = [1, 0, 3, 50, 2044, 1, 126, 127, 128, 5000, 17, 8191, 2]
    x = rand(rng, sum(cuts), cplx)
    st = sdr_amd.FirState(taps, cplx)
    parts, p = [], 0
    for c in cuts:
        parts.append(sdr_amd.fir(st, x[p:p + c]))
        p += c
    y = np.concatenate(parts)
    ref = O.fir(taps, x)
    assert rel_err(y, ref) <= TOL
    # reset() zeroes the history (filter_fir::reset, fir.hpp:195-199)
    st.reset()
    y2 = sdr_amd.fir(st, x[:3000])
    assert rel_err(y2, ref[:3000]) <= TOL


def test_golden_kfr_outputs(golden_dir):
    """Against the reference's own kfr::fir outputs (tests/golden/fir.npz)."""
    g = np.load(os.path.join(golden_dir, "fir.npz"))
    for name, cplx in [("lowpass127_complex", True), ("lowpass127_real", False),
                       ("bandpass255_complex", True), ("bandpass4095_real", False)]:
        x, y, t = g[name + "_x"], g[name + "_y"], g[name + "_taps"]
        if cplx:
            x, y = x.view(np.complex64), y.view(np.complex64)
        st = sdr_amd.FirState(t, cplx)
        got = np.concatenate([sdr_amd.fir(st, x[:777]), sdr_amd.fir(st, x[777:])])
        assert rel_err(got, y) <= TOL, name


def test_golden_hilbert_block_samples(golden_dir):
    """Sample parity with the reference `hilbert` block's output stream: the
    filtered samples of the processed packets (delay skip applied on the host,
    blocks/hilbert.cpp:75-86)."""
    g = np.load(os.path.join(golden_dir, "hilbert.npz"))
    for name, ntaps, cplx in [("real_127", 127, False), ("complex_127", 127, True),
                              ("real_128", 128, False), ("gen_sine_127", 127, False)]:
        pk_in = O.parse_packets(bytes(g[name + "_in"]))
        pk_out = O.parse_packets(bytes(g[name + "_out"]))
        dt = np.complex64 if cplx else np.float32
        x = np.concatenate([np.frombuffer(p[4], dt) for p in pk_in])
        ref = np.concatenate([np.frombuffer(p[4], dt) for p in pk_out])
        st = sdr_amd.FirState(sdr_amd.hilbert(ntaps), cplx)
        y = sdr_amd.fir(st, x)[(ntaps - 1) // 2:]
        assert y.shape == ref.shape
        assert rel_err(y, ref) <= TOL, name


@pytest.mark.parametrize("cplx", [False, True])
def test_multichannel(cplx):
    rng = np.random.default_rng(11)
    taps = sdr_amd.hilbert(127)
    C, n = 5, 9001
    x = np.stack([rand(rng, n, cplx) for _ in range(C)])
    st = sdr_amd.FirState(taps, cplx, channels=C)
    y1 = sdr_amd.fir(st, x[:, :4000])
    y2 = sdr_amd.fir(st, x[:, 4000:])
    y = np.concatenate([y1, y2], axis=1)
    for c in range(C):
        assert rel_err(y[c], O.fir(taps, x[c])) <= TOL


@pytest.mark.parametrize("cplx", [False, True])
def test_device_buffers_unaligned_and_strided(cplx):
    """Device-resident entry point: pointers that are only sample-aligned (the
    element-wise load/store path) and channels with a stride larger than n."""
    import torch
    rng = np.random.default_rng(5)
    taps = sdr_amd.fir_lowpass(127, 0.1)
    C, n, stride = 3, 6001, 7000
    x = np.stack([rand(rng, n, cplx) for _ in range(C)])
    w = 2 if cplx else 1
    host = np.zeros((C, stride + 1, w), np.float32)
    host[:, 1:n + 1, :] = x.view(np.float32).reshape(C, n, w)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.full_like(d_in, 777.0)
    st = sdr_amd.FirState(taps, cplx, channels=C)
    off = w * 4  # skip one sample: 4- or 8-byte aligned only
    st.process_device(d_in.data_ptr() + off, n, d_out.data_ptr() + off,
                      in_stride=stride + 1, out_stride=stride + 1,
                      stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_out.cpu().numpy()
    # nothing outside [1, n] of each channel may be written
    assert np.all(got[:, 0] == 777.0) and np.all(got[:, n + 1:] == 777.0)
    y = got[:, 1:n + 1, :].reshape(C, n * w)
    for c in range(C):
        yc = y[c].view(np.complex64) if cplx else y[c]
        assert rel_err(yc, O.fir(taps, x[c])) <= TOL


@pytest.mark.parametrize("ntaps", [127, 128, 255, 1003, 4095])
def test_real_stream_as_complex_pairs(ntaps):
    """Real samples through a filter whose odd-index taps are zero (the Hilbert design) on the overlap-save
    path: calls with an even length, even strides and 8-byte aligned buffers run the COMPLEX kernel on the
    (x[2m], x[2m+1]) view with the kept taps; every other call runs the real kernel.  Both leave the same
    history, so a stream cut into calls of both kinds -- even, odd, one sample, odd-aligned device pointers,
    odd channel strides -- must come out as one filter."""
    import torch
    rng = np.random.default_rng(ntaps)
    taps = taps_for("hilbert", ntaps)
    st = sdr_amd.FirState(taps, False, path="fft")
    assert "pairs" in st.kernel_name and st.tap_stride == 2, st.kernel_name
    # (a design whose kept taps sit on the ODD indices -- hilbert(1001): centre tap 500 -- keeps the real kernel)
    assert "pairs" not in sdr_amd.FirState(taps_for("hilbert", 1001), False, path="fft").kernel_name
    n = 90001
    x = rand(rng, n, False)
    want = O.fir(taps, x)
    cuts = [0, 30000, 30001, 50002, 50003, 50004, 70004, n]   # even, 1, odd (20001), 1, 1, even, odd
    y = np.concatenate([sdr_amd.fir(st, x[cuts[i]:cuts[i + 1]]) for i in range(len(cuts) - 1)])
    assert rel_err(y, want) <= TOL
    # device pointers: 8-byte aligned (pair view) and only 4-byte aligned (real view), same stream
    st.reset()
    xt = torch.from_numpy(x).cuda()
    outs = [sdr_amd.fir(st, xt[:40000]), sdr_amd.fir(st, xt[40000:40001]), sdr_amd.fir(st, xt[40001:60001]),
            sdr_amd.fir(st, xt[60001:])]
    y2 = torch.cat(outs).cpu().numpy()
    assert rel_err(y2, want) <= TOL
    # channels: even stride (pair view) and odd stride (real view)
    C = 3
    xc = np.stack([rand(rng, 20001, False) for _ in range(C)])
    stc = sdr_amd.FirState(taps, False, channels=C, path="fft")
    yc = np.concatenate([sdr_amd.fir(stc, xc[:, :12000]), sdr_amd.fir(stc, xc[:, 12000:])], axis=1)
    for c in range(C):
        assert rel_err(yc[c], O.fir(taps, xc[c])) <= TOL, c


def test_linearity_and_impulse_full_size():
    """Config C2 size (2^28 complex samples, 127-tap lowpass) through
    size-independent properties: impulse response = taps at the impulse
    positions, windows against the oracle, and no sample dropped or reordered
    across tile boundaries (ramp of impulses)."""
    import torch
    n = 1 << 28
    taps = sdr_amd.fir_lowpass(127, 0.1)
    st = sdr_amd.FirState(taps, True)
    g = torch.Generator(device="cuda").manual_seed(0x5D12)
    x = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    x.uniform_(-1, 1, generator=g)
    y = st.process_tensor(x)
    torch.cuda.synchronize()
    rng = np.random.default_rng(3)
    starts = [0, n - 65536] + [int(s) for s in rng.integers(0, n - 70000, 8)] + [2048 * 777 - 100, (1 << 27) - 300]
    for s in starts:
        lo = max(0, s - 126)
        xin = x[lo:s + 4096].cpu().numpy().reshape(-1).view(np.complex64)
        ref = O.fir(taps, xin)[s - lo:]
        got = y[s:s + 4096].cpu().numpy().reshape(-1).view(np.complex64)
        assert rel_err(got, ref) <= TOL, s
    # impulses spaced 1000 apart: output is the tap vector at each, exactly
    del y
    x.zero_()
    pos = torch.arange(0, n - 200, 1000003, device="cuda")
    x[pos, 0] = 1.0
    x[pos, 1] = -2.0
    st.reset()
    y = st.process_tensor(x)
    torch.cuda.synchronize()
    t = torch.from_numpy(taps).cuda()
    # exact for the direct form; the overlap-save FFT path reproduces the taps to
    # fp32 FFT rounding (a few 1e-7 of the largest tap)
    atol = 2e-6 * float(t.abs().max())
    for p in pos[:: max(1, len(pos) // 16)].tolist() + [int(pos[-1])]:
        seg = y[p:p + 127]
        assert float((seg[:, 0] - t).abs().max()) <= atol, p
        assert float((seg[:, 1] + 2.0 * t).abs().max()) <= 2 * atol, p
    # nothing leaks between the impulses (no sample dropped, duplicated or moved)
    assert float(y.abs().sum()) == pytest.approx(float(len(pos) * 3.0 * t.abs().sum()), rel=2e-3)


@pytest.mark.parametrize("path,ntaps,reach", [("direct", 15, 16), ("mfma", 31, 144), ("fft", 127, 8192)])
@pytest.mark.parametrize("bad", [np.nan, np.inf])
def test_non_finite_sample_stays_local(path, ntaps, reach, bad):
    """kfr::fir is a direct form: a NaN / Inf input sample reaches the next `ntaps` outputs and no
    other.  The direct-form kernel pads its taps with zeros to a whole number of chunks (NaN * 0: up
    to 15 more outputs); the matrix-core kernel multiplies the explicit zeros of its Toeplitz operand,
    so up to 144 neighbouring outputs on either side turn NaN too; the overlap-save kernel loses the tile the sample sits in -- for real signals the PAIR of tiles
    that share one complex transform, up to 8192 samples -- (documented in
    INTEGRATION.md).  Everything outside that reach matches the oracle."""
    rng = np.random.default_rng(ntaps)
    taps = taps_for("lowpass", ntaps)
    n, p = 30000, 13000
    x = rand(rng, n, False)
    x[p] = bad
    st = sdr_amd.FirState(taps, False, path=path)
    y = sdr_amd.fir(st, x)
    want = O.fir(taps, x)
    assert not np.isfinite(want[p:p + ntaps]).any() and np.isfinite(want[:p]).all() and np.isfinite(want[p + ntaps:]).all()
    assert not np.isfinite(y[p:p + ntaps]).any()
    lo, hi = p - reach, p + ntaps + reach
    assert np.isfinite(y[:lo]).all() and np.isfinite(y[hi:]).all()
    keep = np.r_[0:lo, hi:n]
    assert rel_err(y[keep], want[keep]) <= TOL


@pytest.mark.parametrize("cplx,reach", [(True, 16384), (False, 32768)])
def test_non_finite_sample_in_a_16384_point_tile(cplx, reach):
    """The 16384-point kernel loses the tile a NaN sits in (real samples: the two stretches that share the transform),
    nothing else."""
    rng = np.random.default_rng(5)
    ntaps = 4095
    taps = taps_for("bandpass", ntaps)
    n, p = (200000, 100000) if cplx else (400000, 200000)
    x = rand(rng, n, cplx)
    x[p] = np.nan
    st = sdr_amd.FirState(taps, cplx, path="fft16k")
    y = sdr_amd.fir(st, x)
    want = O.fir(taps, x)
    assert not np.isfinite(y[p:p + ntaps]).any()
    lo, hi = p - reach, p + ntaps + reach
    assert np.isfinite(y[:lo]).all() and np.isfinite(y[hi:]).all()
    keep = np.r_[0:lo, hi:n]
    assert rel_err(y[keep], want[keep]) <= TOL


@pytest.mark.parametrize("kind,ntaps,cplx,family", [("hilbert", 127, False, "fir_direct"), ("lowpass", 127, True, "fir_direct"),
                                                    ("lowpass", 31, True, "fir_mfma"), ("bandpass", 4095, True, "fir_direct")])
def test_per_output_numerics(kind, ntaps, cplx, family):
    """sdr_fir_create_numerics(SDR_NUMERICS_PER_OUTPUT): the public switch that keeps a handle on the direct-form
    kernels whatever the tap count, with kfr::fir's own character (libs/kfr/dsp/fir.hpp:129-146): parity as ever, a
    NaN confined to the outputs it reaches, and -- what the overlap-save default cannot give -- a weak stretch of
    the stream right behind a strong one held to the bar relative to ITS OWN level."""
    rng = np.random.default_rng(ntaps)
    taps = taps_for(kind, ntaps)
    st = sdr_amd.FirState(taps, cplx, numerics="per-output")
    assert family in st.kernel_name and "fft" not in st.kernel_name
    fast = sdr_amd.FirState(taps, cplx, numerics="fast")
    assert fast.kernel_name == sdr_amd.FirState(taps, cplx).kernel_name
    n, cut = 60000, 30000
    x = rand(rng, n, cplx)
    x[:cut] *= 1e3
    x[cut:] *= 1e-3
    y = np.concatenate([sdr_amd.fir(st, x[:12345]), sdr_amd.fir(st, x[12345:])])
    want = O.fir(taps, x)
    assert rel_err(y, want) <= TOL
    weak = slice(cut + ntaps, n)                       # past the filter's memory of the strong stretch
    assert rel_err(y[weak], want[weak]) <= TOL         # relative to the weak stretch's own RMS (1e-6 of the strong one's)
    if "fft" in fast.kernel_name:                      # the default path cannot: its error is relative to the tile
        yf = sdr_amd.fir(fast, x)
        assert rel_err(yf, want) <= TOL and rel_err(yf[weak], want[weak]) > TOL
    st.reset()
    x2 = rand(rng, n, cplx)
    x2[cut] = np.nan
    y2 = sdr_amd.fir(st, x2)
    # (the every-other-tap kernel never multiplies the Hilbert design's zero taps: kfr::fir's NaN * 0 stays finite here)
    assert not np.isfinite(y2[cut + np.flatnonzero(taps)]).any()
    assert np.isfinite(y2[:cut - 144]).all() and np.isfinite(y2[cut + ntaps + 144:]).all()


def test_per_output_numerics_arguments():
    taps = taps_for("lowpass", 127)
    with pytest.raises(sdr_amd.SdrError, match="numerics must be"):
        sdr_amd.FirState(taps, True, numerics=7)
    with pytest.raises(sdr_amd.SdrError, match="too large for the direct-form"):
        sdr_amd.FirState(taps_for("lowpass", 20001), True, numerics="per-output")


@pytest.mark.parametrize("kind,ntaps,cplx", [("lowpass", 127, True), ("hilbert", 127, False), ("lowpass", 31, True),
                                             ("lowpass", 15, False), ("bandpass", 4095, True)])
def test_hip_graph_of_two_calls_replays_as_a_stream(kind, ntaps, cplx):
    """Two device calls captured once and replayed three times filter six consecutive chunks as one stream;
    nothing in the device entry point allocates or synchronises, so the capture itself must succeed.  (Any
    number of calls per graph works since round 3 -- tests/test_graph_capture_gpu.py captures ONE.)"""
    import torch
    rng = np.random.default_rng(ntaps + cplx)
    taps = taps_for(kind, ntaps)
    n, reps = 30000, 3
    x = rand(rng, 2 * n * reps, cplx)
    st = sdr_amd.FirState(taps, cplx)
    shape = (2 * n, 2) if cplx else (2 * n,)
    d_in = torch.zeros(shape, dtype=torch.float32, device="cuda")
    d_out = torch.zeros(shape, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        st.process_tensor(d_in[:n], d_out[:n])
        st.process_tensor(d_in[n:], d_out[n:])
    got = []
    for r in range(reps):
        chunk = x[2 * n * r:2 * n * (r + 1)]
        d_in.copy_(torch.from_numpy(chunk.view(np.float32).reshape(shape)))
        graph.replay()
        torch.cuda.synchronize()
        o = d_out.cpu().numpy()
        got.append(o.reshape(-1).view(np.complex64) if cplx else o.reshape(-1))
    assert rel_err(np.concatenate(got), O.fir(taps, x)) <= TOL


@pytest.mark.parametrize("ntaps,cplx,cuts", [(4604, False, [396]), (4604, True, [100, 200, 396, 4000, 4604, 4700, 30000]),
                                             (2500, False, [1, 2499, 2500, 2501, 9000]), (1000, True, [999, 5000]),
                                             (127, False, [50, 126, 127, 4000])])
def test_onset_of_a_stream_shorter_than_its_filter(ntaps, cplx, cuts):
    """kfr::fir rounds every output relative to itself (libs/kfr/dsp/fir.hpp:137-143).  A stream still shorter than
    its filter is all onset: with a windowed design the first taps are tiny and the output lies orders of magnitude
    below the settled level -- below the rounding of a fast convolution, which works at the level of the whole
    spectrum.  Calls that end inside the first ntaps samples run the direct form; every PIECE of the stream is held
    to 1e-5 of ITS OWN output's RMS here (the first case is seed 307 of the round-2 fuzz soak: 4604 taps, 396
    samples), and a reset starts the onset again."""
    rng = np.random.default_rng(ntaps + cplx)
    taps = (rng.uniform(-1, 1, ntaps) * np.hanning(ntaps + 2)[1:-1]).astype(np.float32)
    n = cuts[-1]
    x = rand(rng, n, cplx)
    st = sdr_amd.FirState(taps, cplx)
    assert "fft" in st.kernel_name
    want = O.fir(taps, x)
    for rep in range(2):
        lo = 0
        for hi in cuts:
            y = sdr_amd.fir(st, x[lo:hi])
            assert rel_err(y, want[lo:hi]) <= TOL, (rep, lo, hi)
            lo = hi
        st.reset()


def test_in_place_is_refused():
    """Tiles re-read their neighbours' inputs: the ABI says in and out must not overlap, and
    the library checks rather than filter its own outputs."""
    import torch
    st = sdr_amd.FirState(sdr_amd.fir_lowpass(127, 0.1), False)
    x = torch.zeros(100000, dtype=torch.float32, device="cuda")
    with pytest.raises(sdr_amd.SdrError, match="overlap"):
        st.process_device(x.data_ptr(), 50000, x.data_ptr() + 4 * 1000)
    st.process_device(x.data_ptr(), 50000, x.data_ptr() + 4 * 50000)   # adjacent is fine
    torch.cuda.synchronize()
