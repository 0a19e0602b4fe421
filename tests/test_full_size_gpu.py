"""BASELINE configs 3, 4 and 5 at their FULL sizes (config 2's is in test_fir_gpu.py):
the oracle cannot filter 2^26..2^30 samples in test time, so parity is checked on windows --
the first and last outputs, random positions, positions straddling tile / partition / channel
boundaries -- each against the oracle run on the window's own inputs (plus the samples its
filters reach back to), and through size-independent properties (output counts, nothing
written past the end)."""
import numpy as np
import pytest

import oracle_lib as O
import sdr_amd

pytestmark = pytest.mark.gpu
TOL = 1e-5
STAGES = lambda: [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.1), 0)]  # noqa: E731


def max_err(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b).astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    return float(np.max(np.abs(a - b)))


def signal_rms(t):
    """RMS of the output signal in steady state (the tolerance is stated against the signal's
    RMS: a window inside a long filter's warm-up is far below it)."""
    import torch
    mid = t.reshape(-1)[t.numel() // 4: t.numel() // 4 + (1 << 22)].to(torch.float64)
    scale = 2.0 if t.dim() == 2 and t.shape[-1] == 2 else 1.0   # (n, 2) = complex samples
    return float(mid.pow(2).mean().sqrt()) * np.sqrt(scale)


def staged_window(stages, x_window):
    """Staged oracle on a window cut out of the middle of a stream.  Returns (outputs, first):
    outputs[first:] equal the stream's outputs at the same positions -- before `first` the
    window's zero history (and the skip bookkeeping) still shows."""
    y = x_window
    first = 0
    for taps, skip in stages:
        y = O.fir(taps, y)[skip:]
        first += len(taps) - 1 - skip
        first = max(first, 0)
    return y, first + sum(s for _, s in stages)


def test_config3_full_size_windows():
    """C3: 4095-tap bandpass on 2^26 complex samples (two partitions, cached-spectrum kernel)."""
    import torch
    n, T = 1 << 26, 4095
    taps = sdr_amd.fir_bandpass(T, 0.1, 0.2)
    st = sdr_amd.FirState(taps, True)
    g = torch.Generator(device="cuda").manual_seed(0x5D12)
    x = torch.empty((n, 2), dtype=torch.float32, device="cuda")
    x.uniform_(-1, 1, generator=g)
    y = st.process_tensor(x)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (n, 2)
    rms = signal_rms(y)
    rng = np.random.default_rng(5)
    W = 1024
    starts = [0, n - W] + [int(s) for s in rng.integers(0, n - W, 5)] + [2048 * 9001 - 500, (1 << 25) - 300]
    for s in starts:
        lo = max(0, s - (T - 1))
        xin = x[lo:s + W].cpu().numpy().reshape(-1).view(np.complex64)
        ref = O.fir(taps, xin)[s - lo:]
        got = y[s:s + W].cpu().numpy().reshape(-1).view(np.complex64)
        assert max_err(got, ref) <= TOL * rms, s


def test_config4_full_size_64_channels():
    """C4: 64 channels x 2^24 real samples, channel c a sine at (c+1)/160 cycles/sample,
    hilbert(127, delay skip) then lowpass(127), layout [channel][sample]."""
    import torch
    C, n = 64, 1 << 24
    stages = STAGES()
    chain = sdr_amd.Chain(stages, False, channels=C)
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    x = torch.empty((C, n), dtype=torch.float32, device="cuda")
    for c in range(C):
        x[c] = torch.sin(t * (2 * np.pi * (c + 1) / 160.0)).to(torch.float32)
    del t
    out, n_out = chain.process_tensor(x)
    torch.cuda.synchronize()
    assert n_out == n - 63
    rms = 1.0 / np.sqrt(2.0)                  # the signal: unit sines (stop-band channels come out near zero)
    rng = np.random.default_rng(11)
    W = 4096
    for c in (0, 1, 17, 40, 63):
        # the stream's first outputs (warm-up of both stages) ...
        ref = O.fir(stages[1][0], O.fir(stages[0][0], x[c, :W + 63].cpu().numpy())[63:])
        assert max_err(out[c, :W].cpu().numpy(), ref) <= TOL * rms, c
        # ... its last ones, and windows inside
        for s in [n_out - W] + [int(v) for v in rng.integers(300, n_out - W, 3)]:
            lo = s - 252
            ref, first = staged_window(stages, x[c, lo:s + W + 63].cpu().numpy())
            got = out[c, lo:lo + len(ref)].cpu().numpy()
            assert max_err(got[first:], ref[first:]) <= TOL * rms, (c, s)
    # a sine in the lowpass's passband comes out as its quadrature at full amplitude; far
    # outside it is gone: every channel at once, no channel mixed up with its neighbour
    # (|H| of the chain at f_c, from the taps)
    fused = np.convolve(stages[0][0].astype(np.float64), stages[1][0].astype(np.float64))
    k = np.arange(len(fused))
    mid = slice(1 << 20, (1 << 20) + (1 << 16))
    out_rms = out[:, mid].to(torch.float64).pow(2).mean(dim=1).sqrt().cpu().numpy()
    for c in range(C):
        gain = abs(np.sum(fused * np.exp(-2j * np.pi * (c + 1) / 160.0 * k)))
        assert out_rms[c] == pytest.approx(gain / np.sqrt(2.0), abs=2e-4), c


def test_config5_full_size_chunk():
    """C5's share of one GPU: 2^18 generator packets (2044 samples each, 5.4e8 samples) through
    gen -> hilbert(127, skip) -> lowpass(127) with the samples synthesised inside the FIR
    kernel, in one call; windows against the oracle's generator + staged filters."""
    import torch
    B = 2044
    blocks = 1 << 18
    cps = float(np.float32(100e3) / np.float32(1e6))
    stages = STAGES()
    gen = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
    chain = sdr_amd.Chain(stages, False)
    out = torch.full((blocks * B,), 777.0, dtype=torch.float32, device="cuda")
    done = chain.generate_device(gen, blocks, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert done == blocks * B - 63
    assert bool((out[done:] == 777.0).all())
    rms = signal_rms(out[:done])

    # the oracle's generator runs the whole stream (its phase recurrence is serial), in
    # pieces; only the windows are kept
    rng = np.random.default_rng(23)
    W = 2 * B
    want = sorted([0, blocks - 4] + [int(v) for v in rng.integers(1, blocks - 4, 6)] + [(1 << 17) - 1])
    piece, phi, windows = 4096, 0.0, {}
    tail = np.empty(0, np.float32)            # the previous piece's last 4 packets
    for b0 in range(0, blocks, piece):
        x, phi = O.gen_real(piece, B, 1.0, cps, phi, sdr_amd.WAVEFORMS["sine"])
        both, base = np.concatenate([tail, x]), b0 - len(tail) // B
        for wb in want:
            if wb not in windows and base <= wb and wb + 4 <= b0 + piece:
                windows[wb] = both[(wb - base) * B:(wb - base + 4) * B].copy()
        tail = x[-4 * B:]
    assert sorted(windows) == sorted(set(want))
    for wb, xw in windows.items():
        s = wb * B
        if wb == 0:
            ref = O.fir(stages[1][0], O.fir(stages[0][0], xw)[63:])
            got = out[:len(ref)].cpu().numpy()
            assert max_err(got, ref) <= TOL * rms
            continue
        ref, first = staged_window(stages, xw)
        got = out[s:s + len(ref)].cpu().numpy()
        assert max_err(got[first:first + W], ref[first:first + W]) <= TOL * rms, wb


@pytest.mark.parametrize("ntaps,cplx,log2n", [(127, False, 32), (15, False, 32), (127, True, 31), (4095, False, 32)])
def test_calls_beyond_2_to_the_32_samples(ntaps, cplx, log2n):
    """One call on more than 2^32 real (2^31 complex) samples -- the whole of BASELINE config 5's
    buffer on one GPU: sample indices are 64-bit end to end.  Windows around 2^31, 2^32 and the
    end of the buffer against the oracle; nothing written past the end."""
    import torch
    n = (1 << log2n) + 12345
    w = 2 if cplx else 1
    taps = sdr_amd.fir_lowpass(ntaps, 0.1) if ntaps < 1000 else sdr_amd.fir_bandpass(ntaps, 0.1, 0.2)
    st = sdr_amd.FirState(taps, cplx)
    x = torch.empty(n * w, dtype=torch.float32, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(ntaps)
    x.uniform_(-1, 1, generator=g)
    y = torch.full((n * w + 64,), 777.0, dtype=torch.float32, device="cuda")
    st.process_device(x.data_ptr(), n, y.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((y[n * w:] == 777.0).all())
    rms = float(y[w * (1 << 20): w * ((1 << 20) + (1 << 22))].to(torch.float64).pow(2).mean().sqrt()) * np.sqrt(w)
    W = 2048
    for s in [0, (1 << 31) // w - 1000, (1 << 31) - 1000, (1 << 32) // w - 1000, n - W]:
        if s + W > n:
            continue
        lo = max(0, s - (ntaps - 1))
        xin = x[lo * w:(s + W) * w].cpu().numpy()
        xin = xin.view(np.complex64) if cplx else xin
        ref = O.fir(taps, xin)[s - lo:]
        got = y[s * w:(s + W) * w].cpu().numpy()
        got = got.view(np.complex64) if cplx else got
        assert max_err(got, ref) <= TOL * rms, (ntaps, cplx, s)


def test_config5_whole_buffer_in_one_call():
    """BASELINE config 5's whole 2^32-sample buffer generated and filtered by ONE call on one GPU
    (2 101 300 packets of 2044 samples, 17 GB of outputs): the tail of the buffer, past 2^32,
    equals what a second generator + chain positioned there (Gen.advance, Chain.seek, primed with
    the halo packets) produce -- packet and sample indices are 64-bit inside the fused kernel."""
    import torch
    from sdr_amd import sharding
    B = 2044
    blocks = 2101300                          # x 2044 = 4 295 057 200 > 2^32
    cps = float(np.float32(100e3) / np.float32(1e6))
    stages = STAGES()
    stream = torch.cuda.current_stream().cuda_stream
    gen = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
    chain = sdr_amd.Chain(stages, False)
    out = torch.empty(blocks * B, dtype=torch.float32, device="cuda")
    done = chain.generate_device(gen, blocks, out.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert done == blocks * B - 63 and done > (1 << 32)
    # the start of the stream against the oracle
    x, _ = O.gen_real(8, B, 1.0, cps, 0.0, sdr_amd.WAVEFORMS["sine"])
    ref = O.fir(stages[1][0], O.fir(stages[0][0], x)[63:])
    assert max_err(out[:len(ref)].cpu().numpy(), ref) <= TOL * np.sqrt(np.mean(ref[1000:].astype(np.float64) ** 2))
    # the last 900 packets, produced independently from their own position
    tail_blocks = 900
    first = blocks - tail_blocks
    halo = sharding.plan_generated_chain(blocks, 2, B, 253)[1][2]
    gen2 = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
    chain2 = sdr_amd.Chain(stages, False)
    gen2.advance(first - halo)
    chain2.seek((first - halo) * B)
    tmp = torch.empty((tail_blocks + halo) * B, dtype=torch.float32, device="cuda")
    assert chain2.generate_device(gen2, halo, tmp.data_ptr(), stream=stream) == halo * B      # priming, discarded
    n2 = chain2.generate_device(gen2, tail_blocks, tmp.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    assert n2 == tail_blocks * B
    a = out[done - n2:done]
    b = tmp[:n2]
    rms = float(a.to(torch.float64).pow(2).mean().sqrt())
    assert float((a - b).abs().max()) <= 3e-6 * rms      # two fp32 results with different tile offsets
