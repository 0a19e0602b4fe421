"""HIP graph capture of the device entry points (through the C ABI).

A handle's stream position -- which of its two history buffers is current, the overlap-save
kernels' tile tickets -- is DEVICE state, read and moved on by the kernels themselves
(sdr_amd/csrc/fir_kernels.h: FirDevState).  So ONE captured call replays as the stream's next
call, and replays mix freely with ordinary calls: the contract of kfr::fir_state (state lives
across calls, reference libs/kfr/dsp/fir.hpp:57-68) holds whoever launches the kernels.
Calls whose work depends on host-side position refuse to be captured instead of recording
something that would replay wrongly.  Tolerance as everywhere: 1e-5 of the output's RMS.
"""
import numpy as np
import pytest

import oracle_lib as O
import sdr_amd

pytestmark = pytest.mark.gpu
TOL = 1e-5


def rel_err(a, b):
    a = np.asarray(a).astype(np.complex128 if np.iscomplexobj(a) else np.float64)
    b = np.asarray(b).astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    return float(np.max(np.abs(a - b)) / np.sqrt(np.mean(np.abs(b) ** 2)))


def rand(rng, n, cplx):
    if cplx:
        return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    return rng.uniform(-1, 1, n).astype(np.float32)


def to_dev(torch, x, cplx):
    a = np.ascontiguousarray(x)
    return torch.from_numpy(a.view(np.float32).reshape(-1, 2) if cplx else a).cuda()


def to_host(t, cplx):
    o = t.cpu().numpy()
    return o.reshape(-1).view(np.complex64) if cplx else o.reshape(-1)


# (kernel family pinned through the internal hook, taps, complex?, samples per captured call)
FAMILIES = [
    ("direct", sdr_amd.fir_lowpass, 15, False, 30000),
    ("direct", sdr_amd.fir_lowpass, 127, True, 9001),
    ("mfma", sdr_amd.fir_lowpass, 31, True, 30000),
    ("fft", sdr_amd.fir_lowpass, 127, True, 300000),     # ordered tile tickets in play (many interior tiles)
    ("fft", sdr_amd.hilbert, 127, False, 300000),        # real samples as complex pairs
    ("fft", sdr_amd.fir_lowpass, 127, False, 30001),     # odd length: the real kernel
    ("fft", None, 4095, True, 200000),                   # two partitions: cached spectrum, one launch
    ("fft", None, 9001, True, 60000),                    # five partitions: main launch + carry / edge launch
    ("fft16k", None, 4095, True, 200000),                # 16384-point tiles: interior, edge and carry workgroups in one launch
    ("fft16k", None, 6143, False, 150001),               # the same on real samples (two stretches per transform), halo 8192
]


def make_taps(design, ntaps):
    if design is sdr_amd.hilbert:
        return sdr_amd.hilbert(ntaps)
    if design is None:
        return sdr_amd.fir_bandpass(ntaps, 0.1, 0.2)
    return design(ntaps, 0.1)


@pytest.mark.parametrize("path,design,ntaps,cplx,n", FAMILIES)
def test_one_captured_call_replays_as_the_stream(path, design, ntaps, cplx, n):
    """ONE sdr_fir_process_device captured into a graph, replayed five times with ordinary calls of other
    lengths in between: the eight pieces are one stream.  (Before round 3 the history buffer in use and
    the ticket set were toggled on the host at call time: a graph of an odd number of calls re-read stale
    history and drew tickets from used counters.)"""
    import torch
    rng = np.random.default_rng(ntaps + 7 * cplx + n)
    taps = make_taps(design, ntaps)
    st = sdr_amd.FirState(taps, cplx, path=path)
    shape = (n, 2) if cplx else (n,)
    d_in = torch.zeros(shape, dtype=torch.float32, device="cuda")
    d_out = torch.zeros(shape, dtype=torch.float32, device="cuda")
    # an ordinary call first: the stream is under way when the capture is taken
    first = rand(rng, 5000, cplx)
    pieces_in, pieces_out = [first], [to_host(st.process_tensor(to_dev(torch, first, cplx)), cplx)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        st.process_tensor(d_in, d_out)
    # capturing runs nothing: the stream has not moved
    between = [None, 777, None, None, 12345, None, 4096]   # None = a replay of the captured call
    for b in between:
        if b is None:
            chunk = rand(rng, n, cplx)
            d_in.copy_(to_dev(torch, chunk, cplx))
            graph.replay()
            torch.cuda.synchronize()
            pieces_out.append(to_host(d_out, cplx).copy())
        else:
            chunk = rand(rng, b, cplx)
            y = st.process_tensor(to_dev(torch, chunk, cplx))
            torch.cuda.synchronize()
            pieces_out.append(to_host(y, cplx))
        pieces_in.append(chunk)
    x = np.concatenate(pieces_in)
    assert rel_err(np.concatenate(pieces_out), O.fir(taps, x)) <= TOL


def test_captured_call_with_channels_and_reset():
    """Four channels with a padded stride through one captured call; sdr_fir_reset between two runs of
    replays starts the stream again (history, position and tickets cleared on the device)."""
    import torch
    rng = np.random.default_rng(5)
    taps = sdr_amd.fir_lowpass(127, 0.1)
    ch, n, stride = 4, 70000, 70016
    st = sdr_amd.FirState(taps, False, channels=ch)
    d_in = torch.zeros((ch, stride), dtype=torch.float32, device="cuda")
    d_out = torch.zeros((ch, stride), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        st.process_device(d_in.data_ptr(), n, d_out.data_ptr(), in_stride=stride, out_stride=stride,
                          stream=torch.cuda.current_stream().cuda_stream)
    for run in range(2):
        x = rng.uniform(-1, 1, (ch, 3 * n)).astype(np.float32)
        got = []
        for r in range(3):
            d_in[:, :n].copy_(torch.from_numpy(x[:, r * n:(r + 1) * n]))
            graph.replay()
            torch.cuda.synchronize()
            got.append(d_out[:, :n].cpu().numpy())
        y = np.concatenate(got, axis=1)
        for c in range(ch):
            assert rel_err(y[c], O.fir(taps, x[c])) <= TOL, (run, c)
        st.reset()


def test_chain_call_past_the_start_of_the_stream_is_capturable_and_the_start_is_refused():
    """sdr_chain_process_device: inside the start-of-stream region (group-delay skip, staged warm-up) the
    work depends on the host-side position and a capture is refused with a message; a later call is one
    call of the fused filter and replays as the stream."""
    import torch
    rng = np.random.default_rng(11)
    h, lp = sdr_amd.hilbert(127), sdr_amd.fir_lowpass(127, 0.1)
    chain = sdr_amd.Chain([(h, 63), (lp, 0)], False)
    n, reps = 100000, 5
    x = rng.uniform(-1, 1, n * (reps + 1)).astype(np.float32)
    d_in = torch.zeros(n, dtype=torch.float32, device="cuda")
    d_out = torch.zeros(n, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    g0 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g0, stream=side):
        d_out.zero_()
        with pytest.raises(sdr_amd.SdrError, match="start-of-stream"):
            chain.process_tensor(d_in, d_out)
    # first piece by an ordinary call (swallows the 63 skipped outputs), then one captured call replayed
    d_in.copy_(torch.from_numpy(x[:n]))
    _, n0 = chain.process_tensor(d_in, d_out)
    torch.cuda.synchronize()
    got = [d_out[:n0].cpu().numpy()]
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        chain.process_tensor(d_in, d_out)
    for r in range(reps):
        d_in.copy_(torch.from_numpy(x[(r + 1) * n:(r + 2) * n]))
        graph.replay()
        torch.cuda.synchronize()
        got.append(d_out.cpu().numpy().copy())
    want = O.fir(lp, O.fir(h, x)[63:])
    assert rel_err(np.concatenate(got), want) <= TOL


def staged(stages, x):
    y = x
    for taps, skip in stages:
        y = O.fir(taps, y)[skip:]
    return y


@pytest.mark.parametrize("wf,phase", [("sine", 0.0), ("cosine", 90.0), ("triangle", 45.0)])
def test_one_captured_fused_generator_call_replays_as_the_stream(wf, phase):
    """ONE sdr_chain_generate_device (gen -> hilbert -> lowpass, samples synthesised inside the FIR kernel)
    captured and replayed five times, an ordinary call in between: eight calls, one stream.  The generator's
    position is device state when its phase recurrence has a closed form (phase and per-block increment whole
    multiples of 2^-23: the usual case, here 0.1 cycles per sample in 2044-sample packets): the kernels compute
    every block's start phase from it and the call's last workgroup moves it on."""
    import torch
    B = sdr_amd.optimal_block_size(4, 1000000)
    cps = float(np.float32(100e3) / np.float32(1e6))
    stages = [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.12), 0)]
    gen = sdr_amd.Gen(wf, False, 0.8, cps, phase, B)
    chain = sdr_amd.Chain(stages, False)
    nb = 40
    out = torch.zeros(nb * B, dtype=torch.float32, device="cuda")
    got = []
    n0 = chain.generate_device(gen, 3, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)  # start of the stream
    torch.cuda.synchronize()
    got.append(out[:n0].cpu().numpy())
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        assert chain.generate_device(gen, nb, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream) == nb * B
    blocks = 3
    for step in (None, None, 7, None, None, None):
        if step is None:
            graph.replay()
            torch.cuda.synchronize()
            got.append(out.cpu().numpy().copy())
            blocks += nb
        else:
            n1 = chain.generate_device(gen, step, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got.append(out[:n1].cpu().numpy())
            blocks += step
    x, _ = O.gen_real(blocks, B, 0.8, cps, float(np.float32(phase) / np.float32(360.0)), sdr_amd.WAVEFORMS[wf])
    want = staged(stages, x)
    y = np.concatenate(got)
    assert y.shape == want.shape and rel_err(y, want) <= TOL


def test_captured_generator_kernel_and_what_cannot_be_captured():
    """sdr_gen_generate_device with a closed-form phase chain: one captured call replays as consecutive packets,
    bit for bit the oracle's recipe.  A generator whose phase recurrence rounds (here 1e-5 cycles per sample: the
    per-block increment is no multiple of 2^-23) needs its per-call phase table from the host: a capture is
    refused with a message, and so is one of the quadrature waveforms."""
    import torch
    B = 2044
    cps = float(np.float32(0.1))
    gen = sdr_amd.Gen("sine", False, 1.0, cps, 45.0, B)   # 45 / 360 = 2^-3: a whole multiple of 2^-23 (30 degrees is not)
    nb = 16
    buf = torch.zeros(nb * B, dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        gen.generate_device(nb, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    got = []
    for _ in range(3):
        graph.replay()
        torch.cuda.synchronize()
        got.append(buf.cpu().numpy().copy())
    got.append(gen.generate(5))   # the host entry point continues where the replays stopped
    ref, _ = O.gen_real(3 * nb + 5, B, 1.0, cps, float(np.float32(45.0) / np.float32(360.0)), 1)
    assert np.array_equal(np.concatenate(got), ref)

    slow = sdr_amd.Gen("sine", False, 1.0, float(np.float32(1e-5)), 0.0, B)
    quad = sdr_amd.Gen("square", True, 1.0, cps, 0.0, 1022)
    cbuf = torch.zeros(2 * nb * B, dtype=torch.float32, device="cuda")
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side):
        buf.zero_()
        with pytest.raises(sdr_amd.SdrError, match="HIP graph"):
            slow.generate_device(nb, buf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        with pytest.raises(sdr_amd.SdrError, match="HIP graph"):
            quad.generate_device(nb, cbuf.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    # uncaptured, both work as ever
    ref2, _ = O.gen_real(4, B, 1.0, float(np.float32(1e-5)), 0.0, 1)
    assert np.array_equal(slow.generate(4), ref2)
