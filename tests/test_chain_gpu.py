"""GPU tests of the fused chain (sdr_chain_*) and the device generator (sdr_gen_*):
BASELINE configs 4 and 5 in small -- hilbert (127 taps, group-delay skip) followed by
a 127-tap lowpass, per channel, as one pass -- against the ORACLE evaluated stage by
stage exactly like a `hilbert | stream-filter` pipeline of reference blocks would."""
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


def staged_oracle(stages, x):
    """Stage by stage: FIR, then drop the first `skip` outputs (blocks/hilbert.cpp:75-86)."""
    y = x
    for taps, skip in stages:
        y = O.fir(taps, y)[skip:]
    return y


def run_chain(chain, x, cuts, cplx):
    import torch
    outs, p = [], 0
    for c in cuts:
        part = x[p:p + c]
        p += c
        if c == 0:
            continue
        h = part.view(np.float32).reshape(-1, 2) if cplx else part
        d = torch.from_numpy(np.ascontiguousarray(h)).cuda()
        out, n_out = chain.process_tensor(d)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        o = o.reshape(-1).view(np.complex64) if cplx else o
        outs.append(o[:n_out])
    return np.concatenate(outs)


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("cuts", [[30000], [10, 53, 1, 100, 25, 400, 29411], [63, 126, 189, 29622]])
def test_hilbert_then_lowpass_equals_staged_pipeline(cplx, cuts):
    rng = np.random.default_rng(17)
    n = sum(cuts)
    x = rng.uniform(-1, 1, n * (2 if cplx else 1)).astype(np.float32)
    x = x.view(np.complex64) if cplx else x
    stages = [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.1), 0)]
    chain = sdr_amd.Chain(stages, cplx)
    assert chain.ntaps == 253
    got = run_chain(chain, x, cuts, cplx)
    ref = staged_oracle(stages, x)
    assert got.shape == ref.shape          # N - 63 outputs: no sample dropped or duplicated
    assert rel_err(got, ref) <= TOL
    # the warm-up region is where a single 253-tap filter would be wrong
    assert rel_err(got[:200], ref[:200]) <= TOL


def test_three_stage_chain_with_two_skips():
    rng = np.random.default_rng(3)
    x = rng.uniform(-1, 1, 20000).astype(np.float32)
    stages = [(sdr_amd.hilbert(31), 15), (sdr_amd.fir_lowpass(63, 0.2), 31), (sdr_amd.fir_bandpass(255, 0.05, 0.2), 0)]
    chain = sdr_amd.Chain(stages, False)
    got = run_chain(chain, x, [5, 40, 3, 700, 19252], False)
    ref = staged_oracle(stages, x)
    assert got.shape == ref.shape and rel_err(got, ref) <= TOL
    chain.reset()
    got2 = run_chain(chain, x[:5000], [5000], False)
    assert rel_err(got2, ref[:len(got2)]) <= TOL


def test_multichannel_chain_config4_small():
    """Config C4 in small: independent channels, sines at f_c = (c+1)/160, hilbert then
    lowpass, channels laid out [channel][sample]."""
    import torch
    C, n = 8, 12000
    t = np.arange(n)
    x = np.stack([np.sin(2 * np.pi * (c + 1) / 160.0 * t) for c in range(C)]).astype(np.float32)
    stages = [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.1), 0)]
    chain = sdr_amd.Chain(stages, False, channels=C)
    d = torch.from_numpy(x).cuda()
    out, n_out = chain.process_tensor(d)
    torch.cuda.synchronize()
    got = out.cpu().numpy()[:, :n_out]
    assert n_out == n - 63
    for c in range(C):
        ref = staged_oracle(stages, x[c])
        assert np.max(np.abs(got[c] - ref)) <= TOL * np.sqrt(np.mean(ref.astype(np.float64) ** 2))


def test_gen_device_matches_oracle_recipe():
    """sdr_gen_*: bit-level restatement of the reference generator (polynomial sine,
    per-block fp32 phase recurrence) -- compared with the oracle's restatement, which
    is itself pinned to the reference's packets (tests/test_oracle_golden.py)."""
    B = sdr_amd.optimal_block_size(4, 1000000)
    assert B == 2044
    cps = float(np.float32(100e3) / np.float32(1e6))
    for wf, code in (("sine", 1), ("cosine", 0), ("square", 2), ("triangle", 3), ("sawtooth", 4)):
        g = sdr_amd.Gen(wf, False, 0.75, cps, 30.0, B)
        a = np.concatenate([g.generate(3), g.generate(70)])   # phase carries across calls
        phi0 = np.float32(30.0) / np.float32(360.0)
        ref, _ = O.gen_real(73, B, 0.75, cps, float(phi0), code)
        d = np.abs(a - ref)
        assert d.max() == 0.0, (wf, float(d.max()), int(np.count_nonzero(d)), int(np.argmax(d)))  # same fp32 recipe, no FMA: bit-exact
    Bc = sdr_amd.optimal_block_size(8, 1000000)
    g = sdr_amd.Gen("sine", True, 0.5, cps, 0.0, Bc)
    a = g.generate(5)
    ref, _ = O.gen_complex(5, Bc, 0.5, cps, 0.0, 1)
    assert np.max(np.abs(a - ref)) <= 3e-7


@pytest.mark.parametrize("wf,amp,phase,freq,rate,ht,preroll", [
    ("square", 1.0, 0.0, 100e3, 1000000, 127, True), ("triangle", 0.5, 30.0, 1e3, 48000, 31, True),
    ("sawtooth", 1.0, 0.0, -7e3, 250000, 127, True), ("triangle", 1.0, 0.0, 3e3, 250000, 255, False)])
def test_gen_complex_quadrature_waveforms(wf, amp, phase, freq, rate, ht, preroll):
    """sdr_gen_*: complex square / triangle / sawtooth = waveform + (+-j) * Hilbert FIR of the
    waveform advanced by the filter delay (blocks/gen.cpp:138-164, 231-260), phase and FIR
    history carried across calls; against the oracle's restatement (pinned to the reference's
    packets in tests/test_oracle_golden.py)."""
    B = sdr_amd.optimal_block_size(8, rate)
    cps = float(np.float32(freq) / np.float32(rate))
    g = sdr_amd.Gen(wf, True, amp, cps, phase, B)
    g.set_hilbert(ht, preroll)
    got = np.concatenate([g.generate(2), g.generate(1), g.generate(9)])
    phi0 = np.float32(phase) / np.float32(360.0)
    ref, _ = O.gen_complex_quad(12, B, amp, cps, float(phi0), sdr_amd.WAVEFORMS[wf], ht, preroll, acc64=True)
    rms = np.sqrt(np.mean(np.abs(ref.astype(np.complex128)) ** 2))
    assert np.max(np.abs(got - ref)) <= TOL * rms


def test_fused_gen_chain_config5_small():
    """Config C5 in small: generator -> hilbert(127, delay skip) -> lowpass(127) with the
    generated samples never leaving the GPU, chunk by chunk; compared with the oracle's
    generator + staged filters."""
    import torch
    B = 2044
    cps = float(np.float32(100e3) / np.float32(1e6))
    blocks, per_call = 24, 5
    gen = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
    stages = [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.12), 0)]
    chain = sdr_amd.Chain(stages, False)
    buf = torch.empty(per_call * B, dtype=torch.float32, device="cuda")
    outs = []
    done = 0
    stream = torch.cuda.current_stream().cuda_stream
    while done < blocks:
        nb = min(per_call, blocks - done)
        gen.generate_device(nb, buf.data_ptr(), stream=stream)
        out, n_out = chain.process_tensor(buf[: nb * B])
        torch.cuda.synchronize()
        outs.append(out[:n_out].cpu().numpy())
        done += nb
    got = np.concatenate(outs)
    x, _ = O.gen_real(blocks, B, 1.0, cps, 0.0, 1)
    ref = staged_oracle(stages, x)
    assert got.shape == ref.shape and rel_err(got, ref) <= TOL


FUSED_CASES = [
    # waveform, amp, phase, sample_rate (sets B), second-stage taps, calls (blocks each)
    ("sine", 1.0, 0.0, 1000000, 127, [40]),
    ("sine", 1.0, 0.0, 1000000, 127, [1, 2, 30, 7]),
    ("cosine", 0.75, 30.0, 48000, 255, [3, 37]),
    ("square", 1.0, 0.0, 1000000, 127, [25]),
    ("triangle", 0.5, 90.0, 1000000, 127, [25]),
    ("sawtooth", 1.0, 0.0, 1000000, 127, [25]),
    ("sine", 1.0, 0.0, 1000000, 4095, [6, 60]),     # two partitions: cached-spectrum kernel
    ("sine", 1.0, 0.0, 1000000, 9001, [70]),        # five partitions
]


@pytest.mark.parametrize("wf,amp,phase,rate,ntaps2,calls", FUSED_CASES)
def test_generator_fused_into_the_fir_kernel(wf, amp, phase, rate, ntaps2, calls):
    """sdr_chain_generate_device: gen -> hilbert(127, skip 63) -> FIR with the samples
    synthesised inside the FIR kernel.  Must equal the unfused device path (generate,
    then filter) and the oracle's generator + staged filters."""
    import torch
    B = sdr_amd.optimal_block_size(4, rate)
    cps = float(np.float32(100e3) / np.float32(rate)) if rate == 1000000 else float(np.float32(5e3) / np.float32(rate))
    taps2 = sdr_amd.fir_lowpass(ntaps2, 0.12) if ntaps2 < 1000 else sdr_amd.fir_bandpass(ntaps2, 0.05, 0.2)
    stages = [(sdr_amd.hilbert(127), 63), (taps2, 0)]
    total = sum(calls)
    stream = torch.cuda.current_stream().cuda_stream

    gen = sdr_amd.Gen(wf, False, amp, cps, phase, B)
    chain = sdr_amd.Chain(stages, False)
    out = torch.full((total * B,), 777.0, dtype=torch.float32, device="cuda")
    done = 0
    for nb in calls:
        done += chain.generate_device(gen, nb, out.data_ptr() + 4 * done, stream=stream)
    torch.cuda.synchronize()
    assert done == total * B - 63
    assert bool((out[done:] == 777.0).all())   # nothing written past the outputs
    fused = out[:done].cpu().numpy()

    gen2 = sdr_amd.Gen(wf, False, amp, cps, phase, B)
    chain2 = sdr_amd.Chain(stages, False)
    buf = torch.empty(max(calls) * B, dtype=torch.float32, device="cuda")
    parts = []
    for nb in calls:
        gen2.generate_device(nb, buf.data_ptr(), stream=stream)
        o, n_out = chain2.process_tensor(buf[: nb * B])
        torch.cuda.synchronize()
        parts.append(o[:n_out].cpu().numpy())
    unfused = np.concatenate(parts)
    assert fused.shape == unfused.shape
    e_fu = rel_err(fused, unfused)
    # same stream, same tiles; the fused kernels evaluate the sine behind the (bit-exact) phase argument with fused
    # multiply-adds (gen_device.h: gen_sine_pair_fast), so samples differ by an ulp or two: measured 0.7-1.03e-6 of
    # the output RMS here, with the distance to the oracle unchanged (0.65-1.0e-6 either way; bar TOL = 1e-5)
    assert e_fu <= 2e-6, e_fu

    x, _ = O.gen_real(total, B, amp, cps, float(np.float32(phase) / np.float32(360.0)), sdr_amd.WAVEFORMS[wf])
    ref = staged_oracle(stages, x)
    e_ref = rel_err(fused, ref)
    print("fused vs unfused %.3g, fused vs oracle %.3g, unfused vs oracle %.3g" % (e_fu, e_ref, rel_err(unfused, ref)))
    assert fused.shape == ref.shape and e_ref <= TOL


@pytest.mark.parametrize("B,nblocks", [(256, 300), (257, 200), (511, 150), (4096, 30), (5000, 20), (65536, 3), (1 << 20, 2)])
@pytest.mark.parametrize("wf,cps", [("cosine", -0.2137), ("sine", 0.0931), ("triangle", 0.013)])
def test_fused_generator_block_sizes(B, nblocks, wf, cps):
    """The fused generator's walk over a tile: generator blocks shorter than a row step of 256 samples
    would wrap more than once per step (256 is the smallest the fusion takes), blocks of a few rows wrap
    every few rows and inside waves, blocks longer than a tile never wrap within one.  Two calls, so the
    phase table of the second starts where the first one's recurrence ended.  Against the oracle's
    generator (per-block fp32 phase recurrence, gen.cpp:185-227) + staged filters."""
    import torch
    stages = [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.12), 0)]
    amp, phase = 0.8, 77.0
    gen = sdr_amd.Gen(wf, False, amp, float(np.float32(cps)), phase, B)
    chain = sdr_amd.Chain(stages, False)
    calls = [nblocks - nblocks // 3, nblocks // 3] if nblocks >= 3 else [nblocks]
    total = sum(calls)
    out = torch.full((total * B,), 777.0, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    done = 0
    for nb in calls:
        done += chain.generate_device(gen, nb, out.data_ptr() + 4 * done, stream=stream)
    torch.cuda.synchronize()
    assert done == total * B - 63
    assert bool((out[done:] == 777.0).all())
    phi0 = np.float32(phase) / np.float32(360.0)
    x, _ = O.gen_real(total, B, amp, float(np.float32(cps)), float(phi0 - np.floor(phi0)), sdr_amd.WAVEFORMS[wf])
    ref = staged_oracle(stages, x)
    got = out[:done].cpu().numpy()
    rms = np.sqrt(np.mean(ref.astype(np.float64) ** 2))
    err = float(np.max(np.abs(got.astype(np.float64) - ref)))
    assert err <= TOL * max(rms, amp / np.sqrt(2.0) * 0.05), (B, wf, err / rms)


@pytest.mark.parametrize("B", [256, 2044])
def test_fused_generator_retuned_and_advanced_between_calls(B):
    """Frequency changes and skipped packets between calls of the fused chain: the host's phase table switches
    between its two forms (closed form when phase and increment are multiples of 2^-23, the serial fp32
    recurrence otherwise -- a frequency of 1e-5 cycles per sample gives an increment with finer bits), and
    every segment must continue the phase the previous one left (gen.cpp:185-227, :124 for the increment)."""
    import torch
    stages = [(sdr_amd.hilbert(31), 15), (sdr_amd.fir_lowpass(63, 0.2), 0)]
    amp, phase = 1.25, 30.0
    plan = [(0.0931, 40, 0), (1e-5, 25, 3), (-0.2137, 33, 0), (3.3e-4, 20, 7), (0.125, 30, 1)]  # cps, blocks, packets skipped first
    gen = sdr_amd.Gen("sine", False, amp, float(np.float32(plan[0][0])), phase, B)
    chain = sdr_amd.Chain(stages, False)
    total = sum(nb for _, nb, _ in plan)
    out = torch.full((total * B,), 777.0, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    phi = np.float32(phase) / np.float32(360.0)
    phi = float(phi - np.floor(phi))
    xs, done = [], 0
    for cps, nb, skip in plan:
        cps = float(np.float32(cps))
        gen.set_frequency(cps)
        if skip:
            gen.advance(skip)
            _, phi = O.gen_real(skip, B, amp, cps, phi, sdr_amd.WAVEFORMS["sine"])
        done += chain.generate_device(gen, nb, out.data_ptr() + 4 * done, stream=stream)
        x, phi = O.gen_real(nb, B, amp, cps, phi, sdr_amd.WAVEFORMS["sine"])
        xs.append(x)
    torch.cuda.synchronize()
    ref = staged_oracle(stages, np.concatenate(xs))
    assert done == ref.shape[0] and bool((out[done:] == 777.0).all())
    got = out[:done].cpu().numpy()
    rms = np.sqrt(np.mean(ref.astype(np.float64) ** 2))
    assert float(np.max(np.abs(got.astype(np.float64) - ref))) <= TOL * rms


def test_generator_fusion_rejects_what_it_cannot_do():
    B = 2044
    gen_c = sdr_amd.Gen("sine", True, 1.0, 0.1, 0.0, 1022)
    gen_small = sdr_amd.Gen("sine", False, 1.0, 0.1, 0.0, 100)
    gen = sdr_amd.Gen("sine", False, 1.0, 0.1, 0.0, B)
    stages = [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.1), 0)]
    import torch
    out = torch.empty(4 * B, dtype=torch.float32, device="cuda")
    for g, ch in ((gen_c, sdr_amd.Chain(stages, False)), (gen_small, sdr_amd.Chain(stages, False)),
                  (gen, sdr_amd.Chain(stages, False, channels=2))):
        with pytest.raises(sdr_amd.SdrError):
            ch.generate_device(g, 2, out.data_ptr())
    # a chain short enough for the direct-form kernel moves to the overlap-save kernels on its
    # first generated packet -- unless it has already filtered samples the other way
    short = sdr_amd.Chain([(sdr_amd.hilbert(31), 15)], False)
    assert short.generate_device(gen, 2, out.data_ptr()) == 2 * B - 15
    used = sdr_amd.Chain([(sdr_amd.hilbert(31), 15)], False)
    used.process_tensor(torch.zeros(1000, dtype=torch.float32, device="cuda"))
    with pytest.raises(sdr_amd.SdrError, match="already filtered"):
        used.generate_device(gen, 2, out.data_ptr())
    used.reset()
    assert used.generate_device(sdr_amd.Gen("sine", False, 1.0, 0.1, 0.0, B), 2, out.data_ptr()) == 2 * B - 15


@pytest.mark.parametrize("world", [2, 3])
def test_config5_sharding_recipe_on_one_gpu(world):
    """BASELINE config 5's multi-GPU cut, rehearsed on one GPU: every "rank" positions
    its own generator and chain (Gen.advance, Chain.seek), primes with its halo packets
    and produces its share; the concatenation equals the stream produced in one go --
    no sample dropped, duplicated or reordered at the chunk boundaries."""
    import torch
    from sdr_amd import sharding
    B = 2044
    cps = float(np.float32(100e3) / np.float32(1e6))
    stages = [(sdr_amd.hilbert(127), 63), (sdr_amd.fir_lowpass(127, 0.1), 0)]
    n_blocks = 50
    stream = torch.cuda.current_stream().cuda_stream

    def produce(first, blocks, halo):
        gen = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
        chain = sdr_amd.Chain(stages, False)
        out = torch.empty((blocks + halo) * B, dtype=torch.float32, device="cuda")
        if first:
            gen.advance(first - halo)
            chain.seek((first - halo) * B)
            n0 = chain.generate_device(gen, halo, out.data_ptr(), stream=stream)   # priming: discarded
            assert n0 == halo * B
        n1 = chain.generate_device(gen, blocks, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        return out[:n1].cpu().numpy()

    whole = produce(0, n_blocks, 0)
    assert len(whole) == n_blocks * B - 63
    plan = sharding.plan_generated_chain(n_blocks, world, B, 253)
    parts = [produce(*p) for p in plan]
    got = np.concatenate(parts)
    assert got.shape == whole.shape
    # two fp32 results whose tiles fall at different offsets: each is within ~6e-7 of the
    # exact answer, so they may differ from one another by twice that
    assert rel_err(got, whole) <= 3e-6


def test_short_chain_seek_then_generate():
    """A chain whose fused filter is below the overlap-save crossover (15 taps: direct form) on a
    rank that does not own the start of the stream: Chain.seek zeroes the histories, so
    generate_device may still move the filter to the overlap-save kernels (where the generator
    lives).  The seeked rank's outputs equal the same packets of the stream produced in one go;
    once the chain has filtered samples through the direct form, the move is refused."""
    import torch
    B = 2044
    cps = float(np.float32(100e3) / np.float32(1e6))
    stages = [(sdr_amd.hilbert(9), 4), (sdr_amd.fir_lowpass(7, 0.2), 0)]
    stream = torch.cuda.current_stream().cuda_stream

    def produce(first, blocks, halo):
        gen = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
        chain = sdr_amd.Chain(stages, False)
        assert chain.ntaps == 15
        out = torch.empty((blocks + halo) * B, dtype=torch.float32, device="cuda")
        if first:
            assert (first - halo) * B >= chain.min_seek
            gen.advance(first - halo)
            chain.seek((first - halo) * B)
            assert chain.generate_device(gen, halo, out.data_ptr(), stream=stream) == halo * B
        n1 = chain.generate_device(gen, blocks, out.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        return out[:n1].cpu().numpy()

    whole = produce(0, 12, 0)
    assert len(whole) == 12 * B - 4
    tail = produce(5, 7, 1)
    assert len(tail) == 7 * B
    assert rel_err(tail, whole[5 * B - 4:]) <= 3e-6

    # the documented bound: min_seek is accepted, one less is not
    chain = sdr_amd.Chain(stages, False)
    chain.seek(chain.min_seek)
    with pytest.raises(sdr_amd.SdrError):
        chain.seek(chain.min_seek - 1)
    # a short chain that has already run on the direct-form kernels cannot be moved
    chain.reset()
    x = torch.zeros(5000, dtype=torch.float32, device="cuda")
    chain.process_tensor(x)
    gen = sdr_amd.Gen("sine", False, 1.0, cps, 0.0, B)
    with pytest.raises(sdr_amd.SdrError):
        chain.generate_device(gen, 1, x.data_ptr(), stream=stream)
