"""Seeded fuzz of the device generator (sdr_gen_*) against the oracle's restatement of
blocks/gen.cpp (pinned to the reference's packets in tests/test_oracle_golden.py): every
waveform, real and complex, random amplitude / phase / frequency (negative and above Nyquist
included) / block size, the stream produced in calls of random length, with `advance`
(packets skipped) and a mid-stream frequency change.  Real waveforms are the same fp32 recipe
without FMA contraction: bit-exact.  Complex sine/cosine: accurate sincos of the faithfully
built fp32 angle (1 ulp of the amplitude); complex square/triangle/sawtooth: through the
Hilbert FIR, 1e-5 of the RMS."""
import os

import numpy as np
import pytest

import oracle_lib as O
import sdr_amd

pytestmark = pytest.mark.gpu
TOL = 1e-5
N_SEEDS = int(os.environ.get("SDR_FUZZ_SEEDS", "40"))
NAMES = ["cosine", "sine", "square", "triangle", "sawtooth"]


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_generators(seed):
    rng = np.random.default_rng(11000 + seed)
    wf = NAMES[seed % 5]
    cplx = bool((seed // 5) % 2)
    code = sdr_amd.WAVEFORMS[wf]
    amp = float(np.float32(rng.choice([1.0, 0.5, 0.75, float(rng.uniform(0.01, 4.0))])))
    phase = float(np.float32(rng.choice([0.0, 30.0, 90.0, float(rng.uniform(-720, 720))])))
    cps = float(np.float32(rng.choice([0.1, 0.001, -0.07, 0.49, float(rng.uniform(-0.9, 0.9))])))
    B = int(rng.choice([2044 if not cplx else 1022, 2040, 256, 300, int(rng.integers(256, 4097))]))
    calls = [int(v) for v in rng.integers(1, 12, int(rng.integers(1, 5)))]
    total = sum(calls)
    phi0 = float(np.float32(phase) / np.float32(360.0))
    phi0 = float(np.float32(phi0) - np.floor(np.float32(phi0)))     # fract(phase / 360), gen.cpp:123
    g = sdr_amd.Gen(wf, cplx, amp, cps, phase, B)
    quad = cplx and code >= 2
    ht = int(rng.choice([31, 127, 255]))
    if quad:
        g.set_hilbert(ht, True)
    got = np.concatenate([g.generate(nb) for nb in calls])
    if not cplx:
        ref, _ = O.gen_real(total, B, amp, cps, phi0, code)
        d = np.abs(got - ref)
        assert d.max() == 0.0, (wf, amp, phase, cps, B, float(d.max()), int(np.argmax(d)))
        # skipping packets lands on the same samples
        skip = int(rng.integers(0, total))
        g2 = sdr_amd.Gen(wf, False, amp, cps, phase, B)
        g2.advance(skip)
        assert np.array_equal(g2.generate(total - skip), ref[skip * B:])
    elif not quad:
        ref, _ = O.gen_complex(total, B, amp, cps, phi0, code)
        assert np.max(np.abs(got - ref)) <= 4e-7 * max(1.0, abs(amp)), (wf, amp, phase, cps, B)
    else:
        ref, _ = O.gen_complex_quad(total, B, amp, cps, phi0, code, ht, True, acc64=True)
        rms = np.sqrt(np.mean(np.abs(ref.astype(np.complex128)) ** 2))
        assert np.max(np.abs(got - ref)) <= TOL * rms, (wf, amp, phase, cps, B, ht)
