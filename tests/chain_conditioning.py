"""Not collected by pytest: how the REFERENCE's own arithmetic (kfr::fir in fp32, one stage after the other:
oracle/_ref/ref_harness) stands against the double-precision oracle on the one ill-conditioned shape the 600-seed
soak of tests/test_chain_fuzz_gpu.py found (generated chain, seed 120: a near-DC cosine into a Hilbert stage,
whose null it sits in, then a 255-tap random filter of l2 norm 5.6).  CPU only; run here:
    python tests/chain_conditioning.py            ->  profiles/r03_chain_conditioning.txt
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O   # noqa: E402
import sdr_amd           # noqa: E402

REF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "ref_harness")


def ref_fir(taps, x):
    with tempfile.TemporaryDirectory() as d:
        tp = os.path.join(d, "taps.bin")
        np.asarray(taps, np.float32).tofile(tp)
        r = subprocess.run([REF, "fir", "real", tp, "4096"], input=np.asarray(x, np.float32).tobytes(), stdout=subprocess.PIPE, check=True)
    return np.frombuffer(r.stdout, np.float32)


def case(seed):
    rng = np.random.default_rng(8000 + seed)
    names = ["cosine", "sine", "square", "triangle", "sawtooth"]
    wf = names[seed % 5]
    amp = float(np.float32(rng.choice([1.0, 0.5, float(rng.uniform(0.05, 3.0))])))
    phase = float(np.float32(rng.choice([0.0, 45.0, float(rng.uniform(0, 360))])))
    cps = float(np.float32(rng.choice([0.1, 0.013, -0.2, float(rng.uniform(-0.45, 0.45))])))
    B = int(rng.choice([2044, 2040, 256, 1000, int(rng.integers(256, 5000))]))
    nt1 = int(rng.choice([127, 31, int(rng.integers(5, 300))]))
    nt2 = int(rng.choice([127, 255, int(rng.integers(1, 400)), int(rng.integers(2100, 4200))]))
    stages = [(sdr_amd.hilbert(nt1), (nt1 - 1) // 2 if rng.integers(0, 2) else 0),
              ((rng.uniform(-1, 1, nt2) * np.hanning(nt2 + 2)[1:-1]).astype(np.float32), int(rng.integers(0, min(nt2, 50))))]
    if rng.integers(0, 3) == 0:
        stages = stages[:1]
    calls = [int(v) for v in rng.choice([1, 2, 3, 17, 60, 200], int(rng.integers(1, 5)))]
    return wf, amp, phase, cps, B, stages, sum(calls)


def main():
    wf, amp, phase, cps, B, stages, total = case(120)
    phi0 = np.float32(phase) / np.float32(360.0)
    phi0 = float(phi0 - np.floor(phi0))
    x, _ = O.gen_real(total, B, amp, cps, phi0, sdr_amd.WAVEFORMS[wf])
    want = x
    got = x
    for taps, skip in stages:
        want = O.fir(taps, want)[skip:]          # double accumulation
        got = ref_fir(taps, got)[skip:]           # the reference's fp32 kfr::fir
    h = np.array([1.0])
    for taps, _ in stages:
        h = np.convolve(h, np.asarray(taps, np.float64))
    rms_out = float(np.sqrt(np.mean(want.astype(np.float64) ** 2)))
    rms_in = amp / np.sqrt(2.0)
    err = float(np.max(np.abs(got.astype(np.float64) - want)))
    print("generated chain fuzz, seed 120: %s amp %.4f cps %.6g, stages %s taps, %d samples" % (wf, amp, cps, [len(t) for t, _ in stages], len(x)))
    print("  l2 norm of the fused taps %.3f; input RMS %.4f; output RMS %.4f = %.3f of input RMS x l2 norm" % (np.sqrt(np.sum(h * h)), rms_in, rms_out, rms_out / (rms_in * np.sqrt(np.sum(h * h)))))
    print("  reference arithmetic (fp32 kfr::fir, staged) against the double-precision oracle: max error %.3e = %.3e of the output RMS" % (err, err / rms_out))


if __name__ == "__main__":
    main()
