"""Seeded fuzz of the `hilbert` block (blocks/bin/hilbert, the whole host pipeline: batching,
foreign packets riding in the batches, content switches, delay skip and header arithmetic)
against the oracle's restatement of blocks/hilbert.cpp:53-96, which tests/test_oracle_golden.py
pins to streams produced by the reference's own binary.  Random packet streams: three stream
ids, Signal / ComplexSignal / other contents, packets from one sample to several batches long,
a tail cut inside a payload (on a sample boundary) or inside a header; random tap count, delay
on or off, small and large batch sizes.  Headers byte-exact, samples within 1e-5 of the RMS."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "blocks", "bin")
TOL = 1e-5
N_SEEDS = int(os.environ.get("SDR_FUZZ_SEEDS", "24"))


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_random_packet_streams(seed):
    rng = np.random.default_rng(9000 + seed)
    taps = int(rng.choice([5, 6, 31, 127, 128, 255, int(rng.integers(5, 600))]))
    delay = bool(rng.integers(0, 2))
    my_id = int(rng.integers(0, 3))
    parts = []
    for i in range(int(rng.integers(1, 40))):
        pid = int(rng.integers(0, 3))
        kind = int(rng.choice([O.SIGNAL, O.SIGNAL, O.COMPLEX_SIGNAL, 1, 3, 8]))
        if kind in (O.SIGNAL, O.COMPLEX_SIGNAL):
            n = int(rng.choice([1, 2, int(rng.integers(1, 300)), int(rng.integers(300, 5000)), int(rng.integers(5000, 60000))]))
            w = 2 if kind == O.COMPLEX_SIGNAL else 1
            payload = rng.uniform(-1, 1, n * w).astype(np.float32).tobytes()
            dur = int(rng.integers(0, 10 ** 9))
        else:
            payload = bytes(rng.integers(0, 256, int(rng.integers(0, 100)), dtype=np.uint8))
            dur = int(rng.integers(0, 1000))
        parts.append((pid, kind, payload, dur))
    stream = b"".join(O.pack_packet(p, k, d, t) for p, k, d, t in parts)
    cut = int(rng.integers(0, 3))
    if cut == 1:       # inside the last header
        stream = stream[:len(stream) - len(parts[-1][2]) - int(rng.integers(1, 16))]
    elif cut == 2 and parts[-1][1] in (O.SIGNAL, O.COMPLEX_SIGNAL):
        elt = 8 if parts[-1][1] == O.COMPLEX_SIGNAL else 4
        keep = int(rng.integers(0, len(parts[-1][2]) // elt + 1)) * elt
        stream = stream[:len(stream) - len(parts[-1][2]) + keep]
    args = [str(my_id), str(taps)] + ([] if delay else ["delay=false"])
    env = dict(os.environ, SDR_AMD_BATCH_BYTES=str(int(rng.choice([65536, 262144, 8388608]))))
    p = subprocess.run([os.path.join(BIN, "hilbert")] + args, input=stream, capture_output=True, env=env, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    got = O.parse_packets(p.stdout)
    want = O.parse_packets(O.hilbert_block(stream, stream_id=my_id, taps=taps, delay=delay))
    assert [q[:4] for q in got] == [q[:4] for q in want], seed          # framing byte-exact, order kept
    fa, fb = [], []
    for a, b in zip(got, want):
        if a[0] == my_id and a[1] in (O.SIGNAL, O.COMPLEX_SIGNAL):
            assert len(a[4]) == len(b[4])
            fa.append(np.frombuffer(a[4], np.float32))
            fb.append(np.frombuffer(b[4], np.float32))
        else:
            assert a[4] == b[4]                                          # forwarded untouched
    if fb and sum(len(v) for v in fb) > 16:
        a, b = np.concatenate(fa).astype(np.float64), np.concatenate(fb).astype(np.float64)
        rms = np.sqrt(np.mean(b ** 2))
        # a stream shorter than the transformer's delay is all onset: its outputs lie far below the level the
        # block settles at (unit gain: the input's RMS), while the overlap-save kernel rounds at the level of
        # the whole filter -- there the bar is 1e-5 of a twentieth of the settled level (tests/test_fuzz_gpu.py)
        mine = [np.frombuffer(d, np.float32).astype(np.float64) for q, k, d, _ in parts
                if q == my_id and k in (O.SIGNAL, O.COMPLEX_SIGNAL) and len(d)]
        settled = float(np.sqrt(np.mean(np.concatenate(mine) ** 2))) if mine else 0.0
        if rms > 0:
            assert np.max(np.abs(a - b)) <= TOL * max(rms, 0.05 * settled), (seed, taps, delay, float(np.max(np.abs(a - b)) / rms))
