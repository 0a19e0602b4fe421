#!/usr/bin/env python3
"""bench.py -- throughput of the FIR hot path on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2]

Default workload = BASELINE.json configs[1]: stream_filter's 127-tap real-tap
lowpass on 2^28 complex-float samples, resident in HBM, one MI355X.  A "step" is
one pass of the FIR over that 2^28-sample buffer (the stream continues from step
to step: the filter history carries).  With N > 1 (launched by torch.distributed.run,
one rank per GPU) the stream is N * 2^28 samples cut into contiguous chunks, one per
GPU, each primed with its left halo (ntaps-1 samples regenerated from the
counter-based input), no data-path collective: weak scaling.

Prints ONE JSON line (rank 0).  `value` = samples all ranks filtered / max-over-ranks
time, inputs already in HBM.  `roofline` prices the dominant kernel against HBM
(algorithmic bytes: 16 B per complex sample = 8 read + 8 written).
`cpu_baseline` times the reference's own kfr::fir (oracle/_ref/ref_harness, built
from /root/reference in the build container) on a bounded sample on one host core.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
USE_DIST = False       # set in main(): a torch.distributed process group is up

WORKLOADS = {
    # name: (description, complex?, taps builder, log2 samples per GPU, channels)
    "c2": ("stream_filter 127-tap complex-float lowpass on 2^28 samples, in-memory (BASELINE configs[1])",
           True, lambda S: S.fir_lowpass(127, 0.1), 28, 1),
    "c1": ("hilbert default taps (127) on 2^24 real float32 samples, in-memory (BASELINE configs[0] kernel-only)",
           False, lambda S: S.hilbert(127), 24, 1),
    "c1big": ("hilbert 127 taps on 2^28 real float32 samples, in-memory", False, lambda S: S.hilbert(127), 28, 1),
    "c2real": ("127-tap lowpass on 2^28 real float32 samples, in-memory", False, lambda S: S.fir_lowpass(127, 0.1), 28, 1),
    "hilbc": ("hilbert 127 taps on 2^28 complex samples, in-memory", True, lambda S: S.hilbert(127), 28, 1),
    "c3": ("stream_filter 4095-tap bandpass on 2^26 complex samples (partitioned overlap-save), in-memory "
           "(BASELINE configs[2])", True, lambda S: S.fir_bandpass(4095, 0.1, 0.2), 26, 1),
    "c3real": ("4095-tap bandpass on 2^27 real samples (partitioned overlap-save), in-memory",
               False, lambda S: S.fir_bandpass(4095, 0.1, 0.2), 27, 1),
    "c4": ("64 channels x 2^24 real samples, hilbert 127 taps (BASELINE configs[3] first stage)",
           False, lambda S: S.hilbert(127), 24, 64),
    "lp31": ("31-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(31, 0.1), 28, 1),
    "lp15": ("15-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(15, 0.1), 28, 1),
    "lp63r": ("63-tap lowpass on 2^28 real samples, in-memory", False, lambda S: S.fir_lowpass(63, 0.1), 28, 1),
    "lp255": ("255-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(255, 0.1), 28, 1),
    # fused chains (sdr_chain_*): handled by bench_chain() below
    "c4chain": ("64 channels x 2^24 real samples, hilbert(127, delay skip) + lowpass(127) fused in one pass "
                "(BASELINE configs[3])", False, None, 24, 64),
    "c5": ("fused gen(sine) -> hilbert(127) -> lowpass(127): 2^29 real outputs per GPU, samples synthesised inside "
           "the FIR kernel (BASELINE configs[4], per-GPU share)", False, None, 29, 1),
}


def bench_chain(args, S, torch, dist, rank, local_rank, world, dev):
    """c4chain / c5: per step, every rank pushes its share through the fused chain."""
    desc, cplx, _, log2n, channels = WORKLOADS[args.workload]
    n = 1 << log2n
    stages = [(S.hilbert(127), 63), (S.fir_lowpass(127, 0.1), 0)]
    chain = S.Chain(stages, False, channels=channels, device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream
    if args.workload == "c4chain":
        # channel c of the global set: sine at f = (c+1)/160 cycles/sample
        t = torch.arange(n, device=dev, dtype=torch.float64)
        x = torch.empty((channels, n), dtype=torch.float32, device=dev)
        for c in range(channels):
            x[c] = torch.sin(2 * np.pi * ((rank * channels + c) % 64 + 1) / 160.0 * t).to(torch.float32)
        del t
        y = torch.empty_like(x)

        def step():
            chain.process_tensor(x, y)
        bytes_per_step = n * channels * 8
        units = n * channels
    else:
        B = S.optimal_block_size(4, 1000000)
        blocks_per_chunk = (1 << args.chunk_log2) // B
        chunk = blocks_per_chunk * B
        nchunks = n // chunk
        gen = S.Gen("sine", False, 1.0, float(np.float32(100e3) / np.float32(1e6)), 0.0, B, device=local_rank)
        buf = torch.empty(chunk if args.unfused else 1, dtype=torch.float32, device=dev)
        y = torch.empty(nchunks * chunk, dtype=torch.float32, device=dev)

        if args.unfused:
            def step():
                for i in range(nchunks):
                    gen.generate_device(blocks_per_chunk, buf.data_ptr(), stream=stream)
                    chain.process_device(buf.data_ptr(), chunk, y.data_ptr() + 4 * i * chunk, stream=stream)
        else:
            # the generator runs inside the FIR kernel: 4 bytes written per output, none read
            # rank r owns packets [first, first + blocks) of the world * 2^29-sample stream:
            # generator and chain positioned there, primed with the halo packets
            from sdr_amd import sharding
            first, _, halo = sharding.plan_generated_chain(world * nchunks * blocks_per_chunk, world, B, 253)[rank]
            if first:
                gen.advance(first - halo)
                chain.seek((first - halo) * B)
                chain.generate_device(gen, halo, y.data_ptr(), stream=stream)   # outputs discarded
            else:
                chain.generate_device(gen, 8, y.data_ptr(), stream=stream)  # start-of-stream warm-up packets
            for _ in range(4):  # every slot of the generator's phase-table ring at its working size
                chain.generate_device(gen, blocks_per_chunk, y.data_ptr(), stream=stream)

            def step():
                for i in range(nchunks):
                    chain.generate_device(gen, blocks_per_chunk, y.data_ptr() + 4 * i * chunk, stream=stream)
        bytes_per_step = nchunks * chunk * 4
        units = nchunks * chunk
    name = "sdr_chain(253 fused taps) -> fir_fft4096<REAL%s>" % (
        ", GEN" if args.workload == "c5" and not args.unfused else "")
    return step, units, bytes_per_step, name, 253, desc



def fill_counter_based(x, first_sample, seed=0x5D12):
    """Uniform [-1,1) input defined by a counter-based generator: the value of
    float k of sample s depends only on (seed, s, k), so any rank can regenerate
    any part of the global stream (its chunk and the halo before it)."""
    import torch
    n = x.shape[0]
    w = x.shape[1]
    step = 1 << 24
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        idx = torch.arange(first_sample + lo, first_sample + hi, device=x.device, dtype=torch.int64)
        for k in range(w):
            v = (idx * w + k + seed * 0x9E3779B9) & 0xFFFFFFFF
            v = (v ^ (v >> 16)) * 0x45D9F3B & 0xFFFFFFFF
            v = (v ^ (v >> 16)) * 0x45D9F3B & 0xFFFFFFFF
            v = v ^ (v >> 16)
            x[lo:hi, k] = (v >> 8).to(torch.float32) * (2.0 / (1 << 24)) - 1.0


def cpu_baseline(taps, cplx, log2n):
    """Time the reference's kfr::fir (oracle/_ref/ref_harness) on one core over a
    bounded sample of the same workload; falls back to the C port of the oracle."""
    n = 1 << log2n
    rng = np.random.default_rng(0x5D12)
    x = rng.uniform(-1, 1, n * (2 if cplx else 1)).astype(np.float32)
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    what = "first 2^%d %s samples of the workload's stream shape, %d taps, 1 process" % (
        log2n, "complex" if cplx else "real", len(taps))
    if os.path.exists(harness):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            tp, xp = os.path.join(d, "taps.bin"), os.path.join(d, "x.bin")
            np.asarray(taps, np.float32).tofile(tp)
            x.tofile(xp)
            with open(xp, "rb") as fin, open(os.devnull, "wb") as fout:
                t0 = time.perf_counter()
                subprocess.run([harness, "fir", "complex" if cplx else "real", tp, "65536"],
                               stdin=fin, stdout=fout, check=True)
                dt = time.perf_counter() - t0
        return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
                "sample": what + " (reference kfr::fir via oracle/_ref/ref_harness, g++ -O3 -ffast-math -DNDEBUG)"}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    f = O.Fir(taps, cplx, acc64=False)
    xs = x.view(np.complex64) if cplx else x
    t0 = time.perf_counter()
    f.process(xs)
    dt = time.perf_counter() - t0
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": what + " (oracle/sdr_oracle.c fp32 port; oracle/_ref not present)"}


_REAL_STDOUT = None


def keep_stdout_for_the_json_line():
    """Libraries write to stdout behind Python's back (RCCL prints a five-line version banner
    on rank 0 when its communicator comes up).  The contract is ONE JSON line on stdout: fd 1
    is pointed at stderr for the run, and the JSON line goes to the saved descriptor."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def emit_json_line(obj):
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        os.write(1, line)
    else:
        os.write(_REAL_STDOUT, line)


def main():
    keep_stdout_for_the_json_line()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--path", default="auto", choices=["auto", "direct", "fft", "fft_once"],
                    help="experiments: pin the kernel family (the default is the library's own choice)")
    ap.add_argument("--cpu-log2n", type=int, default=28)
    ap.add_argument("--chunk-log2", type=int, default=29, help="c5: samples per call")
    ap.add_argument("--unfused", action="store_true", help="c5: generator kernel + chain kernel through an HBM buffer")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sdr_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libsdr_amd has no CPU path")
    # one rank per GPU; SDR_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with
    # fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("SDR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    global USE_DIST
    # SDR_BENCH_FORCE_DIST=1: go through the process group even with one rank (rehearses the
    # RCCL calls of the multi-GPU run on a one-GPU box)
    USE_DIST = world > 1 or bool(os.environ.get("SDR_BENCH_FORCE_DIST"))
    if USE_DIST:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    desc, cplx, mk_taps, log2n, channels = WORKLOADS[args.workload]
    if mk_taps is None:
        step, units, step_bytes, kernel_name, ntaps, desc = bench_chain(args, S, torch, dist, rank, local_rank, world, dev)
        run_and_report(args, torch, dist, rank, world, dev, step, units, step_bytes, kernel_name, ntaps, desc,
                       "float32", channels, None)
        return
    taps = mk_taps(S)
    n = 1 << log2n
    w = 2 if cplx else 1
    st = S.FirState(taps, cplx, channels=channels, device=local_rank, path=args.path)

    from sdr_amd import sharding
    x = torch.empty((channels * n, w), dtype=torch.float32, device=dev)
    # weak scaling: the global stream is world * n samples (or world * channels
    # channels); this rank owns one contiguous chunk (one group of channels)
    first, count = sharding.plan_chunks(world * n * channels, world, n * channels)[rank]
    assert count == n * channels
    fill_counter_based(x, first)
    y = torch.empty_like(x)
    if channels == 1:
        lo, hl = sharding.halo_range(first, len(taps))
        if hl:
            # prime the history with the left halo of this rank's chunk, regenerated
            # from the counter-based source (no neighbour exchange)
            halo = torch.empty((hl, w), dtype=torch.float32, device=dev)
            fill_counter_based(halo, lo)
            st.process_tensor(halo.view(-1) if not cplx else halo)
    xin = x if cplx else x.view(channels, n) if channels > 1 else x.view(-1)
    yout = y if cplx else y.view(channels, n) if channels > 1 else y.view(-1)
    if cplx and channels > 1:
        xin, yout = x.view(channels, n, 2), y.view(channels, n, 2)

    def step():
        st.process_tensor(xin, yout)  # launches on torch's current stream
    run_and_report(args, torch, dist, rank, world, dev, step, n * channels, n * channels * 8 * w, st.kernel_name,
                   int(len(taps)), desc, "complex64" if cplx else "float32", channels, (taps, cplx, log2n))


def run_and_report(args, torch, dist, rank, world, dev, step, units, step_bytes, kernel_name, ntaps, desc, kind,
                   channels, cpu_spec):
    """W warm-up steps, K timed steps between barrier + synchronize, max over ranks, one JSON line."""
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if USE_DIST:
        dist.barrier()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if USE_DIST:
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # device time per step: HIP events on the launch stream (torch's current stream is
    # the stream every kernel of the step is launched on)
    kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
    kern_ms_avg = float(np.mean(kern_ms))

    if rank == 0:
        value = units * world * args.steps / elapsed / 1e6
        achieved = step_bytes / (kern_ms_avg * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s (FIR hot path: %s)" % args.workload,
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "samples_per_gpu": units, "channels": channels,
                       "taps": ntaps, "sample_kind": kind, "kernel": kernel_name,
                       "sharding": "contiguous chunks with regenerated halo / channel groups, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel_ms": kern_ms_avg, "kernel_ms_min": float(np.min(kern_ms)),
                         "algorithmic_bytes_per_launch": step_bytes,
                         # what a direct-form evaluation of the same outputs would need
                         # (2*taps flop per real MAC lane); the FFT path does ~4x fewer flops
                         "direct_form_equivalent_tflops": 2.0 * ntaps * units * (2 if kind == "complex64" else 1) / (kern_ms_avg * 1e-3) / 1e12},
        }
        tj = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tj):
            with open(tj) as f:
                out["roofline"]["traffic"] = json.load(f).get("hbm_bytes_per_launch")
        if not args.no_cpu_baseline and world == 1 and cpu_spec is not None:
            taps, cplx, log2n = cpu_spec
            out["cpu_baseline"] = cpu_baseline(taps, cplx, min(args.cpu_log2n, log2n))
        else:
            out["cpu_baseline"] = None
        emit_json_line(out)
    if USE_DIST:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
