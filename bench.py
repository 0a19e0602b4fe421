#!/usr/bin/env python3
"""bench.py -- throughput of the FIR hot path on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2]

Default workload = BASELINE.json configs[1]: stream_filter's 127-tap real-tap
lowpass on 2^28 complex-float samples, resident in HBM, one MI355X.  A "step" is
one pass of the FIR over that 2^28-sample buffer (the stream continues from step
to step: the filter history carries).

N > 1: one process per GPU over RCCL.  Either something else launched the ranks
(torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE in the environment), or --
`python bench.py --gpus N` on its own -- this process, which never touches a GPU,
spawns the N ranks itself and rank 0 prints the JSON line.  Sharding (no data-path
collective, sdr_amd/sharding.py):
  * long-stream workloads (c2, c3, c5 ...): WEAK scaling -- the stream is N * 2^28
    samples cut into contiguous chunks, one per GPU, each primed with its left halo
    regenerated from the counter-based source;
  * 64-channel workloads (c4, c4chain = BASELINE configs[3]): STRONG scaling -- the 64
    channels are dealt 64/N per GPU (sharding.plan_channels); `value_with_gather`
    also times the optional gather of every rank's outputs on rank 0.

Prints ONE JSON line (rank 0).  `value` = samples all ranks filtered / max-over-ranks
time, inputs already in HBM.  `roofline` prices the dominant kernel against both
rooflines: achieved_hbm = algorithmic bytes / kernel time / 8 TB/s,
achieved_valu = the kernel's own fp32 operations / kernel time / 157.3 TFLOP/s; `bound`
names the larger fraction, `achieved` / `peak` / `frac` are that roofline's.
`cpu_baseline` times the reference's own kfr::fir (oracle/_ref/ref_harness, built from
/root/reference in the build container) on a bounded sample: one process on one core,
plus the aggregate of P independent processes (the reference's only parallelism).
"""
import argparse
import json
import os
import socket
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 vector (= FP32 matrix) rate
USE_DIST = False          # set in main(): a torch.distributed process group is up
RAMP_S = 0.2              # untimed clock-ramp prelude ahead of the warm-up steps (reported in the JSON line)

WORKLOADS = {
    # name: (description, complex?, taps builder, log2 samples per channel, channels, scaling)
    "c2": ("stream_filter 127-tap complex-float lowpass on 2^28 samples, in-memory (BASELINE configs[1])",
           True, lambda S: S.fir_lowpass(127, 0.1), 28, 1, "weak"),
    "c1": ("hilbert default taps (127) on 2^24 real float32 samples, in-memory (BASELINE configs[0] kernel-only)",
           False, lambda S: S.hilbert(127), 24, 1, "weak"),
    "c1big": ("hilbert 127 taps on 2^28 real float32 samples, in-memory", False, lambda S: S.hilbert(127), 28, 1, "weak"),
    "c2real": ("127-tap lowpass on 2^28 real float32 samples, in-memory", False, lambda S: S.fir_lowpass(127, 0.1), 28, 1, "weak"),
    "hilbc": ("hilbert 127 taps on 2^28 complex samples, in-memory", True, lambda S: S.hilbert(127), 28, 1, "weak"),
    "c3": ("stream_filter 4095-tap bandpass on 2^26 complex samples (partitioned overlap-save), in-memory "
           "(BASELINE configs[2])", True, lambda S: S.fir_bandpass(4095, 0.1, 0.2), 26, 1, "weak"),
    "c3real": ("4095-tap bandpass on 2^27 real samples (partitioned overlap-save), in-memory",
               False, lambda S: S.fir_bandpass(4095, 0.1, 0.2), 27, 1, "weak"),
    "c4": ("64 channels x 2^24 real samples sharded over the GPUs, hilbert 127 taps (BASELINE configs[3] first stage)",
           False, lambda S: S.hilbert(127), 24, 64, "strong"),
    "lp31": ("31-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(31, 0.1), 28, 1, "weak"),
    "lp15": ("15-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(15, 0.1), 28, 1, "weak"),
    "lp17": ("17-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(17, 0.1), 28, 1, "weak"),
    "lp41": ("41-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(41, 0.1), 28, 1, "weak"),
    "lp15r": ("15-tap lowpass on 2^28 real samples, in-memory", False, lambda S: S.fir_lowpass(15, 0.1), 28, 1, "weak"),
    "lp5": ("5-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(5, 0.1), 28, 1, "weak"),
    "lp31r": ("31-tap lowpass on 2^28 real samples, in-memory", False, lambda S: S.fir_lowpass(31, 0.1), 28, 1, "weak"),
    "lp47r": ("47-tap lowpass on 2^28 real samples, in-memory", False, lambda S: S.fir_lowpass(47, 0.1), 28, 1, "weak"),
    "hilb63": ("hilbert 63 taps on 2^28 real samples, in-memory", False, lambda S: S.hilbert(63), 28, 1, "weak"),
    "hilb95": ("hilbert 95 taps on 2^28 real samples, in-memory", False, lambda S: S.hilbert(95), 28, 1, "weak"),
    "lp49": ("49-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(49, 0.1), 28, 1, "weak"),
    "lp65": ("65-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(65, 0.1), 28, 1, "weak"),
    "lp63r": ("63-tap lowpass on 2^28 real samples, in-memory", False, lambda S: S.fir_lowpass(63, 0.1), 28, 1, "weak"),
    "lp255": ("255-tap lowpass on 2^28 complex samples, in-memory", True, lambda S: S.fir_lowpass(255, 0.1), 28, 1, "weak"),
    # fused chains (sdr_chain_*): handled by bench_chain() below
    "c4chain": ("64 channels x 2^24 real samples sharded over the GPUs, hilbert(127, delay skip) + lowpass(127) "
                "fused in one pass (BASELINE configs[3])", False, None, 24, 64, "strong"),
    # BASELINE configs[0] as it is stated: the drop-in BLOCKS through a unix pipe (bench_pipe() below; not a kernel timing)
    "c1pipe": ("gen sine -> hilbert (default taps), 2^24 float32 samples, through a unix pipe: blocks/bin/gen | head -c | "
               "blocks/bin/hilbert, the reference's own blocks beside it (BASELINE configs[0])", False, None, 24, 1, "weak"),
    "c5": ("fused gen(sine) -> hilbert(127) -> lowpass(127): 2^29 real outputs per GPU, samples synthesised inside "
           "the FIR kernel (BASELINE configs[4], per-GPU share)", False, None, 29, 1, "weak"),
}


# ---- what the kernels compute (for achieved_valu) ---------------------------------------
# fp32 operations the kernel really executes per sample, counted from the code:
#   overlap-save, N = 4096, 16 points per lane: a 16-point DFT in registers = 8 radix-4
#   butterflies (16 adds each) + 8 constant twiddles (6 flops) + one +-i rotation = 178 flops;
#   three of them per transform + 15 twiddle products (6 flops) twice between stages = 714 per
#   lane per transform; forward + inverse + 16 spectrum products = 1524 per lane = 390 144 per
#   tile.  Partitioned (C3, two 2048-tap partitions, cached X(b-1)): one more product-and-add
#   per point (8 flops).  GEN: + the generator's polynomial sine, 20 flops per sample.
#   Direct form / matrix-core form: 2 flops per kept tap per real lane.
FFT_TRANSFORM_FLOPS = 714 * 256
FFT_TILE_FLOPS = 2 * FFT_TRANSFORM_FLOPS + 16 * 6 * 256


def kernel_flops_per_sample(kernel_name, ntaps, cplx, nonzero_taps, gen=False):
    w = 2 if cplx else 1
    if "fft" not in kernel_name:
        return 2.0 * nonzero_taps * w            # one FMA per kept tap per real lane
    if "pairs" in kernel_name:                   # real samples filtered as complex pairs by the kept taps
        return kernel_flops_per_sample("fir_fft4096<COMPLEX>", nonzero_taps, True, nonzero_taps, gen) / 2.0
    if "fft16k" in kernel_name:
        # 16384-point tiles (fir_fft16k.hip): per lane and transform, three 16-point butterflies (176 flops), one
        # stage of four 4-point ones (64) and 42 twiddle products (6 each): 844; 1024 lanes, two transforms and
        # sixteen spectrum products per lane per tile of 16384 - halo outputs
        # (halo in whole rows of 1024 samples, two to eight; partitioned at 8192 taps beyond: one more forward transform
        # and product per further partition)
        parts = max(1, -(-ntaps // 8192)) if ntaps > 8193 else 1
        halo = 8192 if parts > 1 else max(2048, 1024 * -(-(ntaps - 1) // 1024))
        per_tile = (16384.0 - halo) * (1 if cplx else 2)   # real signals ride in pairs
        return ((parts + 1) * 844 * 1024 + parts * 16 * 6 * 1024 + (parts - 1) * 16 * 2 * 1024) / per_tile
    halo = next((h for h in (64, 128, 192, 256, 384, 512, 768, 1024) if ntaps <= h + 1), 2048)   # fir_fft_plan
    if ntaps <= 2049:
        per_tile = (4096 - halo) * (1 if cplx else 2)   # real signals ride in pairs
        return FFT_TILE_FLOPS / per_tile + (20.0 if gen else 0.0)
    parts = -(-ntaps // 2048)
    per_tile = 2048 * (1 if cplx else 2)
    if parts == 2:   # cached spectrum: one forward, one inverse, two products + one add per point
        return (2 * FFT_TRANSFORM_FLOPS + 4096 * (2 * 6 + 2)) / per_tile
    return ((parts + 1) * FFT_TRANSFORM_FLOPS + 4096 * (parts * 6 + 2 * (parts - 1))) / per_tile


def bench_chain(args, S, torch, dist, rank, local_rank, world, dev):
    """c4chain / c5: per step, every rank pushes its share through the fused chain."""
    from sdr_amd import sharding
    desc, cplx, _, log2n, channels, scaling = WORKLOADS[args.workload]
    n = 1 << log2n
    stages = [(S.hilbert(127), 63), (S.fir_lowpass(127, 0.1), 0)]
    stream = torch.cuda.current_stream().cuda_stream
    gather = None
    if args.workload == "c4chain":
        # STRONG scaling: this rank's contiguous group of the 64 channels
        first_ch, my_ch = sharding.plan_channels(channels, world)[rank]
        chain = S.Chain(stages, False, channels=max(my_ch, 1), device=local_rank)
        # channel c of the global set: sine at f = (c+1)/160 cycles/sample
        t = torch.arange(n, device=dev, dtype=torch.float64)
        x = torch.empty((max(my_ch, 1), n), dtype=torch.float32, device=dev)
        for c in range(my_ch):
            x[c] = torch.sin(2 * np.pi * ((first_ch + c) % 64 + 1) / 160.0 * t).to(torch.float32)
        del t
        y = torch.empty_like(x)

        def step():
            if my_ch:
                chain.process_tensor(x, y)

        def gather(transport, _plan=[]):
            if not _plan:   # built once from the channel plan: sizes are known, the destination is allocated here
                _plan.append(sharding.GatherPlan([c for _, c in sharding.plan_channels(channels, world)], (n,),
                                                 torch.float32, dev, dist, dst=0))
            return _plan[0].run(y[:my_ch], transport)
        bytes_per_step = n * my_ch * 8
        units = n * my_ch
        total_units = n * channels
    else:
        chain = S.Chain(stages, False, channels=channels, device=local_rank)
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
        total_units = units * world
    is_gen = args.workload == "c5" and not args.unfused
    name = "sdr_chain(253 fused taps) -> fir_fft4096<REAL%s>" % (", GEN" if is_gen else "")
    flops = kernel_flops_per_sample("fir_fft4096", 253, False, 253, gen=is_gen)
    return dict(step=step, units=units, total_units=total_units, step_bytes=bytes_per_step, kernel=name, ntaps=253,
                desc=desc, kind="float32", channels=channels, scaling=scaling, flops_per_sample=flops, gather=gather,
                cpu_spec=("chain", [t for t, _ in stages], False, min(args.cpu_log2n, 26)))


def bench_pipe(args):
    """BASELINE configs[0] end to end, reproducible from one command: `gen freq=100k waveform=sine sample_rate=1M
    mode=real | head -c | hilbert > /dev/null` with the drop-in blocks (blocks/bin: two processes, each starting a
    HIP runtime, packets over a pipe, PCIe both ways), and the reference's own blocks (oracle/_ref, built from
    /root/reference in the build container) beside it in the same run.  `value` here is pipe-, PCIe- and
    process-start-inclusive -- NOT the in-memory value of the other workloads -- and says so."""
    log2n = WORKLOADS["c1pipe"][3]
    B = 2044
    packets = -(-(1 << log2n) // B)
    nbytes = packets * (16 + 4 * B)
    nsamp = packets * B
    BIN, REF = os.path.join(ROOT, "blocks", "bin"), os.path.join(ROOT, "oracle", "_ref")
    gen_args = "freq=100k waveform=sine sample_rate=1M mode=real"
    for exe in ("gen", "hilbert"):
        if not os.path.exists(os.path.join(BIN, exe)):
            raise SystemExit("bench.py --workload c1pipe: blocks/bin/%s is not built (make -C blocks)" % exe)

    def sh(cmd):
        t0 = time.perf_counter()
        subprocess.run(cmd, shell=True, check=True, executable="/bin/bash", stdout=subprocess.DEVNULL)
        return time.perf_counter() - t0

    def pipe(d):
        return "%s/gen %s | head -c %d | %s/hilbert > /dev/null" % (d, gen_args, nbytes, d)
    for _ in range(max(1, args.warmup and 1)):
        sh(pipe(BIN))                                             # image page-in, first GPU touch
    steps = max(1, min(args.steps, 10))
    ts = [sh(pipe(BIN)) for _ in range(steps)]
    # what of that is start-up: the same two processes on ONE packet (HIP runtime start, handles, first launch)
    one = 16 + 4 * B
    t_start = min(sh("%s/gen %s | head -c %d | %s/hilbert > /dev/null" % (BIN, gen_args, one, BIN)) for _ in range(3))
    t = float(np.median(ts))
    out = {
        "metric": "Msamples/s (gen | hilbert through a unix pipe: c1pipe)", "value": nsamp / t / 1e6, "unit": "Msamples/s",
        "value_kind": "END TO END through unix pipes: process start, HIP start-up, packet framing and PCIe both ways included; "
                      "not comparable with the in-memory workloads' value",
        "n_gpus": 1, "steps": steps, "warmup": 1, "ramp_s": 0.0, "untimed_steps": 1, "ms_per_step": t * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOADS["c1pipe"][0], "samples_per_gpu": nsamp, "packets": packets, "taps": 127,
                   "sample_kind": "float32", "command": pipe("blocks/bin")},
        "startup": {"seconds_one_packet": t_start, "share_of_step": t_start / t,
                    "what": "the same pipeline on ONE 2044-sample packet: two process starts, two HIP runtime start-ups, "
                            "filter set-up and the first launch"},
        "roofline": None,
    }
    if not args.no_cpu_baseline and os.path.exists(os.path.join(REF, "hilbert")):
        tr = float(np.median([sh(pipe(REF)) for _ in range(2)]))
        out["cpu_baseline"] = {"value": nsamp / tr / 1e6, "unit": "Msamples/s", "cores": 2, "kind": "reference",
                               "sample": "the whole workload: the reference's gen | head -c | hilbert (oracle/_ref, g++ -std=gnu++14 "
                                         "-O3 -ffast-math -DNDEBUG), two single-threaded processes, same box, same run"}
    else:
        out["cpu_baseline"] = None
    emit_json_line(out)


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


# ---- CPU comparator ----------------------------------------------------------------------

def _ref_fir_seconds(harness, stage_taps, cplx, xpath, d, procs=1):
    """Wall time of `procs` concurrent copies of the reference pipeline over the samples in
    xpath: one ref_harness fir per stage, the stages of a copy run one after the other (through a
    file), so that a copy keeps to one core."""
    cmds = []
    for i, taps in enumerate(stage_taps):
        tp = os.path.join(d, "taps%d.bin" % i)
        np.asarray(taps, np.float32).tofile(tp)
        cmds.append("%s fir %s %s 65536" % (harness, "complex" if cplx else "real", tp))
    ps = []
    t0 = time.perf_counter()
    for j in range(procs):
        src, parts = xpath, []
        for i, c in enumerate(cmds):
            dst = "/dev/null" if i == len(cmds) - 1 else os.path.join(d, "mid%d_%d.bin" % (j, i))
            parts.append("%s < %s > %s" % (c, src, dst))
            src = dst
        ps.append(subprocess.Popen(["/bin/sh", "-c", " && ".join(parts)]))
    for p in ps:
        if p.wait() != 0:
            raise RuntimeError("ref_harness failed")
    return time.perf_counter() - t0


def cpu_baseline(spec):
    """Time the reference's kfr::fir (oracle/_ref/ref_harness) over a bounded sample of the same
    workload: one process on one core, then P independent processes (the reference's only form of
    parallelism) for the aggregate; falls back to the C port of the oracle."""
    _, stage_taps, cplx, log2n = spec
    n = 1 << log2n
    rng = np.random.default_rng(0x5D12)
    x = rng.uniform(-1, 1, n * (2 if cplx else 1)).astype(np.float32)
    harness = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
    ntap_txt = "+".join(str(len(t)) for t in stage_taps)
    what = "first 2^%d %s samples of the workload's stream shape, %s taps" % (log2n, "complex" if cplx else "real", ntap_txt)
    if os.path.exists(harness):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            xp = os.path.join(d, "x.bin")
            x.tofile(xp)
            dt = _ref_fir_seconds(harness, stage_taps, cplx, xp, d)
            out = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "reference",
                   "sample": what + " (reference kfr::fir via oracle/_ref/ref_harness, g++ -O3 -ffast-math -DNDEBUG, "
                             "1 process%s)" % ("; chain = one kfr::fir per stage, run in sequence" if len(stage_taps) > 1 else "")}
            # P independent processes over 1/4 of the sample each (page cache -> /dev/null)
            P = max(1, min(16, len(os.sched_getaffinity(0))))
            n4 = n // 4
            xq = os.path.join(d, "xq.bin")
            x[: n4 * (2 if cplx else 1)].tofile(xq)
            dtp = _ref_fir_seconds(harness, stage_taps, cplx, xq, d, procs=P)
            out["aggregate"] = {"value": P * n4 / dtp / 1e6, "unit": "Msamples/s", "processes": P,
                                "sample": "P independent reference processes, 2^%d samples each" % (log2n - 2)}
        return out
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    xs = x.view(np.complex64) if cplx else x
    t0 = time.perf_counter()
    for taps in stage_taps:
        xs = O.Fir(taps, cplx, acc64=False).process(xs)
    dt = time.perf_counter() - t0
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": what + " (oracle/sdr_oracle.c fp32 port; oracle/_ref not present)"}


# ---- HBM traffic from PMC counters (optional live collection) -----------------------------

def collect_traffic(workload, path):
    """Two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of a short run of this same
    script, averaged per launch of the dominant kernel.  gfx950: FETCH_SIZE counts 64 B per 128-B
    request of a wide streaming read (MI355X_MICROARCH.md, HBM): doubled.  Counter unit: KB."""
    import csv
    import glob
    total = {}
    for counter, scale in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        d = tempfile.mkdtemp(dir="/tmp", prefix="sdr_pmc_")
        cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
               os.path.abspath(__file__), "--workload", workload, "--path", path, "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--no-pmc"]
        subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=60, check=True,
                       env=dict(os.environ, TMPDIR="/tmp"))
        per = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == counter and "fir_" in row["Kernel_Name"]:
                    per.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if not per:
            return None
        name = max(per, key=lambda k: sum(per[k]))
        vals = per[name][1:] or per[name]
        total[counter] = scale * 1024.0 * sum(vals) / len(vals)
    return {"hbm_bytes_per_launch": total["FETCH_SIZE"] + total["WRITE_SIZE"], "read": total["FETCH_SIZE"],
            "written": total["WRITE_SIZE"]}


def traffic_file_version(path):
    """Which measurement a committed traffic file is: the commit that last touched it (when git is there), else its mtime."""
    try:
        r = subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h %cs", "--", path], capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return "commit " + r.stdout.strip()
    except Exception:
        pass
    return "mtime " + time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path)))


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


def spawn_ranks(n, child_argv=None, ndev=None):
    """`python bench.py --gpus N` without a launcher: this process (which never initialises a
    GPU) starts the N ranks as children, one per device, and exits with the worst of their
    codes.  Rank 0 prints the JSON line on the stdout the children inherit.
    child_argv / ndev: the command of one rank and the number of visible GPUs (tests pass
    their own; the defaults are this script with the same arguments and torch's device count)."""
    backend = os.environ.get("SDR_BENCH_BACKEND", "nccl")
    if ndev is None:
        import torch
        ndev = torch.cuda.device_count()   # counting devices does not initialise the GPU
    if child_argv is None:
        child_argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    if backend == "nccl" and ndev < n:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (one rank per GPU; SDR_BENCH_BACKEND=gloo "
                         "rehearses more ranks than GPUs)" % (n, ndev))
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU: libsdr_amd has no CPU path")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(child_argv, env=env))
    worst = 0
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            rc = p.poll()
            if rc is None:
                continue
            alive.remove(p)
            if rc != 0:
                worst = worst or rc
                for q in alive:      # a failed rank would leave the others in a barrier: end exactly our children
                    q.terminate()
    sys.exit(worst if worst >= 0 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--step-events", action="store_true",
                    help="one HIP event per step inside the timed region (rounds 1-2) instead of one at each end")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-log2n", type=int, default=28)
    ap.add_argument("--chunk-log2", type=int, default=29, help="c5: samples per call")
    ap.add_argument("--unfused", action="store_true", help="c5: generator kernel + chain kernel through an HBM buffer")
    ap.add_argument("--path", default="auto", choices=["auto", "direct", "fft", "mfma", "fft4k", "fft16k"],
                    help="experiments: pin the kernel family (the default is the library's own choice)")
    ap.add_argument("--pmc", action="store_true",
                    help="collect roofline.traffic live: two extra rocprofv3 --pmc passes of a short run (N=1 only).  The full "
                         "default run (one GPU, CPU baseline included, not itself under a profiler) does so by itself and falls "
                         "back to the committed profiles/traffic_<workload>.json if the passes fail")
    ap.add_argument("--no-pmc", action="store_true", help="never start the rocprofv3 passes")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.workload == "c1pipe":
        if args.gpus != 1:
            raise SystemExit("--workload c1pipe is a one-GPU pipeline of processes")
        bench_pipe(args)         # the blocks are processes of their own: this one never touches the GPU
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)   # does not return
    keep_stdout_for_the_json_line()

    import torch
    import torch.distributed as dist
    import sdr_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libsdr_amd has no CPU path")
    # one rank per GPU; SDR_BENCH_BACKEND=gloo rehearses the multi-rank path on a box with
    # fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("SDR_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d has no GPU of its own (%d visible)" % (rank, torch.cuda.device_count()))
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

    desc, cplx, mk_taps, log2n, channels, scaling = WORKLOADS[args.workload]
    if mk_taps is None:
        run_and_report(args, torch, dist, rank, world, dev, bench_chain(args, S, torch, dist, rank, local_rank, world, dev))
        return
    from sdr_amd import sharding
    taps = mk_taps(S)
    n = 1 << log2n
    w = 2 if cplx else 1
    if scaling == "strong":
        # the workload's channels are dealt over the ranks (BASELINE configs[3]: 64 channels over 1/2/4/8 GPUs)
        first_ch, my_ch = sharding.plan_channels(channels, world)[rank]
        total_units = n * channels
        first = first_ch * n
    else:
        my_ch = channels
        total_units = n * channels * world
        # weak scaling: the global stream is world * n samples; this rank owns one contiguous chunk
        first, count = sharding.plan_chunks(world * n * channels, world, n * channels)[rank]
        assert count == n * channels
    nch = max(my_ch, 1)
    st = S.FirState(taps, cplx, channels=nch, device=local_rank, path=args.path)
    x = torch.empty((nch * n, w), dtype=torch.float32, device=dev)
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
    xin = x if cplx else x.view(nch, n) if nch > 1 else x.view(-1)
    yout = y if cplx else y.view(nch, n) if nch > 1 else y.view(-1)
    if cplx and nch > 1:
        xin, yout = x.view(nch, n, 2), y.view(nch, n, 2)

    def step():
        if my_ch:
            st.process_tensor(xin, yout)  # launches on torch's current stream

    def gather(transport, _plan=[]):
        if not _plan:
            _plan.append(sharding.GatherPlan([c * n for _, c in sharding.plan_channels(channels, world)], (w,),
                                             torch.float32, dev, dist, dst=0))
        return _plan[0].run(y[: my_ch * n], transport)
    nonzero = int(np.count_nonzero(taps)) if st.tap_stride == 2 else len(taps)
    run_and_report(args, torch, dist, rank, world, dev, dict(
        step=step, units=n * my_ch, total_units=total_units, step_bytes=n * my_ch * 4 * w * 2, kernel=st.kernel_name,
        ntaps=int(len(taps)), desc=desc, kind="complex64" if cplx else "float32", channels=channels, scaling=scaling,
        flops_per_sample=kernel_flops_per_sample(st.kernel_name, len(taps), cplx, nonzero),
        gather=gather if scaling == "strong" else None, cpu_spec=("fir", [taps], cplx, min(args.cpu_log2n, log2n))))


def run_and_report(args, torch, dist, rank, world, dev, wl):
    """W warm-up steps, K timed steps between barrier + synchronize, max over ranks, one JSON line."""
    step = wl["step"]
    # Set-up, untimed and not counted as warm-up steps: keep the path running for ~0.2 s so that the GPU has left
    # its idle clocks before the W warm-up steps begin (the first ~20 launches after idle run 3-5 % slower; a
    # driver that asks for `--steps 20 --warmup 5` would otherwise time the ramp, not the kernel).
    t_ramp = time.perf_counter()
    ramp_steps = 0
    while time.perf_counter() - t_ramp < RAMP_S:
        step()
        torch.cuda.synchronize()
        ramp_steps += 1
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if USE_DIST:
        dist.barrier()
        torch.cuda.synchronize()
    # HIP events on the launch stream around the timed region.  (One event per step inside it -- rounds 1 and 2, --step-events -- puts
    # a barrier packet behind every launch: the steps then run 2-3 us apart instead of back to back.  The spread of single launches is
    # taken in a short pass of its own behind the timed region.)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1 if args.step_events else 2)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        if args.step_events:
            ev[i + 1].record()
    if not args.step_events:
        ev[1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    on_gpu = USE_DIST and dist.get_backend() == "nccl"
    if USE_DIST:
        dist.barrier()
        torch.cuda.synchronize()
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # device time per step: HIP events on the launch stream (torch's current stream is
    # the stream every kernel of the step is launched on)
    if args.step_events:
        kern_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]
        kern_ms_avg = float(np.mean(kern_ms))
    else:
        kern_ms_avg = ev[0].elapsed_time(ev[1]) / args.steps
        m = min(args.steps, 20)
        ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(m + 1)]
        ev2[0].record()
        for i in range(m):
            step()
            ev2[i + 1].record()
        torch.cuda.synchronize()
        kern_ms = [ev2[i].elapsed_time(ev2[i + 1]) for i in range(m)]

    # strong-scaling workloads: the same steps followed by the gather of every rank's outputs on
    # rank 0 (result collection only; SURVEY 8e asks for throughput with and without it)
    with_gather = {}
    if USE_DIST and wl.get("gather") is not None:
        gsteps = max(1, min(args.steps, 5)) if on_gpu else 1   # the gloo rehearsal moves GBs through host sockets
        for transport in ("p2p", "gather"):
            if rank == 0:
                print("bench.py: timing %d step(s) + result collection on rank 0 (%s)" % (gsteps, transport), file=sys.stderr, flush=True)
            try:
                wl["gather"](transport)         # communicator warm, destination buffer allocated (once, by the plan)
            except ValueError:                  # the rooted gather needs equal slices
                continue
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(gsteps):
                step()
                wl["gather"](transport)
            torch.cuda.synchronize()
            dtg = time.perf_counter() - t1
            dist.barrier()
            t = torch.tensor([dtg], dtype=torch.float64, device=dev if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            with_gather[transport] = (float(t.item()) / gsteps, gsteps)
    # every rank's own device time per step: imbalance between GPUs shows here
    per_rank_ms = [kern_ms_avg]
    if USE_DIST:
        t = torch.zeros(world, dtype=torch.float64, device=dev if on_gpu else "cpu")
        t[rank] = kern_ms_avg
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [float(v) for v in t.tolist()]

    if rank == 0:
        units_all = wl["total_units"]
        value = units_all * args.steps / elapsed / 1e6
        my_rate = wl["units"] / (kern_ms_avg * 1e-3)                       # samples/s of rank 0's kernel
        achieved_gbs = wl["step_bytes"] / (kern_ms_avg * 1e-3) / 1e9
        achieved_tflops = wl["flops_per_sample"] * my_rate / 1e12
        f_hbm, f_valu = achieved_gbs / HBM_PEAK_GBS, achieved_tflops / VALU_PEAK_TFLOPS
        hbm_bound = f_hbm >= f_valu
        out = {
            "metric": "Msamples/s (FIR hot path: %s)" % args.workload,
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ramp_s": RAMP_S, "untimed_steps": ramp_steps + args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": wl["scaling"], "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["desc"], "samples_per_gpu": wl["units"], "channels": wl["channels"],
                       "taps": wl["ntaps"], "sample_kind": wl["kind"], "kernel": wl["kernel"],
                       "sharding": ("64 channels dealt in contiguous groups over the GPUs, no collective"
                                    if wl["scaling"] == "strong" else
                                    "contiguous chunks with regenerated halo, no collective")},
            "roofline": {"bound": "hbm" if hbm_bound else "valu",
                         "achieved": achieved_gbs if hbm_bound else achieved_tflops,
                         "peak": HBM_PEAK_GBS if hbm_bound else VALU_PEAK_TFLOPS,
                         "unit": "GB/s" if hbm_bound else "TFLOP/s",
                         "frac": f_hbm if hbm_bound else f_valu, "traffic": None,
                         "achieved_hbm": f_hbm, "achieved_valu": f_valu,
                         "hbm_gbs": achieved_gbs, "valu_tflops": achieved_tflops,
                         "kernel_flops_per_sample": wl["flops_per_sample"],
                         "kernel_ms": kern_ms_avg, "kernel_ms_min": float(np.min(kern_ms)),
                         "kernel_ms_source": ("one HIP event per step inside the timed region" if args.step_events else
                                              "HIP events at both ends of the timed region / steps; kernel_ms_min from %d single "
                                              "launches timed behind it" % len(kern_ms)),
                         "kernel_ms_per_rank": per_rank_ms,
                         "algorithmic_bytes_per_launch": wl["step_bytes"]},
        }
        if with_gather:
            best = min(with_gather, key=lambda k: with_gather[k][0])
            out["value_with_gather"] = units_all / with_gather[best][0] / 1e6
            out["gather"] = {"transport": best, "steps": with_gather[best][1],
                             "ms_per_step_with_gather": {k: v[0] * 1e3 for k, v in with_gather.items()},
                             "what": "every rank's output slice collected in one preallocated buffer on rank 0 after each step "
                                     "(sharding.GatherPlan: sizes from the plan, no allocation per step); p2p = every rank sends "
                                     "straight into its place (batch_isend_irecv), gather = the rooted collective"}
        tj = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        profiled = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
        want_live = world == 1 and not args.no_pmc and (args.pmc or (not args.no_cpu_baseline and not profiled and
                                                                     args.workload != "c1pipe" and shutil.which("rocprofv3")))
        live = None
        if want_live:
            try:
                live = collect_traffic(args.workload, args.path)
            except Exception as e:  # a failed or timed-out pass must not cost the bench line
                sys.stderr.write("bench.py: live traffic collection failed (%s); using the committed file\n" % (e,))
        if live:
            out["roofline"]["traffic"] = live["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = ("live: rocprofv3 --pmc FETCH_SIZE (x2 on gfx950) and WRITE_SIZE, two separate "
                                                 "passes of a short run of this command after the timed region")
        elif os.path.exists(tj) and args.path == "auto" and world == 1:
            with open(tj) as f:
                rec = json.load(f)
            out["roofline"]["traffic"] = rec.get("hbm_bytes_per_launch")
            out["roofline"]["traffic_source"] = "profiles/traffic_%s.json (%s; NOT measured in this run, pass --pmc for that): %s" % (
                args.workload, traffic_file_version(tj), rec.get("source", "rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE passes of this command"))
        if not args.no_cpu_baseline and wl.get("cpu_spec") is not None:
            # rank 0's host cores, after the timed region (the other ranks wait at the end of the run)
            out["cpu_baseline"] = cpu_baseline(wl["cpu_spec"])
        else:
            out["cpu_baseline"] = None
        emit_json_line(out)
    if USE_DIST:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
