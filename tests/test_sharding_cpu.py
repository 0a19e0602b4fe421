"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of sdr_amd.sharding.  The
GPU FirState cannot run here, so the ORACLE stands in as the per-rank filter
(checker role only): what is under test is the sharding plan, the halo priming and
the gather -- chunked results must equal the unchunked stream bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from sdr_amd import sharding


def test_plan_chunks_cover_and_alignment():
    for n, w, a in [(0, 2, 1), (10, 3, 1), (1 << 20, 8, 2044), (5, 8, 1), (4099, 2, 4)]:
        plan = sharding.plan_chunks(n, w, a)
        assert len(plan) == w and plan[0][0] == 0
        pos = 0
        for s, c in plan:
            assert s == pos and s % a == 0 or c == 0
            pos += c
        assert pos == n
    assert sharding.plan_channels(64, 8) == [(8 * i, 8) for i in range(8)]
    assert sharding.halo_range(0, 127) == (0, 0)
    assert sharding.halo_range(50, 127) == (0, 50)
    assert sharding.halo_range(5000, 127) == (4874, 126)


def test_plan_generated_chain():
    plan = sharding.plan_generated_chain(2101248, 8, 2044, 253)    # 2^32 samples of config 5
    assert [p[1] for p in plan] == [262656] * 8 and plan[0] == (0, 262656, 0)
    assert all(p[2] == 1 for p in plan[1:])                        # one 2044-sample packet covers 252 halo samples
    assert sharding.plan_generated_chain(40, 2, 300, 1000)[1] == (20, 20, 4)
    with pytest.raises(ValueError):
        sharding.plan_generated_chain(4, 4, 256, 253)


def _worker(rank, world, port, n, ntaps, cplx, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(99)
    x = rng.uniform(-1, 1, n * (2 if cplx else 1)).astype(np.float32)
    x = x.view(np.complex64) if cplx else x
    taps = O.hilbert_taps(ntaps)
    start, y = sharding.filter_chunk(lambda: O.Fir(taps, cplx), lambda s, c: x[s:s + c], n, ntaps, rank, world, align=4)
    yt = torch.from_numpy(np.ascontiguousarray(y).view(np.float32).reshape(len(y), -1))
    parts = sharding.gather_chunks(yt, dist, dst=0)
    if rank == 0:
        got = torch.cat(parts).numpy().reshape(-1)
        ref = O.fir(taps, x).view(np.float32)
        q.put(bool(np.array_equal(got, ref)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,cplx", [(2, False), (2, True), (3, False)])
def test_chunked_stream_equals_whole_stream_gloo(world, cplx):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 7 + int(cplx)
    procs = [ctx.Process(target=_worker, args=(r, world, port, 30001, 127, cplx, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _gather_worker(rank, world, port, counts, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tail = (5, 2)
    plan = sharding.GatherPlan(counts, tail, torch.float32, "cpu", dist, dst=0)
    out_ptr = plan.out.data_ptr() if rank == 0 else None
    ok = True
    transports = ["p2p", "gather"] if len(set(counts)) == 1 else ["p2p"]
    for step in range(3):
        for tr in transports:
            # rank r's slice of step s: values that name (rank, step, position)
            y = (torch.arange(counts[rank] * 10, dtype=torch.float32).reshape((counts[rank],) + tail)
                 + 1000.0 * rank + 100000.0 * step)
            got = plan.run(y, tr)
            if rank == 0:
                want = torch.cat([torch.arange(c * 10, dtype=torch.float32).reshape((c,) + tail) + 1000.0 * r + 100000.0 * step
                                  for r, c in enumerate(counts)])
                ok = ok and got.data_ptr() == out_ptr and bool(torch.equal(got, want))   # same buffer every step
            else:
                ok = ok and got is None
    if len(set(counts)) != 1:
        try:
            plan.run(torch.zeros((counts[rank],) + tail), "gather")
            ok = False
        except ValueError:
            pass
    try:
        plan.run(torch.zeros((counts[rank] + 1,) + tail), "p2p")
        ok = False
    except ValueError:
        pass
    q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[8, 8], [22, 21, 21], [3, 0, 5]])
def test_gather_plan_allocates_once_and_needs_no_size_exchange(counts):
    """GatherPlan (what bench.py's value_with_gather times): sizes from the plan, one destination buffer reused
    every step, both transports (direct point-to-point pushes into place; the rooted collective), ragged and
    empty slices, shape mistakes refused."""
    world = len(counts)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + sum(counts)
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, counts, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(q.get(timeout=5) for _ in range(world))
