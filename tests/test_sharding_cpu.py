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
