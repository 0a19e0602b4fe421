"""Multi-GPU sharding of the FIR path: one process per GPU, no data-path collective.

The path shards two ways:
  * independent channels: channel c is one stream id with its own filter history;
    ranks take contiguous groups of channels, nothing is exchanged;
  * one long stream: rank r owns outputs [start_r, start_r + count_r) and needs the
    ntaps-1 inputs before start_r (its left halo).  The halo is taken from the
    SOURCE of the stream (host ring, file, or a regenerable synthetic input), not
    from the neighbouring GPU, so chunks are independent; the only collective is
    the optional gather of results.
The group delay skip of a block (hilbert's (ntaps-1)/2 dropped outputs) applies to
the global stream start only, i.e. on the rank that owns sample 0.
"""


def plan_chunks(n_total, world, align=1):
    """Contiguous, ordered, disjoint cover of [0, n_total): list of (start, count),
    one per rank; every start is a multiple of `align` (the last chunk takes the
    remainder)."""
    if world < 1 or n_total < 0 or align < 1:
        raise ValueError("bad sharding request")
    units = -(-n_total // align)
    base, extra = divmod(units, world)
    out, start = [], 0
    for r in range(world):
        cnt = (base + (1 if r < extra else 0)) * align
        cnt = max(0, min(cnt, n_total - start))
        out.append((start, cnt))
        start += cnt
    return out


def plan_channels(channels, world):
    """Contiguous groups of channels, one per rank: list of (first, count)."""
    return plan_chunks(channels, world, 1)


def halo_range(start, ntaps):
    """Global sample range [lo, start) a chunk starting at `start` must be primed
    with (empty for the chunk that owns sample 0: its history is zeros)."""
    lo = max(0, start - (ntaps - 1))
    return lo, start - lo


def filter_chunk(make_state, source, n_total, ntaps, rank, world, align=1):
    """Filter this rank's chunk of a stream of n_total samples.

    make_state() -> object with .process(x) (streaming, history carried): the GPU
    FirState in production.  source(start, count) -> the input samples of that
    global range.  Returns (start, y) with y = the stream's outputs [start, start+count).
    """
    start, count = plan_chunks(n_total, world, align)[rank]
    st = make_state()
    lo, hl = halo_range(start, ntaps)
    if hl:
        st.process(source(lo, hl))      # prime the history; outputs discarded
    y = st.process(source(start, count)) if count else source(start, 0)
    return start, y


def plan_generated_chain(n_blocks, world, block_size, chain_ntaps, min_seek=None):
    """BASELINE config 5: a generated stream of n_blocks packets of block_size samples
    through a fused chain of chain_ntaps combined taps, cut into whole packets.  Per
    rank: (first_block, blocks, halo_blocks) -- the rank positions its generator at
    packet first_block - halo_blocks (Gen.advance), seeks its chain to that sample
    (Chain.seek; rank 0 starts the stream instead), discards the outputs of the
    halo_blocks priming packets and keeps the next blocks * block_size outputs.  The
    outputs of all ranks, in rank order, are the stream's outputs (rank 0's are
    shorter by the chain's total group-delay skip).  min_seek: Chain.min_seek of the chain
    (sdr_chain_min_seek), the smallest position a rank may seek to; without it a safe
    over-estimate (4 * chain_ntaps) is used."""
    halo = -(-(chain_ntaps - 1) // block_size)
    floor = 4 * chain_ntaps if min_seek is None else int(min_seek)
    plan = []
    for first, cnt in plan_chunks(n_blocks, world, 1):
        h = 0 if first == 0 else halo
        if first and (first - h) * block_size < floor:
            raise ValueError("chunks this small would start inside the start-of-stream region")
        plan.append((first, cnt, h))
    return plan


class GatherPlan:
    """Result collection for a sharded call: every rank's output slice into ONE buffer on rank `dst`, laid out
    in rank order.  Built once from the sharding plan -- every rank knows every slice length, so nothing is
    exchanged or synchronised to learn sizes -- and reused for every step: the destination buffer is allocated
    here, once (empty, not zero-filled: every element is written by the gather), and a step allocates nothing.

    counts: elements of the leading dimension each rank contributes (the plan: plan_chunks / plan_channels);
    tail: shape of one element (e.g. (n,) for a channel of n samples, (n, 2) complex); dtype / device: of the
    slices on this rank.  On ranks other than dst no buffer is held.

    Two transports, same result:
      "p2p"    -- every rank sends its slice straight into its place in rank dst's buffer (one batch of
                  point-to-point operations; RCCL moves each over the sender's own xGMI link, so the 7 links
                  into dst work at once).  Ragged slices need no padding.
      "gather" -- the rooted collective (torch.distributed.gather; ring / tree inside RCCL).  Equal slices only.
    Compute never waits on this: it is result collection."""

    def __init__(self, counts, tail, dtype, device, dist, dst=0):
        import torch
        self.dist, self.dst = dist, dst
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        if len(counts) != self.world:
            raise ValueError("one count per rank")
        self.counts = [int(c) for c in counts]
        self.tail = tuple(tail)
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        self.equal = len(set(self.counts)) == 1
        self.host = dist.get_backend() != "nccl"   # the CPU rehearsal backend (gloo) moves host tensors
        dev = "cpu" if self.host else device
        self.out = None
        self.views = None
        if self.rank == dst:
            self.out = torch.empty((self.offsets[-1],) + self.tail, dtype=dtype, device=dev)
            self.views = [self.out[self.offsets[r]:self.offsets[r + 1]] for r in range(self.world)]
        # the rehearsal backend needs a host copy of this rank's slice: one staging buffer, allocated once
        self.stage = torch.empty((self.counts[self.rank],) + self.tail, dtype=dtype, device="cpu") if self.host else None

    def _mine(self, y):
        if tuple(y.shape) != (self.counts[self.rank],) + self.tail:
            raise ValueError("slice shape %s does not match the plan %s" % (tuple(y.shape), (self.counts[self.rank],) + self.tail))
        if self.host and y.is_cuda:
            self.stage.copy_(y)
            return self.stage
        return y

    def run(self, y, transport="p2p"):
        """Collect this step's slices; returns the whole buffer on dst (rank order), None elsewhere."""
        y = self._mine(y)
        dist = self.dist
        if transport == "gather":
            if not self.equal:
                raise ValueError("the rooted gather needs equal slices; use transport='p2p'")
            dist.gather(y, self.views if self.rank == self.dst else None, dst=self.dst)
        elif transport == "p2p":
            ops = []
            if self.rank == self.dst:
                self.views[self.dst].copy_(y)
                ops = [dist.P2POp(dist.irecv, self.views[r], r) for r in range(self.world)
                       if r != self.dst and self.counts[r]]
            elif self.counts[self.rank]:
                ops = [dist.P2POp(dist.isend, y, self.dst)]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
        else:
            raise ValueError("transport must be 'p2p' or 'gather'")
        return self.out


def gather_chunks(y, dist, dst=0):
    """One-off convenience form for callers that do NOT know the plan (sizes are exchanged, buffers are
    allocated per call): gather every rank's output slice on rank `dst` (torch.distributed, any backend;
    slices may differ in length).  Returns the list of slices on dst, None elsewhere.  Anything that repeats
    -- bench.py, a serving loop -- uses GatherPlan, which allocates once and exchanges nothing."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if y.is_cuda and dist.get_backend() != "nccl":
        y = y.cpu()   # the CPU rehearsal backend (gloo) gathers host tensors
    n = torch.tensor([y.shape[0]], dtype=torch.int64, device=y.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
    pad[: y.shape[0]] = y
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]
