"""Data-parallel gradient exchange: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-device (tf_aerial_images.py:387-393); data parallelism is new here. The global minibatch is
sharded over ranks (same shuffle on every rank, rank r takes its slice), weights and Momentum slots are replicated,
and the per-rank gradients -- already scaled by 1/(GLOBAL pixel count) in the head kernel -- are SUM-all-reduced,
which equals the reference's reduce_mean over the whole batch (tf_aerial_images.py:108).

Gradients live in one flat float32 buffer in variable-creation order; backward produces them from the END of the
buffer towards the START (head, decoder stages, encoder levels, colour adjust), so each finished block is a
contiguous tail slice: it is all-reduced on a side stream while the remaining backward kernels run.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): buckets are whole blocks (MBs, not KBs) so each ring step
moves large messages; the dead conv_dilut_{L-1} variables sit after `n_live` and are never sent.
"""
import torch
import torch.distributed as dist


def shard_indices(indices, offset, global_batch, rank, world_size):
    """Indices of the patches rank `rank` processes for the step starting at `offset` of the shuffled index list
    (tf_aerial_images.py:232-233 batch slicing, split contiguously over ranks)."""
    assert global_batch % world_size == 0, "global batch must divide evenly over ranks"
    per = global_batch // world_size
    lo = offset + rank * per
    return indices[lo:lo + per]


class GradBucketer:
    """All-reduce a flat gradient buffer in tail-first buckets, overlapped with the producer.

    flat: 1-D float32 tensor (CPU with gloo, GPU with nccl); n_live: prefix length that takes part.
    Call ready(lo) when every gradient at positions >= lo has been produced (monotonically decreasing lo), then
    finish() before the optimizer step."""

    def __init__(self, flat, n_live, group=None, min_bucket_elems=1 << 20, overlap=None):
        self.flat, self.n_live, self.group = flat, n_live, group
        self.min_bucket = min_bucket_elems
        # overlap=False (or RSU_DP_OVERLAP=0): one all-reduce of the whole buffer in finish(). The conv kernels are persistent,
        # one workgroup per CU with the whole register file: RCCL's channel blocks and they cannot share a CU, so overlapped
        # buckets trade hidden communication against stretched conv launches -- measure both on the target node.
        if overlap is None:
            import os
            overlap = os.environ.get("RSU_DP_OVERLAP", "1") != "0"
        self.overlap = overlap
        self.extra_streams = []   # producer streams besides the current one (the network's weight-gradient stream)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = flat.is_cuda
        self.comm_stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.reset()

    def reset(self):
        self.hi = self.n_live
        self.pending = []

    def _launch(self, lo, hi):
        if self.world == 1 or hi <= lo:
            return
        view = self.flat[lo:hi]
        if self.cuda:
            for prod in [torch.cuda.current_stream(self.flat.device)] + [s for s in self.extra_streams if s is not None]:
                ev = torch.cuda.Event()
                ev.record(prod)
                self.comm_stream.wait_event(ev)  # only the communication stream waits: the producers keep running
            with torch.cuda.stream(self.comm_stream):
                self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.pending.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, lo):
        if not self.overlap:
            return
        lo = max(0, min(lo, self.hi))
        if self.hi - lo >= self.min_bucket:
            self._launch(lo, self.hi)
            self.hi = lo

    def finish(self):
        self._launch(0, self.hi)
        self.hi = 0
        for w in self.pending:
            w.wait()
        if self.cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.comm_stream)
        self.pending = []


# (overlapped buckets?, CU budget of the conv launches, smallest bucket in elements: 4 MB / 32 MB of fp32)
EXCHANGE_CANDIDATES = ((True, 256, 1 << 20), (True, 256, 1 << 23), (True, 240, 1 << 23), (True, 224, 1 << 23), (True, 208, 1 << 23),
                       (False, 256, 1 << 20))


def tune_overlap(bucketer, step, trials=3, candidates=EXCHANGE_CANDIDATES, set_cu_budget=None):
    """Pick the gradient-exchange schedule by measurement: `step()` (forward + backward with bucketer.reset/ready/finish) runs
    `trials` times under every candidate -- tail-first overlapped buckets with the conv launches planned for 256, 240, 224 or 208
    CUs (RCCL's channel workgroups and a persistent conv workgroup cannot share a CU: with all 256 taken, a conv launch that
    starts while an all-reduce is resident waits for CUs and roughly doubles), or one all-reduce after backward -- and the
    candidate with the smallest max-over-ranks time stays selected (every rank sees the same reduced timings, so all agree).
    Small buckets start the exchange earlier, large ones mean fewer collective launches and bigger per-link messages on the
    point-to-point xGMI links: both sizes are tried.
    Which one wins depends on the node at hand, which no single-GPU run can tell.
    set_cu_budget: callable(int) (the library's rsu_set_cu_budget); None = budgets other than 256 are skipped.
    Returns {"overlap": bool, "cu_budget": int, "min_bucket": int, "ms": {(overlap, budget, min_bucket): ms_per_step}}."""
    import time
    if bucketer is None or bucketer.world == 1:
        return {"overlap": bool(bucketer.overlap) if bucketer is not None else False, "cu_budget": 256,
                "min_bucket": bucketer.min_bucket if bucketer is not None else 0, "ms": {}}
    dev = bucketer.flat.device
    timings = {}
    for mode, budget, min_bucket in candidates:
        if budget != 256 and set_cu_budget is None:
            continue
        bucketer.overlap = mode
        bucketer.min_bucket = min_bucket
        if set_cu_budget is not None:
            set_cu_budget(budget)
        step()  # settle (stream creation, RCCL channel setup for this message pattern)
        if bucketer.cuda:
            torch.cuda.synchronize(dev)
        dist.barrier(group=bucketer.group)
        t0 = time.perf_counter()
        for _ in range(trials):
            step()
        if bucketer.cuda:
            torch.cuda.synchronize(dev)
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=bucketer.group)
        timings[(mode, budget, min_bucket)] = float(t.item()) / trials * 1e3
    best = min(timings, key=lambda k: (timings[k], not k[0], -k[1]))
    bucketer.overlap = best[0]
    bucketer.min_bucket = best[2]
    if set_cu_budget is not None:
        set_cu_budget(best[1])
    return {"overlap": best[0], "cu_budget": best[1], "min_bucket": best[2], "ms": timings}
