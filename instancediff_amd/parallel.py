"""Data parallelism for the InstanceDiff hot path on one MI355X node: one process per GPU,
`torch.distributed` (backend "nccl" == RCCL on ROCm, over xGMI).

Replaces the reference's 10 `DistributedDataParallel` wrappers with `find_unused_parameters=True`
(models/drift_noise_model.py:115-146) and its hard-coded `world_size=2` (trainUM.py:66) by
  * ONE flat fp32 gradient buffer (both nets, ~50 M parameters = ~200 MB) whose slices ARE the
    parameters' .grad tensors, so "packing" is free, and
  * ONE all-reduce(SUM) per step followed by a 1/world scale (fused into the optimizer's grad scale).
xGMI is point-to-point (7 links/GPU): a single large message lets RCCL drive all links; 200 MB is far
above the latency-bound regime, and the exchange is << the ~100 ms backward.

Sampling needs no collective: images are independent, ranks take `indices[rank::world]`
(data/data_sampler.py:59 semantics) -- `shard_indices`.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """env:// rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT (torchrun); returns
    (rank, world, local_rank).  No-op for WORLD_SIZE<=1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_indices(n, rank, world):
    """indices[rank::world] -- the reference sampler's partition (data/data_sampler.py:59)."""
    return list(range(n))[rank::world]


class GradSync:
    """Data-parallel exchange for optimizers that already own flat gradient buffers (train_ops.FusedAdam): one all-reduce(SUM)
    per flat buffer per step over RCCL; the 1/world average is folded into the fused Adam kernel.

      * overlap: `start(flats)` enqueues the all-reduces asynchronously (RCCL runs them on its own stream, ordered after the
        work already on the current stream) and returns; `finish()` makes the current stream wait for them.  The train step
        starts the drift net's exchange right after its backward, so it travels over xGMI while the noise net's backward
        computes (the reference gets the same effect from DDP's bucket hooks, models/drift_noise_model.py:145-146).
      * wire="bf16": the flat fp32 gradients are packed to bf16 (idiff_f32_to_bf16, round to nearest even), summed in bf16 on
        the wire -- half the bytes per link -- and widened back into the fp32 master buffer (BASELINE config c3).  Default fp32.
      * single_rank_collectives: run the collectives at world size 1 too (a test hook: the RCCL path -- communicator, kernels,
        stream ordering -- is exercised on a one-GPU box).
      * timing=True (bench.py): every step records three marks -- `start` (first start() of the step: the point behind which RCCL's
        stream may begin), `wait0` (finish() entered: the backward of both nets and the gradient gather are complete on the current
        stream) and `wait1` (the current stream has waited for every exchange, bf16 widening included) -- as HIP events on the current
        stream (wall-clock stamps on the CPU / gloo path).  timings() turns them into per-step `span_ms` = start -> wait1 (the exchange
        with whatever overlapped it) and `exposed_ms` = wait0 -> wait1 (what the step actually waited for: the non-overlapped part).
    """

    def __init__(self, group=None, wire=None, single_rank_collectives=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.wire = (wire or os.environ.get("IDIFF_GRAD_WIRE", "fp32")).lower()
        if self.wire not in ("fp32", "bf16"):
            raise ValueError(f"gradient wire format must be fp32 or bf16, got {self.wire!r}")
        self.active = dist.is_initialized() and (self.world > 1 or single_rank_collectives)
        self._pending = []
        self._wirebufs = {}
        self.timing = False
        self._marks = []   # per finished step: (start, wait0, wait1)
        self._mark0 = None

    def _mark(self):
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def timings(self, reset=True):
        """[{span_ms, exposed_ms}] of the steps finished since the last reset (synchronises the device)"""
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
        out = []
        for a, b, c in self._marks:
            if isinstance(a, float):
                out.append({"span_ms": (c - a) * 1e3, "exposed_ms": (c - b) * 1e3})
            else:
                out.append({"span_ms": a.elapsed_time(c), "exposed_ms": b.elapsed_time(c)})
        if reset:
            self._marks = []
        return out

    @torch.no_grad()
    def broadcast_parameters(self, params, src=0):
        if not self.active:
            return
        for p in params:
            dist.broadcast(p.data, src=src, group=self.group)

    @torch.no_grad()
    def start(self, flats):
        """enqueue all-reduce(SUM) of every flat gradient buffer; returns immediately"""
        if not self.active:
            return
        if any(f is g for _, g, _ in self._pending for f in flats):
            # a previous step left between start() and finish() (an exception the caller caught): its stale handles must not be
            # waited on -- and, on the bf16 wire, widened over the fresh gradients -- by this step's finish()
            self.drain()
        if self.timing and not self._pending:
            self._mark0 = self._mark()
        for f in flats:
            if self.wire == "bf16" and f.is_cuda:
                from . import ops
                wb = self._wirebufs.get(f.data_ptr())
                if wb is None or wb.numel() != f.numel():
                    wb = self._wirebufs[f.data_ptr()] = torch.empty(f.numel(), device=f.device, dtype=torch.bfloat16)
                ops.f32_to_bf16(f, out=wb)
                self._pending.append((dist.all_reduce(wb, op=dist.ReduceOp.SUM, group=self.group, async_op=True), f, wb))
            else:
                self._pending.append((dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True), f, None))

    @torch.no_grad()
    def finish(self):
        """wait for every started exchange; returns the scale (1/world) still to be applied by the optimizer"""
        if not self.active:
            return 1.0
        pending, self._pending = self._pending, []  # cleared whatever happens below
        w0 = self._mark() if (self.timing and pending) else None
        for work, f, wb in pending:
            work.wait()  # stream-ordered for RCCL (the current stream waits), blocking for gloo
            if wb is not None:
                from . import ops
                ops.bf16_to_f32(wb, out=f)
        if w0 is not None:
            self._marks.append((self._mark0 if self._mark0 is not None else w0, w0, self._mark()))
            self._mark0 = None
        return 1.0 / self.world

    @torch.no_grad()
    def drain(self):
        """wait for exchanges that were started and never finished, and discard their results (collectives are matched across
        ranks by order, so they are completed, not cancelled)"""
        pending, self._pending = self._pending, []
        self._mark0 = None
        for work, _, _ in pending:
            work.wait()

    @torch.no_grad()
    def all_reduce_flat(self, flats):
        """SUM every flat gradient buffer over the ranks; returns the scale (1/world) still to be applied."""
        self.start(flats)
        return self.finish()


class FlatGradAllReduce:
    """Owns one flat gradient buffer; every parameter's .grad is a view into it."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        assert self.params, "no trainable parameters"
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            o += n
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def zero(self):
        self.flat.zero_()

    def rebind(self):
        """re-attach .grad views (an optimizer's zero_grad(set_to_none=True) would drop them)."""
        o = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat[o:o + n].data_ptr():
                p.grad = self.flat[o:o + n].view_as(p)
            o += n

    @torch.no_grad()
    def broadcast_parameters(self, src=0):
        """rank-0 weights -> all ranks (what DDP does at construction; reference C2)."""
        if self.world <= 1:
            return
        for p in self.params:
            dist.broadcast(p.data, src=src, group=self.group)

    @torch.no_grad()
    def all_reduce(self, average=True):
        """SUM over ranks; returns the factor the caller must still apply (1/world when averaging is deferred to
        the optimizer's fused grad scale) -- or applies it here when average=True."""
        if self.world <= 1:
            return 1.0
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if average:
            self.flat.mul_(1.0 / self.world)
            return 1.0
        return 1.0 / self.world
