"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-process; this is the sharding of ITS step over ranks such that the
result equals the reference's step on the GLOBAL batch:
  * every rank draws the same index stream (same seed) and takes rows [rank*B, (rank+1)*B) of each global batch;
  * recon / KL are normalised by the global batch size inside the kernels (inv_B), CE / entropy are sums, so a SUM
    all-reduce of the parameter gradients is the single-batch gradient;
  * MMD couples all rows: z (or y1) and the condition rows are all-gathered, each rank evaluates the kernel
    blocks (own rows x all rows).  K is symmetric, so that row block already holds the complete gradient of the
    rank's own rows: there is NO backward collective for the MMD;
  * the four loss scalars are SUM all-reduced for logging;
  * use_bn: the per-column (mean, M2) partials [2,H] of every BatchNorm layer are all-gathered in the forward and
    combined by the kernel (Chan), the column sums [2,H] of its backward are SUM all-reduced -- SyncBN, so that the
    statistics are those of the global batch.
The gradient exchange runs per bucket of the engine's flat gradient arena, in backward-readiness order, on a side
stream, so the decoder-output bucket (the first 78 MB at 19k genes x 1024) overlaps the rest of the backward.

Two forms of the exchange:
  * `shard_optimizer=False`: SUM all-reduce of every bucket; every rank runs Adam on every parameter.
  * `shard_optimizer=True`: every bucket is REDUCE-SCATTERED (rank r receives the sum of chunk r), each rank runs Adam on
    its chunks only (1/world of the optimizer's 28 bytes per parameter), and the updated chunks of the parameters --
    views of one flat buffer laid out like the gradient arena -- are ALL-GATHERED in place.  Same bytes on the wire as
    the all-reduce (which is a reduce-scatter plus an all-gather), the optimizer's HBM traffic divided by the world size.
    Adam's moments exist in full shape on every rank but are only maintained inside the rank's own chunks; gradient
    accumulation over several backward passes is not available in this form.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class DataParallelStep:
    def __init__(self, group=None, shard_optimizer=False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self._setup(dist.get_world_size(group), dist.get_rank(group), dist.get_backend(group) != "nccl")
        # C2_DP_SINGLE_RANK=1 (tests): a one-rank group still issues every collective -- the only way to run the RCCL
        # branch (stream-ordered exchange on the side stream, reduce-scatter / all-gather through the staging chunk) on a
        # box with one GPU.  Otherwise one rank exchanges nothing.
        self.active = self.world_size > 1 or os.environ.get("C2_DP_SINGLE_RANK") == "1"
        self.sharded = bool(shard_optimizer) and self.active

    def _setup(self, world_size, rank, host_backend):
        self.world_size, self.rank = world_size, rank
        # gloo (CPU tests, rehearsals with several ranks on one GPU): device tensors are exchanged through explicit host
        # copies (gloo's own handling of device tensors was seen to stall after a few steps with ranks sharing a GPU).
        # RCCL ("nccl") is stream-ordered and takes the asynchronous path.
        self._host_backend = host_backend
        self._side = None
        self._pending = []
        self.defer_last = False  # set by the optimizer wiring: the last bucket's wait moves into FusedAdam.step
        self._deferred = None    # (event, first byte, one-past-last byte) of the bucket still being reduced
        self.sharded = False
        self.active = world_size > 1
        self._arena = None       # (gradient arena, flat parameter buffer, buckets) of the attached engine (sharded form)

    # -- wiring
    def attach_model(self, model):
        model.world = self
        model.__dict__["_eng"] = None  # rebuild the engine with the communicator attached

    def attach(self, engine):
        engine.comm = self
        engine.world_size = self.world_size
        engine.rank = self.rank
        engine.exchange = self.active

    def shard(self, n_global):
        """Rows of a global batch owned by this rank (contiguous, equal shares; remainder to the last ranks)."""
        base, rem = divmod(n_global, self.world_size)
        sizes = [base + (1 if r >= self.world_size - rem else 0) for r in range(self.world_size)]
        start = sum(sizes[: self.rank])
        return start, start + sizes[self.rank]

    # -- the two primitives every collective below is made of (a test double overrides exactly these)
    def _sum_(self, t):
        """In-place SUM over ranks, enqueued on the current stream (RCCL) or blocking (gloo on host tensors)."""
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _gather0(self, t):
        """Every rank's `t` concatenated along dim 0 in rank order."""
        out = torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def _reduce_scatter_(self, flat):
        """Chunk `rank` of `flat` (world_size equal chunks) becomes the SUM over ranks of that chunk; the other chunks
        are left as they were (un-reduced).  The collective's in-place form: the output IS chunk `rank` of the input
        (receive address = send address + rank * count, which RCCL defines as in place), so there is no staging chunk
        and nothing to copy back."""
        n = flat.numel() // self.world_size
        dist.reduce_scatter_tensor(flat[self.rank * n:(self.rank + 1) * n], flat, op=dist.ReduceOp.SUM, group=self.group)

    def _all_gather_(self, flat):
        """Every rank's chunk `rank` of `flat` is distributed to all (in place: the input is that chunk of the output)."""
        n = flat.numel() // self.world_size
        dist.all_gather_into_tensor(flat, flat[self.rank * n:(self.rank + 1) * n], group=self.group)

    # -- the sharded form: arena bookkeeping
    def set_arena(self, grad_flat, param_flat, buckets):
        for lo, hi in buckets:
            assert (hi - lo) % (64 * self.world_size) == 0, "bucket does not cut into aligned chunks"
        self._arena = (grad_flat, param_flat, [(lo, hi) for lo, hi in buckets if hi > lo])

    def owned_ranges(self):
        """Byte-address ranges of the gradient arena this rank owns (one chunk per bucket)."""
        grad_flat, _, buckets = self._arena
        base = grad_flat.data_ptr()
        out = []
        for lo, hi in buckets:
            n = (hi - lo) // self.world_size
            out.append((base + 4 * (lo + self.rank * n), base + 4 * (lo + (self.rank + 1) * n)))
        return out

    def owned_parts(self, entries):
        """Adam entries (p, g, m, v, shadow) cut down to the elements whose gradient lies in this rank's chunks."""
        if self._arena is None:
            raise RuntimeError("sharded optimizer step before any backward pass built the gradient arena")
        grad_flat = self._arena[0]
        a_lo, a_hi = grad_flat.data_ptr(), grad_flat.data_ptr() + 4 * grad_flat.numel()
        owned = self.owned_ranges()
        out = []
        for p, g, m, v, _ in entries:
            g0 = g.data_ptr()
            if not (a_lo <= g0 < a_hi):
                raise NotImplementedError(
                    "shard_optimizer=True needs every .grad to be the engine's arena view (zero_grad(set_to_none=True), "
                    "one backward per step): accumulated copies hold un-reduced chunks")
            g1 = g0 + 4 * g.numel()
            for lo, hi in owned:
                s, e = max(g0, lo), min(g1, hi)
                if s < e:
                    i0, i1 = (s - g0) // 4, (e - g0) // 4
                    out.append((p.view(-1)[i0:i1], g.view(-1)[i0:i1], m.view(-1)[i0:i1], v.view(-1)[i0:i1], None))
        return out

    def owned_slices(self, p):
        """Element ranges [(i0, i1), ...] of the flattened parameter `p` (a view of the flat parameter buffer, which is
        laid out like the gradient arena) whose optimizer state this rank maintains; None before the arena exists."""
        if self._arena is None:
            return None
        _, param_flat, buckets = self._arena
        off = (p.data_ptr() - param_flat.data_ptr()) // 4
        if not (0 <= off and off + p.numel() <= param_flat.numel()):
            raise RuntimeError("parameter is not a view of the engine's flat parameter buffer")
        out = []
        for lo, hi in buckets:
            n = (hi - lo) // self.world_size
            s, e = max(off, lo + self.rank * n), min(off + p.numel(), lo + (self.rank + 1) * n)
            if s < e:
                out.append((s - off, e - off))
        return out

    def allgather_params(self):
        """After the sharded Adam: every bucket of the flat parameter buffer is completed from the ranks' chunks
        (stream-ordered on the compute stream: the next forward reads the parameters)."""
        _, param_flat, buckets = self._arena
        for lo, hi in buckets:
            seg = param_flat[lo:hi]
            if self._host_backend:  # (gloo: no in-place form; through a host copy, as the other rehearsal exchanges)
                n = (hi - lo) // self.world_size
                seg.copy_(self._gather0(seg[self.rank * n:(self.rank + 1) * n].cpu()).to(seg.device))
            else:
                self._all_gather_(seg)

    # -- forward collectives
    def allgather_rows(self, feat, cond):
        """[B,D] fp32 / [B,C] int64 of every rank stacked in rank order (equal B on all ranks)."""
        if self._host_backend and feat.is_cuda:
            return self._host_gather(feat), self._host_gather(cond)
        return self._gather0(feat), self._gather0(cond)

    # gloo with device tensors: the exchange is made on host copies (blocking), explicitly
    def _host_gather(self, t):
        return self._gather0(t.detach().contiguous().cpu()).to(t.device)

    def _host_reduce(self, t):
        h = t.detach().contiguous().cpu()
        self._sum_(h)
        t.copy_(h.to(t.device).view_as(t))
        return t

    def allgather_stats(self, part):
        """[2,H] fp32 statistics partial of every rank stacked in rank order -> [world,2,H]."""
        if self._host_backend and part.is_cuda:
            return self._host_gather(part).view((self.world_size,) + tuple(part.shape))
        return self._gather0(part).view((self.world_size,) + tuple(part.shape))

    def allreduce_scalars(self, t):
        if self._host_backend and t.is_cuda:
            return self._host_reduce(t)
        self._sum_(t)
        return t

    # -- backward collectives
    def reduce_bucket_async(self, flat):
        """The exchange of one gradient bucket: all-reduce, or reduce-scatter in the sharded form."""
        self.allreduce_bucket_async(flat, self._reduce_scatter_ if self.sharded else None)

    def allreduce_bucket_async(self, flat, collective=None):
        """SUM all-reduce (or `collective`) of one gradient bucket on a side stream, ordered after the work already
        queued.  (gloo rehearsals: the blocking host all-reduce serves both forms -- a reduce-scatter's result is a part
        of it.)"""
        collective = collective or self._sum_
        if flat.is_cuda and self._host_backend:
            self._host_reduce(flat)
        elif flat.is_cuda:
            if self._side is None:
                self._side = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(flat.device))
            self._side.wait_event(ev)
            with torch.cuda.stream(self._side):
                collective(flat)
                done = torch.cuda.Event()
                done.record(self._side)
            self._pending.append((done, flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()))
        else:
            self._sum_(flat)  # (host tensors: CPU tests)

    def wait_all(self):
        """Order the compute stream after the bucket all-reduces.  With `defer_last`, the newest bucket (the encoder's
        gene-width layer, reduced last) is left in flight: FusedAdam updates every other tensor first and calls
        wait_deferred() before it touches that bucket -- the optimizer overlaps the exposed tail of the exchange."""
        pend = self._pending
        self._pending = []
        self.wait_deferred()  # never replace a bucket that is still in flight without ordering after it
        if self.defer_last and len(pend) > 1:
            self._deferred = pend.pop()
        for ev, _, _ in pend:
            torch.cuda.current_stream().wait_event(ev)

    def forget_pending(self):
        """Drop the records of exchanges that were only recorded into a stream capture that has been abandoned."""
        self._pending = []
        self._deferred = None

    def deferred_range(self):
        return None if self._deferred is None else self._deferred[1:]

    def wait_deferred(self):
        if self._deferred is not None:
            torch.cuda.current_stream().wait_event(self._deferred[0])
            self._deferred = None
