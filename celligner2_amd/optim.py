"""Fused multi-tensor Adam behind the torch.optim.Optimizer interface.

Semantics = torch.optim.Adam as the reference trainer builds it
(celligner2/trainers/celligner2/trainer.py:262-266): L2 weight decay folded into the gradient,
bias-corrected moments, `eps` added to sqrt(v_hat).  `param_groups[i]["lr"]` may be changed
between steps (trainer.py:365-368).  One kernel launch updates every tensor of a group and, in
bf16 mode, refreshes the bf16 weight shadows the GEMMs read.

`capturable=True` keeps the step count and the hyper-parameters in device memory (c2_adam_prepare /
c2_adam_step_dev), so that the launches of one step are the same bytes every step and the step can be
replayed as a HIP graph: `sync_hyper()` carries a changed `lr` over (outside the graph), `replayed()`
does the host-side bookkeeping of one replay.
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0,
                 shadow_lookup=None, grad_sync=None, on_updated=None, capturable=False):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_value=clip_value)
        super().__init__(params, defaults)
        self.capturable = bool(capturable)
        self._last = None  # (touched, refreshed) parameters of the latest step, for replayed()
        self._dev = {}     # id(group) -> device-resident step state
        self._shadow_lookup = shadow_lookup  # callable(param) -> bf16 shadow / planes tensor or None
        # The kernel writes the parameters through raw pointers; step() then advances each parameter's autograd version
        # counter (as an in-place torch op would), so that anything caching a function of the weights sees the change.
        # on_updated(param) is called right after that for parameters whose operand copy the kernel refreshed itself.
        self._on_updated = on_updated
        # data-parallel runs: object with deferred_range() / wait_deferred() (parallel.DataParallelStep).  Tensors whose
        # gradient lies in the bucket that is still being all-reduced are updated by a second launch, after the wait.
        self._grad_sync = grad_sync
        if grad_sync is not None:
            grad_sync.defer_last = True

    def _sharded(self):
        return self._grad_sync is not None and getattr(self._grad_sync, "sharded", False)

    def state_dict(self):
        """The usual dictionary.  In the sharded form (parallel.DataParallelStep(shard_optimizer=True)) every rank holds
        full-shape moment tensors but maintains only the elements of its own chunks, so the call is a COLLECTIVE there:
        every rank must make it; each contributes the elements it maintains (zeros elsewhere) to a SUM over the ranks, and
        every rank gets the complete moments back -- what the all-reduce form, or a single process, would have saved.
        `load_state_dict` needs nothing special: it puts whole moments on every rank, of which each goes on maintaining
        its own elements.  Before the first backward pass (no arena yet) the state is whatever was loaded, whole."""
        sd = super().state_dict()
        if not self._sharded():
            return sd
        params = [p for g in self.param_groups for p in g["params"]]
        state = {}
        for i, p in enumerate(params):
            st = sd["state"].get(i)
            if st is None:
                continue
            own = self._grad_sync.owned_slices(p)
            if own is None:  # no step yet: nothing is partial
                state[i] = st
                continue
            whole = dict(st)
            for key in ("exp_avg", "exp_avg_sq"):
                full = torch.zeros_like(st[key], memory_format=torch.contiguous_format)
                src = st[key].reshape(-1)
                for i0, i1 in own:
                    full.view(-1)[i0:i1].copy_(src[i0:i1])
                whole[key] = self._grad_sync.allreduce_scalars(full)
            state[i] = whole
        sd["state"] = state
        return sd

    @staticmethod
    def _hyper_of(group):
        return (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                group.get("clip_value", 0.0))

    def _device_state(self, group, step_before, device):
        """The group's device-resident step state; (re)built from the host's numbers when absent or when the host's
        step count has moved without it (a loaded state_dict)."""
        ds = self._dev.get(id(group))
        if ds is None or ds["step"] != step_before or ds["device"] != device:
            hyper = self._hyper_of(group)
            state, block = ops.adam_state(step_before, *hyper, device)
            ds = self._dev[id(group)] = dict(state=state, block=block, step=step_before, hyper=hyper, device=device)
        return ds

    def sync_hyper(self):
        """Write hyper-parameters the host has changed since the last step (the trainer's learning-rate decay,
        trainer.py:365-368) into the device state.  A few python compares per call when nothing changed."""
        for group in self.param_groups:
            ds = self._dev.get(id(group))
            if ds is not None and ds["hyper"] != self._hyper_of(group):
                ds["hyper"] = hyper = self._hyper_of(group)
                ds["state"][1:].copy_(torch.tensor(hyper, dtype=torch.float64), non_blocking=False)

    def replayed(self):
        """Host-side bookkeeping of ONE replay of a captured step(): step counts, version counters, operand-copy
        records -- what step() itself does after its launches."""
        for group in self.param_groups:
            ds = self._dev.get(id(group))
            if ds is not None:
                ds["step"] += 1
        if self._last is not None:
            touched, refreshed = self._last
            for p in touched:
                self.state[p]["step"] += 1
            self._publish(touched, refreshed)

    def step_counts(self):
        """Host-side step counts (per tensor, per group's device state): what a capture attempt that fails must put back."""
        return ([(p, self.state[p]["step"]) for g in self.param_groups for p in g["params"] if len(self.state[p])],
                [(id(g), self._dev[id(g)]["step"]) for g in self.param_groups if id(g) in self._dev])

    def restore_step_counts(self, saved):
        for p, n in saved[0]:
            self.state[p]["step"] = n
        for gid, n in saved[1]:
            self._dev[gid]["step"] = n

    def captured(self):
        """Call after a step() that ran under stream capture: its launches were recorded, not executed, so the step
        counts it advanced on the host are taken back (the first replay performs that step)."""
        for group in self.param_groups:
            ds = self._dev.get(id(group))
            if ds is not None:
                ds["step"] -= 1
        if self._last is not None:
            for p in self._last[0]:
                self.state[p]["step"] -= 1

    def _publish(self, touched, refreshed):
        if touched:
            torch._C._autograd._unsafe_set_version_counter(tuple(touched), tuple(p._version + 1 for p in touched))
            if self._on_updated is not None:
                for p in refreshed:
                    self._on_updated(p)

    def _launch(self, group, entries, step_t):
        if self.capturable:
            ds = self._device_state(group, step_t - 1, entries[0][0].device)
            if not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()
            ops.adam_prepare(ds["state"], ds["block"])
            ds["step"] = step_t
            run = lambda part: ops.adam_step_dev(part, ds["block"])  # noqa: E731
        else:
            hyper = (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                     group.get("clip_value", 0.0), step_t)
            run = lambda part: ops.adam_step(part, *hyper)  # noqa: E731
        if self._sharded():  # this rank updates only the elements whose reduced gradient it received
            entries = self._grad_sync.owned_parts(entries)
        rng = self._grad_sync.deferred_range() if self._grad_sync is not None else None
        if rng is not None:
            late = [e for e in entries if rng[0] <= e[1].data_ptr() < rng[1]]
            early = [e for e in entries if not (rng[0] <= e[1].data_ptr() < rng[1])]
            if early:
                run(early)
            self._grad_sync.wait_deferred()
            entries = late
        if entries:
            run(entries)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        touched, refreshed = [], []
        for group in self.param_groups:
            entries = []
            step_t = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                if step_t is None:
                    step_t = st["step"]
                elif step_t != st["step"]:
                    if self.capturable:
                        raise RuntimeError("capturable FusedAdam: every tensor of a group must have taken the same "
                                           "number of steps (one device-resident step count per group)")
                    # tensors that joined later: flush what we have and start a new launch
                    self._launch(group, entries, step_t)
                    entries, step_t = [], st["step"]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                # (sharded form: the kernel sees a chunk of the tensor only, so the operand copies are converted by the
                # engine after the all-gather -- the version counters advanced below tell it to)
                sh = self._shadow_lookup(p) if (self._shadow_lookup is not None and not self._sharded()) else None
                entries.append((p.data, g, st["exp_avg"], st["exp_avg_sq"], sh))
                touched.append(p)
                if sh is not None:
                    refreshed.append(p)
            if entries:
                self._launch(group, entries, step_t)
        if self._grad_sync is not None:
            self._grad_sync.wait_deferred()  # nothing of the step may outlive it un-synchronised
            if self._sharded() and touched:
                self._grad_sync.allgather_params()
        self._last = (touched, refreshed)
        self._publish(touched, refreshed)
        return loss
