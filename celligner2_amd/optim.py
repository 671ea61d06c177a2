"""Fused multi-tensor Adam behind the torch.optim.Optimizer interface.

Semantics = torch.optim.Adam as the reference trainer builds it
(celligner2/trainers/celligner2/trainer.py:262-266): L2 weight decay folded into the gradient,
bias-corrected moments, `eps` added to sqrt(v_hat).  `param_groups[i]["lr"]` may be changed
between steps (trainer.py:365-368).  One kernel launch updates every tensor of a group and, in
bf16 mode, refreshes the bf16 weight shadows the GEMMs read.
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=0.0,
                 shadow_lookup=None, grad_sync=None, on_updated=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_value=clip_value)
        super().__init__(params, defaults)
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

    def _launch(self, group, entries, step_t):
        rng = self._grad_sync.deferred_range() if self._grad_sync is not None else None
        hyper = (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"],
                 group.get("clip_value", 0.0), step_t)
        if rng is not None:
            late = [e for e in entries if rng[0] <= e[1].data_ptr() < rng[1]]
            early = [e for e in entries if not (rng[0] <= e[1].data_ptr() < rng[1])]
            if early:
                ops.adam_step(early, *hyper)
            self._grad_sync.wait_deferred()
            entries = late
        if entries:
            ops.adam_step(entries, *hyper)

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
                    # tensors that joined later: flush what we have and start a new launch
                    self._launch(group, entries, step_t)
                    entries, step_t = [], st["step"]
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                sh = self._shadow_lookup(p) if self._shadow_lookup is not None else None
                entries.append((p.data, g, st["exp_avg"], st["exp_avg_sq"], sh))
                touched.append(p)
                if sh is not None:
                    refreshed.append(p)
            if entries:
                self._launch(group, entries, step_t)
        if self._grad_sync is not None:
            self._grad_sync.wait_deferred()  # nothing of the step may outlive it un-synchronised
        if touched:
            torch._C._autograd._unsafe_set_version_counter(tuple(touched), tuple(p._version + 1 for p in touched))
            if self._on_updated is not None:
                for p in refreshed:
                    self._on_updated(p)
        return loss
