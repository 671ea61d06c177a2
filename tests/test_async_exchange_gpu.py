"""The asynchronous gradient exchange (side stream, buckets in flight, last bucket deferred into the optimizer) on ONE
GPU, with a test double for the process group: a 2-rank world whose other rank holds the same rows, so SUM = 2x and
all-gather = (t, t), and whose every collective lands LATE (a spin kernel runs first on the stream it is enqueued on).
The step must then equal the single-process step on the doubled batch -- it does not if any consumer (the optimizer, a
clone for gradient accumulation, the next backward) reads a bucket before its exchange has landed.

The MMD is off in these runs: a twin world puts pairs of DISTINCT rows with identical latent codes into one batch,
which no real batch holds (eps differs per row) and where the kernel's contraction form of the gradient
(sum_j w_ij z_j - z_i sum_j w_ij, with w_ij ~ 1e6 at the 1e-6 bandwidth when the distance is ~0) cancels to rounding
noise instead of an exact zero; the MMD's own exchange (all-gather of latent rows) is covered by the real 2-rank runs of
tests/test_parallel_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _twin_world(delay):
    from celligner2_amd.parallel import DataParallelStep

    class TwinWorld(DataParallelStep):
        def __init__(self):
            self.group = None
            self._setup(2, 0, host_backend=False)
            self.n_sum = 0

        def _sum_(self, t):
            if t.is_cuda and delay:
                torch.cuda._sleep(delay)
            t.mul_(2)
            self.n_sum += 1

        def _gather0(self, t):
            return torch.cat([t, t], 0)

    return TwinWorld()


def _setup(use_bn):
    from oracle import c2_oracle as orc
    G, hidden, L, C, P, B = 320, [64, 32], 8, 3, 4, 48
    cfg = orc.Cfg(input_dim=G, hidden_layer_sizes=hidden, latent_dim=L, n_conditions=C, n_predictors=P,
                  recon_loss="nb", use_mmd=False, use_bn=use_bn)
    rng = np.random.default_rng(3)
    params = orc.init_params(cfg, 0, np.float32)
    x = rng.poisson(rng.uniform(0.5, 2, (B, 1)) * np.exp(rng.normal(1, 1, (1, G)))).astype(np.float32)
    cond = np.zeros((B, C), np.int64)
    cond[np.arange(B), rng.integers(0, C, B)] = 1
    cls = np.zeros((B, P), np.int64)
    cls[np.arange(B), rng.integers(0, P, B)] = 1
    cls[::5] = -1
    batch = dict(x=x, batch=cond, sizefactor=x.sum(1), goodloc=rng.uniform(size=(B, G)) > 0.01, classes=cls)
    eps = rng.standard_normal((B, L)).astype(np.float32)
    return (G, hidden, L, C, P, B), params, batch, eps


def _run(shape, params, batch, eps, use_bn, world, set_to_none, steps=3, accumulate=False):
    from celligner2_amd.model import Celligner2
    from celligner2_amd.optim import FusedAdam
    G, hidden, L, C, P, B = shape
    m = Celligner2(G, hidden, L, 0.0, False, "nb", use_bn, conditions=["a", "b", "c"], use_mmd=False,
                   predictors=list("wxyz"))
    m.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=False)
    m.to(DEV).train()
    if world is not None:
        world.attach_model(m)
    opt = FusedAdam(list(m.parameters()), lr=1e-2, eps=0.01, weight_decay=0.04, grad_sync=world)
    m._forced_noise = dict(eps=torch.tensor(eps).to(DEV), masks={})
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=set_to_none)
        for _ in range(2 if accumulate else 1):
            r, k, mm, ce = m(**{k_: torch.tensor(v).to(DEV) for k_, v in batch.items()})
            (r + 1e-3 * k + mm + ce).backward()
        losses.append([r.item(), k.item(), mm.item(), ce.item()])
        opt.step()
    torch.cuda.synchronize()
    return m, np.array(losses)


@pytest.mark.parametrize("use_bn", [False, True])
@pytest.mark.parametrize("set_to_none", [True, False])
def test_late_exchange_equals_single_process_on_the_doubled_batch(use_bn, set_to_none):
    shape, params, batch, eps = _setup(use_bn)
    two = {k: np.concatenate([v, v], 0) for k, v in batch.items()}
    ref, ref_l = _run(shape, params, two, np.concatenate([eps, eps], 0), use_bn, None, True)
    world = _twin_world(delay=30_000_000)
    got, got_l = _run(shape, params, batch, eps, use_bn, world, set_to_none)
    assert world.n_sum >= 3 * 3 and world.defer_last and world.deferred_range() is None
    assert np.allclose(got_l, ref_l, rtol=2e-5, atol=1e-6), (got_l, ref_l)
    rp = dict(ref.named_parameters())
    for n, p in got.named_parameters():
        den = max(rp[n].abs().max().item(), 1e-30)
        assert (p.detach() - rp[n].detach()).abs().max().item() / den < 2e-5, n
    for n, b in ref.named_buffers():
        assert torch.allclose(dict(got.named_buffers())[n].float(), b.float(), rtol=1e-4, atol=1e-5), n


def test_gradient_accumulation_over_two_backwards_with_late_exchange():
    """Two forward/backward pairs before one optimizer step: the second backward must not rewrite the bucket the first
    one still has in flight, and .grad must hold the SUM of both (the first hand-over is an arena view)."""
    shape, params, batch, eps = _setup(False)
    world = _twin_world(delay=30_000_000)
    acc, _ = _run(shape, params, batch, eps, False, world, True, steps=1, accumulate=True)
    one, _ = _run(shape, params, batch, eps, False, _twin_world(delay=0), True, steps=1)
    # Adam's first step moves every weight by lr * g/(|g| + eps*...) -- compare the raw gradients instead
    ga = {n: p.grad.clone() for n, p in acc.named_parameters() if p.grad is not None}
    go = {n: p.grad.clone() for n, p in one.named_parameters() if p.grad is not None}
    assert set(ga) == set(go) and len(ga) > 10
    for n in ga:
        den = max(go[n].abs().max().item(), 1e-30)
        assert (ga[n] - 2 * go[n]).abs().max().item() / den < 2e-5, n
