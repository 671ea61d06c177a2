"""The training iteration replayed as a HIP graph (Trainer(graph=True)) against the same iteration launched the ordinary
way: same seeds, same rows, same draws -> the same weights, optimizer state and epoch logs.  Plus the device-scalar
entry points the captured step is made of (c2_combine_losses_dev, c2_scale4_dev, c2_dropout_masks_dev,
c2_adam_prepare / c2_adam_step_dev) against their by-value twins, bit for bit."""
import numpy as np
import pytest
import torch

from tests.test_trainer_gpu import make_adata

pytestmark = pytest.mark.gpu


def _run(graph, n_epochs, precision="fp32", use_bn=False, dr=0.1, anneal=None, batch=32, n=150, lr_decay_at=None):
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer
    np.random.seed(0)
    adata = make_adata(n, 64, 3)
    torch.manual_seed(5)
    model = Celligner2(64, [32, 16], 8, dr, False, "nb", conditions=["ccle", "gtex", "tcga"], use_mmd=True, beta=3.0,
                       predictors=["lung", "skin", "blood"], use_bn=use_bn, use_ln=not use_bn, precision=precision)

    class T(Celligner2Trainer):
        def on_epoch_begin(self, lr, eps):
            if lr_decay_at is not None and self.epoch == lr_decay_at:
                for g in self.optimizer.param_groups:
                    g["lr"] *= 0.5

    kw = dict(condition_keys=["dataset"], predictor_keys=["lineage"], use_leiden=False, batch_size=batch,
              monitor=False, train_frac=0.8, seed=11, rng="device", use_early_stopping=False, graph=graph)
    if anneal == "iter":
        kw["alpha_iter_anneal"] = 7
    elif anneal == "epoch":
        kw["alpha_epoch_anneal"] = 3
    tr = T(model, adata, **kw)
    tr.train(n_epochs=n_epochs, lr=1e-2)
    return model, tr


@pytest.mark.parametrize("case", ["plain", "iter_anneal", "epoch_anneal_lr_decay", "batchnorm", "bf16"])
def test_replayed_step_equals_the_launched_step(case):
    kw = dict(plain={}, iter_anneal=dict(anneal="iter"), epoch_anneal_lr_decay=dict(anneal="epoch", lr_decay_at=2),
              batchnorm=dict(use_bn=True), bf16=dict(precision="bf16"))[case]
    m0, t0 = _run(False, 4, **kw)
    m1, t1 = _run(True, 4, **kw)
    # 120 training rows, batches of 32: three replayable batches and a short one (launched the ordinary way) per epoch
    assert len(t0.train_data) % 32 != 0 and t1.iters_per_epoch == t0.iters_per_epoch
    for (k, a), (_, b) in zip(m0.state_dict().items(), m1.state_dict().items()):
        assert torch.equal(a, b), f"{case}: {k} differs by {(a.float() - b.float()).abs().max().item():.3e}"
    for k in t0.logs:
        assert np.allclose(t0.logs[k], t1.logs[k], rtol=1e-6, atol=1e-9), (case, k, t0.logs[k], t1.logs[k])
    s0 = [t0.optimizer.state[p]["step"] for p in t0.optimizer.param_groups[0]["params"]]
    s1 = [t1.optimizer.state[p]["step"] for p in t1.optimizer.param_groups[0]["params"]]
    assert s0 == s1 and s0[0] == 4 * t0.iters_per_epoch


def test_graph_is_one_launch_per_full_batch():
    """After the warm-up and capture, a full batch costs the host one index copy and one graph launch: the engine's
    forward is not entered again."""
    from celligner2_amd import engine
    calls = []
    orig = engine.StepEngine.forward

    def counting(self, *a, **k):
        calls.append(1)
        return orig(self, *a, **k)

    engine.StepEngine.forward = counting
    try:
        _, tr = _run(True, 3, batch=30, n=150)  # 120 rows / 30: four full batches per epoch, no short one
    finally:
        engine.StepEngine.forward = orig
    n_val = 3 * int(np.ceil(len(tr.valid_data) / 30))
    # two warm-up steps + the capture pass enter the engine; the other 9 iterations are replays
    assert len(calls) - n_val == 3, (len(calls), n_val)
    assert len(tr.logs["epoch_loss"]) == 3 and np.isfinite(tr.logs["epoch_loss"]).all()


def test_graph_option_refuses_what_it_cannot_replay():
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer
    np.random.seed(0)
    adata = make_adata(60, 64, 3)
    model = Celligner2(64, [32, 16], 8, 0.0, False, "nb", conditions=["ccle", "gtex", "tcga"])
    for bad in (dict(rng="reference"), dict(rng="device", log_sync="step")):
        with pytest.raises(ValueError, match="graph=True"):
            Celligner2Trainer(model, adata, condition_keys=["dataset"], use_leiden=False, graph=True, **bad)


# ------------------------------------------------------------------ the device-scalar entry points
def test_combine_and_scale4_with_device_weights_are_the_by_value_kernels():
    from celligner2_amd import ops
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    losses = torch.randn(4, generator=g).to(dev)
    w = [1.0, 0.00037, 1.0, 1.0]
    wd = torch.tensor(w, dtype=torch.float32, device=dev)
    a, b = torch.empty(6, device=dev), torch.empty(6, device=dev)
    sums = torch.zeros(6, dtype=torch.float64, device=dev)
    ops.combine_losses(losses, w, a)
    for _ in range(3):
        ops.combine_losses_dev(losses, wd, b, sums)
    assert torch.equal(a, b)
    assert torch.equal(sums, 3 * a.double())
    up = torch.tensor(0.25, device=dev)
    x, y = torch.empty(4, device=dev), torch.empty(4, device=dev)
    for seed in (None, up):
        ops.scale4(seed, w, x)
        ops.scale4_dev(seed, wd, y)
        assert torch.equal(x, y)


def test_backward_seed_with_its_two_scaled_vectors_in_one_launch():
    """c2_scale4_dev_scale2 = c2_scale4_dev followed by two c2_scale launches with out4[0], bit for bit."""
    from celligner2_amd import ops
    dev = "cuda"
    w = torch.tensor([0.75, 1e-3, 2.0, 0.5], device=dev)
    x1, x2 = torch.randn(19001, device=dev), torch.randn(19000 * 3, device=dev)
    for up in (None, torch.tensor(0.25, device=dev)):
        ref4 = ops.scale4_dev(up, w, torch.empty(4, device=dev))
        r1, r2 = torch.empty_like(x1), torch.empty_like(x2)
        ops.scale(x1, ref4[0:1], r1)
        ops.scale(x2, ref4[0:1], r2)
        for jobs in ([(x1, torch.empty_like(x1)), (x2, torch.empty_like(x2))], [(x2, torch.empty_like(x2))], []):
            got4 = ops.scale4_dev_scale2(up, w, torch.empty(4, device=dev), jobs)
            assert torch.equal(got4, ref4)
            for (x, y) in jobs:
                assert torch.equal(y, r1 if x is x1 else r2)


def test_dropout_masks_with_a_device_counter_continue_the_host_counters_stream():
    from celligner2_amd import ops
    n = 16 * 1000 + 5
    a, b = (torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(2))
    ctr = torch.tensor([12345], dtype=torch.int64, device="cuda")
    adv = (n + 15) // 16
    for k in range(3):
        ops.dropout_masks(a, 0.3, 99, 12345 + k * adv)
        ops.dropout_masks_dev(b, 0.3, 99, ctr, adv)
        assert torch.equal(a, b), k
    assert int(ctr.item()) == 12345 + 3 * adv
    # a rank's stretch of a data-parallel step's counter range: masks from counter + lead, counter += world x blocks
    ctr = torch.tensor([500], dtype=torch.int64, device="cuda")
    for k in range(2):
        ops.dropout_masks(a, 0.3, 99, 500 + k * 4 * adv + 2 * adv)
        ops.dropout_masks_dev(b, 0.3, 99, ctr, 4 * adv, lead=2 * adv)
        assert torch.equal(a, b), k
    assert int(ctr.item()) == 500 + 8 * adv
    assert abs(float(b.float().mean()) - (1 - 77 / 256)) < 0.02  # thresh8 = round(0.3*256) = 77


def test_one_launch_noise_masks_normals_and_counter():
    """c2_noise_dev: masks equal to c2_dropout_masks at the same counter values; eps standard normal, fresh on every call,
    a rank's rows equal to the same rows of the one-process draw; the counter advanced by the launch itself."""
    from celligner2_amd import ops
    n = 16 * 700 + 9
    B, L, W = 96, 26, 4   # (B * L not a multiple of 4: ranks' element ranges start inside a Philox block)
    adv_m = (n + 15) // 16
    adv = max(W * adv_m, (W * B * L + 3) // 4)
    ref = torch.empty(n, dtype=torch.uint8, device="cuda")
    state = torch.tensor([777, 0], dtype=torch.int64, device="cuda")
    whole = torch.empty(W * B, L, device="cuda")
    seen = []
    for k in range(3):
        # one process over the global batch ...
        st1 = torch.tensor([777 + k * adv, 0], dtype=torch.int64, device="cuda")
        ops.noise_dev(None, 0.0, whole, 0, 99, st1, advance=adv)
        assert st1.tolist() == [777 + (k + 1) * adv, 0]
        # ... and rank 2 of 4 (its own state advances as every rank's does)
        m = torch.empty(n, dtype=torch.uint8, device="cuda")
        e = torch.empty(B, L, device="cuda")
        ops.noise_dev(m, 0.3, e, 2 * B * L, 99, state, advance=adv, lead=2 * adv_m)
        ops.dropout_masks(ref, 0.3, 99, 777 + k * adv + 2 * adv_m)
        assert torch.equal(m, ref), k
        assert torch.equal(e, whole[2 * B:3 * B]), k
        seen.append(whole.clone())
    assert state.tolist() == [777 + 3 * adv, 0]
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    # the normals: moments of 2^20 draws, and no correlation between neighbours (the two Box-Muller outputs of a pair)
    big = torch.empty(1 << 20, device="cuda")
    ops.noise_dev(None, 0.0, big, 0, 5, torch.tensor([0, 0], dtype=torch.int64, device="cuda"), advance=1 << 18)
    d = big.double()
    assert abs(float(d.mean())) < 4e-3 and abs(float(d.var()) - 1) < 6e-3
    assert abs(float((d ** 3).mean())) < 2e-2 and abs(float((d ** 4).mean()) - 3) < 5e-2
    assert abs(float((d[0::2] * d[1::2]).mean())) < 4e-3 and abs(float((d[:-4] * d[4:]).mean())) < 4e-3
    assert float(d.abs().max()) < 6.0 and bool(torch.isfinite(big).all())
    # a launch with nothing to draw still advances; an empty call with no advance is a no-op
    st = torch.tensor([5, 0], dtype=torch.int64, device="cuda")
    ops.noise_dev(None, 0.0, None, 0, 1, st, advance=7)
    ops.noise_dev(None, 0.0, None, 0, 1, st, advance=0)
    assert st.tolist() == [12, 0]


@pytest.mark.parametrize("shadow", [False, True])
def test_adam_with_device_state_is_the_host_scalar_step(shadow):
    from celligner2_amd import ops
    dev = "cuda"
    g = torch.Generator().manual_seed(1)
    shapes = [(33, 7), (128,), (5, 64)]

    def make():
        ps = [torch.randn(s, generator=g).to(dev) for s in shapes]
        return ps, [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]

    pa, ma, va = make()
    pb, mb, vb = [p.clone() for p in pa], [m.clone() for m in ma], [v.clone() for v in va]
    sa = [p.to(torch.bfloat16) if shadow else None for p in pa]
    sb = [p.to(torch.bfloat16) if shadow else None for p in pb]
    hyper = dict(lr=3e-3, beta1=0.9, beta2=0.999, eps=0.01, weight_decay=0.04, clip_value=0.5)
    state, block = ops.adam_state(0, hyper["lr"], hyper["beta1"], hyper["beta2"], hyper["eps"], hyper["weight_decay"],
                                  hyper["clip_value"], torch.device(dev))
    for t in range(1, 6):
        grads = [torch.randn(s, generator=g).to(dev) for s in shapes]
        if t == 4:  # a learning-rate change between steps: a plain device write
            hyper["lr"] *= 0.1
            state[1:2].fill_(hyper["lr"])
        ops.adam_step([(p, gr, m, v, s) for p, gr, m, v, s in zip(pa, grads, ma, va, sa)], step_t=t, **hyper)
        ops.adam_prepare(state, block)
        ops.adam_step_dev([(p, gr, m, v, s) for p, gr, m, v, s in zip(pb, grads, mb, vb, sb)], block)
        for x, y in zip(pa + ma + va + (sa if shadow else []), pb + mb + vb + (sb if shadow else [])):
            assert torch.equal(x, y), t
    assert ops.adam_state_step(state) == 5
