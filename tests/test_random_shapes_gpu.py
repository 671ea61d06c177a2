"""Random model shapes through the HIP step against the fp64 oracle: a pinned slice of tools/fuzz_configs.py (which
draws gene counts, widths, latent sizes and batches on both sides of every dispatch rule of the engine -- multiples
and non-multiples of 4 / 8 / 64, gene-width and narrow layers, batches of 2 to 700 rows, NB / ZINB / MSE, MMD on z or y,
LayerNorm / BatchNorm, with and without conditions and predictors).  Per case: the four losses, every gradient tensor,
the losses after one fused Adam step, bit-equality of the next step with a fresh module built from the updated
state_dict (no stale operand copies), and the inference path (get_latent / reconstructLatent) in eval mode.  fp32 mode: 1e-4 on losses, 2e-3 on gradients (3e-2 for a case in which a
unit sits on the other side of a ReLU than in the oracle, which the tool counts); bf16 mode: 1e-2 on losses and a loose
bound on gradients (30 %: the sweep looks for refusals, non-finite values and structured errors there -- the 1e-2 parity
bound of bf16 mode is test_step_gpu.py's, at the pinned realistic shapes; narrow LayerNorms and batches of a few rows
leave less averaging of the operands' rounding).  Seven sweeps of 30-60 cases (seeds 0-7, 2 x 370 steps) found no
refusal and no structured error; what they did find is written where the tool sets its bounds."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location(
    "fuzz_configs", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_configs.py"))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_match_the_oracle(seed):
    fz = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(fz)
    rng = np.random.default_rng(seed)
    failures = []
    for i in range(12):
        c = fz.draw_case(rng)
        for prec in ("fp32", "bf16"):
            bad, _ = fz.run_case(c, prec, seed * 1000 + i)
            if bad:
                failures.append((i, prec, c, bad))
    assert not failures, failures


def test_gene_count_not_a_multiple_of_8_at_a_large_batch():
    """Found by tools/fuzz_trainer.py --big (17 948 genes, batch 4096, bf16 mode): the decoder's weight gradient took the
    transposed-narrow-operand entry point (c2_linear_wgrad_bt), which refuses a gene-width operand whose rows do not start
    on 16 bytes, instead of c2_linear_wgrad, which falls back by itself.  Smallest shape that reaches that dispatch."""
    fz = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(fz)
    c = dict(G=2052, hidden=[1024, 64], L=10, C=2, P=0, B=1024, recon="nb", use_mmd=False, mmd_on="z", use_bn=False,
             own_kl=False, cls_hidden=[32], freeze=0)
    for prec in ("fp32", "bf16"):
        bad, _ = fz.run_case(c, prec, 5)
        assert not bad, (prec, bad)


def test_two_part_weight_gradient_with_a_strip_below_the_256_kernel():
    """Found by tools/fuzz_trainer.py --big (17 840 genes, hidden [976, 240], batch 4096, bf16 mode): the decoder's weight
    gradient [17840, 976] is 280 tiles of 256x256 -- one full round of the chip plus a strip of 1456 x 976, which is below
    the 256x256 kernel's size and was refused by the transposed-operand entry point.  Same dispatch at a size the oracle
    reaches is not available (the two-part split needs > 256 tiles), so the step is run at that size in both precisions and
    the bf16 gradients are held against fp32 mode's."""
    import torch
    from celligner2_amd.model import Celligner2
    G, hidden, L, C, B = 17840, [976, 240], 23, 4, 4096
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    x = torch.poisson(torch.exp(torch.randn(1, G, device="cuda", generator=g) + 0.5).expand(B, G).contiguous(), generator=g)
    cond = torch.zeros(B, C, dtype=torch.int64, device="cuda")
    cond[torch.arange(B, device="cuda"), torch.randint(0, C, (B,), device="cuda", generator=g)] = 1
    eps = torch.randn(B, L, device="cuda", generator=g)
    batch = dict(x=x, batch=cond, sizefactor=x.sum(1), goodloc=torch.ones(B, G, dtype=torch.bool, device="cuda"))
    grads = {}
    for prec in ("fp32", "bf16"):
        torch.manual_seed(3)
        m = Celligner2(G, hidden, L, 0.0, False, "nb", conditions=[f"c{i}" for i in range(C)], precision=prec).cuda().train()
        m._forced_noise = dict(eps=eps, masks={})
        r, k, mm, ce = m(**batch)
        (r + 1e-3 * k + mm + ce).backward()
        m._engine().check_status()
        grads[prec] = {n: p.grad.double() for n, p in m.named_parameters() if p.grad is not None}
    for n, g32 in grads["fp32"].items():
        e = float((grads["bf16"][n] - g32).norm() / g32.norm().clamp_min(1e-30))
        assert e < 3e-2, (n, e)


@pytest.mark.parametrize("recon", ["nb", "zinb"])
@pytest.mark.parametrize("precision,G", [("fp32", 2051), ("bf16", 2051), ("bf16", 2050), ("fp32", 701)])
def test_nb_on_resident_rows_at_a_gene_count_that_is_not_a_multiple_of_4(precision, G, recon):
    """The NB likelihood at a gene count that is not a multiple of 4 runs over the count rounded up to 4, on rows that are
    that far apart with zero / masked pad columns (-inf logits and zero dispersion rows in the pad: engine._forward), so that
    the loss kernel's 16-byte row walk applies.  The dataset keeps its counts and mask in that form ONCE (constructor,
    from_tensors(repitch=True): no second copy), gathered batches keep the pitch, contiguous tensor batches are padded
    on the way in.  All three are pinned against the same step on the general kernel (C2_ROWS4=0 / engine.rows4_on) on
    unpadded operands: losses and every gradient (two kernels, two summation orders) -- and the evaluation forward too."""
    import torch
    from celligner2_amd.dataset import DeviceDataset
    from celligner2_amd.model import Celligner2
    N, B, L, C = 600, 256, 12, 3
    G4 = (G + 3) // 4 * 4
    LD = (G + 127) // 128 * 128   # ops.rows_pitch: the dataset's rows start on 128-byte lines
    g = torch.Generator(device="cuda")
    g.manual_seed(G)
    X = torch.poisson(torch.exp(torch.randn(1, G, device="cuda", generator=g) + 0.5).expand(N, G).contiguous(), generator=g)
    good = torch.rand(N, G, device="cuda", generator=g) > 0.01
    cond = torch.zeros(N, C, dtype=torch.int64, device="cuda")
    cond[torch.arange(N, device="cuda"), torch.randint(0, C, (N,), device="cuda", generator=g)] = 1
    data = DeviceDataset.from_tensors(X, X.sum(1), good, cond, None, repitch=True)
    assert data.data.stride(0) == LD and data.goodloc.stride(0) == LD and torch.equal(data.data, X)
    assert data.data.untyped_storage().nbytes() == N * LD * 4  # the padded matrix and nothing else
    plain = DeviceDataset.from_tensors(X, X.sum(1), good, cond, None)  # contiguous rows: the legacy form
    idx = torch.randperm(N, device="cuda", generator=g)[:B]
    eps = torch.randn(B, L, device="cuda", generator=g)
    torch.manual_seed(1)
    m = Celligner2(G, [96, 40], L, 0.0, False, recon, conditions=["a", "b", "c"], use_mmd=True, beta=5.0,
                   precision=precision).cuda().train()
    eng = m._engine()
    spec = eng.input_spec()

    def step(batch, rows4=True, train=True):
        eng.rows4_on = rows4
        m.train(train)
        m._forced_noise = dict(eps=eps, masks={})
        for p in m.parameters():
            p.grad = None
        try:
            if not train:
                with torch.no_grad():
                    return torch.stack(m(**batch)).clone(), None
            total, v = m.forward_total((1.0, 1e-3, 1.0, 1.0), **batch)
            total.backward(gradient=torch.ones((), device="cuda"))
            eng.check_status()
            return v.clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
        finally:
            eng.rows4_on = True
            m.train(True)

    res = dict(data.gather(idx, apply_log=True, nb_const=True, resident_rows=True, **spec))
    assert res["x_full"] is data.data and res["goodloc_full"] is data.goodloc and "_rows4" not in data.__dict__
    gat = dict(data.gather(idx, apply_log=True, nb_const=True, resident_rows=False, **spec))
    assert gat["x"].stride(0) == LD and gat["goodloc"].stride(0) == LD and torch.equal(gat["x"], X[idx])
    assert torch.equal(gat["goodloc"], good[idx])
    full = gat["x"].as_strided((B, LD), (LD, 1), gat["x"].storage_offset())
    assert float(full[:, G:].abs().max()) == 0.0  # the pad travelled with the rows
    ten = dict(plain.gather(idx, apply_log=True, nb_const=True, resident_rows=False, **spec))
    ten["x"], ten["goodloc"] = ten["x"].contiguous(), ten["goodloc"].contiguous()
    assert ten["x"].stride(0) == G
    v0, g0 = step(ten, rows4=False)  # the general kernel on unpadded operands
    tol = 2e-4 if precision == "fp32" else 2e-2  # (bf16 mode: the two kernels round dlogits from sums in two orders)
    for name, batch in (("resident", res), ("gathered", gat), ("tensors", ten)):
        v1, g1 = step(batch)
        assert torch.isfinite(v1).all() and torch.allclose(v1, v0, rtol=2e-5, atol=1e-6), (name, v1, v0)
        for n in g0:
            e = float((g1[n].double() - g0[n].double()).norm() / g0[n].double().norm().clamp_min(1e-30))
            assert e < tol, (name, n, e)
    # the evaluation forward (validation, trainer.py:330-352) takes the same route
    e0, _ = step(ten, rows4=False, train=False)
    for name, batch in (("resident", res), ("gathered", gat), ("tensors", ten)):
        e1, _ = step(batch, train=False)
        assert torch.isfinite(e1).all() and torch.allclose(e1, e0, rtol=2e-5, atol=1e-6), (name, e1, e0)
    # the legacy form still works: padded copies made on first use, the dataset then holds both
    leg = dict(plain.gather(idx, apply_log=True, nb_const=True, resident_rows=True, **spec))
    assert leg["x_full"].stride(0) == G4 and "_rows4" in plain.__dict__
    v2, _ = step(leg)
    assert torch.allclose(v2, v0, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("L,mmd_on", [(26, "z"), (10, "z"), (7, "z"), (12, "y")])
def test_mmd_at_a_feature_width_that_is_not_a_multiple_of_4(L, mmd_on):
    """The MMD's four-wave kernel reads 16-byte feature chunks; a latent width like the reference's 26 (or its default 10) is
    widened with zero columns inside the engine (engine._forward) instead of falling to the general pair-tile kernel.  Same
    loss and gradients as the general kernel on the unpadded features (C2_ROWS4=0 / engine.rows4_on), two summation orders."""
    import torch
    from celligner2_amd.model import Celligner2
    G, B, C = 256, 192, 3
    g = torch.Generator(device="cuda")
    g.manual_seed(L)
    x = torch.poisson(torch.full((B, G), 2.0, device="cuda"), generator=g)
    cond = torch.zeros(B, C, dtype=torch.int64, device="cuda")
    cond[torch.arange(B, device="cuda"), torch.randint(0, C, (B,), device="cuda", generator=g)] = 1
    eps = torch.randn(B, L, device="cuda", generator=g)
    torch.manual_seed(2)
    hidden = [64, 50] if mmd_on == "y" else [64, 32]   # (mmd_on='y': the decoder's first layer's output, 50 wide here)
    m = Celligner2(G, hidden, L, 0.0, False, "nb", conditions=["a", "b", "c"], use_mmd=True, mmd_on=mmd_on, beta=20.0).cuda().train()
    eng = m._engine()
    out = {}
    for fast in (True, False):
        eng.rows4_on = fast
        m._forced_noise = dict(eps=eps, masks={})
        for p in m.parameters():
            p.grad = None
        r, k, mm, ce = m(x=x, batch=cond, sizefactor=x.sum(1))
        (r + 1e-3 * k + mm + ce).backward()
        eng.check_status()
        out[fast] = (mm.item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    eng.rows4_on = True
    assert ("mmd.feat4" in eng._ws) == ((L if mmd_on == "z" else hidden[-1]) % 4 != 0)
    assert abs(out[True][0] - out[False][0]) <= 2e-5 * max(abs(out[False][0]), 1e-3), (out[True][0], out[False][0])
    for n, g0 in out[False][1].items():
        e = float((out[True][1][n].double() - g0.double()).norm() / g0.double().norm().clamp_min(1e-30))
        assert e < 2e-4, (n, e)
