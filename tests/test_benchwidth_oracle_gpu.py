"""The benchmark configurations themselves against the fp64 oracle: ONE full training step (four losses, every gradient
tensor, every parameter after the fused Adam step) at 19 000 genes -- C4 (hidden [1024, 512], latent 64, 3 conditions,
NB + MMD), C2 (hidden [256, 128], latent 16, 30 predictors through the classifier [32, 64], beta 500, batch 128) and C3
(hidden [820, 512, 256], latent 26) of bench.py / SURVEY.md 8(d) -- in both precision modes.

Reference step: celligner2/model/celligner2.py:259-374 (forward), torch autograd of it, trainer.py:262-266 + 316-323
(Adam), restated by oracle/c2_oracle.py, which tests/test_oracle_golden.py pins to the reference's own outputs.
Bounds are the north_star's: fp32 mode 1e-4, bf16-MFMA mode 1e-2, relative L2 per tensor (losses: relative).
The batch is 256 cells (C2: its own 128): the oracle needs ~10 s per case on the box's host cores at that size; the
4096-cell batch of the benchmark runs the SAME kernels on 16 times the rows (tests/test_fullsize_gpu.py covers that size
through size-independent properties).  Dropout masks and eps are handed to both sides."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = {
    "c4": dict(G=19000, hidden=[1024, 512], L=64, C=3, P=0, cls=[32], B=256, beta=1.0),
    "c2": dict(G=19000, hidden=[256, 128], L=16, C=2, P=30, cls=[32, 64], B=128, beta=500.0),
    "c3": dict(G=19000, hidden=[820, 512, 256], L=26, C=3, P=30, cls=[32, 64], B=256, beta=1.0),
}


def _case(name):
    from oracle import c2_oracle as orc
    c = CASES[name]
    G, L, C, P, B = c["G"], c["L"], c["C"], c["P"], c["B"]
    rng = np.random.default_rng(11)
    cfg = orc.Cfg(input_dim=G, hidden_layer_sizes=c["hidden"], latent_dim=L, n_conditions=C, n_predictors=P,
                  classifier_hidden_layer_sizes=c["cls"], recon_loss="nb", use_mmd=True, beta=c["beta"], dr_rate=0.05)
    params = orc.init_params(cfg, 3, np.float32)
    # counts as bench.py's synth_dataset draws them: per-gene log-normal means x per-cell depth, 0.5 % masked entries
    x = rng.poisson(rng.uniform(0.5, 2.0, (B, 1)) * np.exp(rng.normal(1.5, 1.5, (1, G)))).astype(np.float32)
    cond = np.zeros((B, C), np.int64)
    cond[np.arange(B), rng.integers(0, C, B)] = 1
    batch = dict(x=x, batch=cond, sizefactor=x.sum(1).astype(np.float32), goodloc=rng.uniform(size=(B, G)) > 0.005)
    if P:
        cls = np.zeros((B, P), np.int64)
        cls[np.arange(B), rng.integers(0, P, B)] = 1
        cls[rng.uniform(size=B) < 0.2] = -1
        batch["classes"] = cls
    eps = rng.standard_normal((B, L)).astype(np.float32)
    masks = {}
    widths = dict(zip(orc.dropout_names(cfg), _dropout_widths(cfg)))
    for k, w in widths.items():
        masks[k] = rng.uniform(size=(B, w)) >= 0.05
    return cfg, params, batch, eps, masks


def _dropout_widths(cfg):
    from oracle import c2_oracle as orc
    out = []
    for k in orc.dropout_names(cfg):
        part, i = k.split(".D")
        i = int(i)
        if part == "enc":
            out.append(cfg.hidden_layer_sizes[i])
        elif part == "dec":
            out.append(list(reversed(cfg.hidden_layer_sizes))[i])
        else:
            out.append(cfg.classifier_hidden_layer_sizes[i])
    return out


_REF = {}


def _reference(name):
    """fp64 oracle step (losses, gradients), computed once per case for both precision modes."""
    if name not in _REF:
        from oracle import c2_oracle as orc
        cfg, params, batch, eps, masks = _case(name)
        p64 = {k: v.astype(np.float64) for k, v in params.items()}
        ref = orc.forward_backward(p64, cfg, dict(batch), eps.astype(np.float64), masks=masks, gscale=(1, 1e-3, 1, 1))
        _REF[name] = dict(losses=np.array(ref["losses"], np.float64), grads=ref["grads"])
    return _REF[name]


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 1e-2)])
@pytest.mark.parametrize("name", ["c4", "c2", "c3"])
def test_benchmark_shape_step_matches_the_oracle(name, precision, tol):
    from celligner2_amd.model import Celligner2
    from celligner2_amd.optim import FusedAdam
    cfg, params, batch, eps, masks = _case(name)
    ref = _reference(name)
    c = CASES[name]
    m = Celligner2(c["G"], c["hidden"], c["L"], 0.05, False, "nb", conditions=[f"c{i}" for i in range(c["C"])],
                   use_mmd=True, mmd_on="z", beta=c["beta"], predictors=[f"p{i}" for i in range(c["P"])],
                   classifier_hidden_layer_sizes=c["cls"], precision=precision)
    m.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    m.to(DEV).train()
    m._forced_noise = dict(eps=torch.tensor(eps).to(DEV),
                           masks={k: torch.tensor(v.astype(np.uint8)).to(DEV) for k, v in masks.items()})
    opt = FusedAdam([p for p in m.parameters()], lr=1e-3, eps=0.01, weight_decay=0.04,
                    shadow_lookup=m._adam_hooks()[0], on_updated=m._adam_hooks()[1])
    r, k, mm, ce = m(**{k_: torch.tensor(v).to(DEV) for k_, v in batch.items()})
    (r + 1e-3 * k + mm + ce).backward()
    eng = m._engine()
    if precision == "bf16":
        assert eng.enc[0].gt == torch.bfloat16 and eng.head_t == torch.bfloat16
    eng.check_status()
    got = np.array([r.item(), k.item(), mm.item(), ce.item()])
    for i in range(4):
        assert abs(got[i] - ref["losses"][i]) <= tol * max(abs(ref["losses"][i]), 1.0), (i, got, ref["losses"])
    errs = {}
    for n, p in m.named_parameters():
        gr = ref["grads"].get(n)
        if gr is None:
            assert p.grad is None, n
            continue
        g = p.grad.double().cpu().numpy()
        errs[n] = float(np.linalg.norm(g - gr) / max(np.linalg.norm(gr), 1e-30))
    print(name, precision, "gradient errors (relative L2):",
          sorted(((float(f"{e:.3g}"), n) for n, e in errs.items()), reverse=True)[:6])
    assert all(e < tol for e in errs.values()), {n: float(f"{e:.3g}") for n, e in errs.items() if e >= tol}
    # the optimizer (trainer.py:262-266, 316-323): the fused Adam kernel on the gradients just checked against the oracle's
    # Adam on THE SAME gradients (read back from the GPU) -- every parameter after the step at fp32 rounding.  (Comparing
    # with the oracle's Adam on the oracle's own gradients would measure the gradient error a second time, amplified: the
    # first step moves an entry by lr * g / (|g| + eps), which is linear in g where |g| << eps = 0.01.)
    from oracle import c2_oracle as orc
    grads = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters() if p.grad is not None}
    want = {n: params[n].copy() for n in grads}
    orc.adam_step(want, grads, {}, lr=1e-3, eps=0.01, weight_decay=0.04)
    opt.step()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        if n in want:
            got_p = p.detach().cpu().numpy()
            assert np.abs(got_p - want[n]).max() <= 2e-7 * max(np.abs(want[n]).max(), 1e-3) + 1e-9, n
    if precision == "bf16":  # the operand copies the next forward reads were refreshed by the same kernel
        w0 = dict(m.named_parameters())[eng.enc[0].w]
        assert torch.equal(eng.weight(eng.enc[0].w), w0.detach().to(eng.fwd16 or torch.bfloat16))
        hw = dict(m.named_parameters())[eng.head + ".weight"]
        assert torch.equal(eng.weight(eng.head + ".weight"), hw.detach().to(torch.bfloat16))
