"""Full training step (forward, backward, Adam) on the MI355X vs golden vectors produced by the
reference itself (tests/golden/*.npz, tools/gen_golden.py).  fp32 mode: 1e-4 relative (north_star);
bf16-MFMA mode: 1e-2 on the losses and on every gradient's relative max-norm error."""
import numpy as np
import pytest
import torch

from tests import golden_util as gu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build_model(meta, z, precision):
    from celligner2_amd.model import Celligner2
    kw = dict(meta["model_kwargs"])
    C, P = meta["n_conditions"], meta["n_predictors"]
    conds = [f"c{i}" for i in range(C)]
    main = meta.get("main_dataset")
    m = Celligner2(conditions=conds, predictors=[f"p{i}" for i in range(P)],
                   main_dataset=(conds[main] if main is not None else None), precision=precision, **kw)
    sd = {k: torch.tensor(v) for k, v in gu.group(z, "p0/").items()}
    assert set(sd) == set(m.state_dict()), "state_dict keys differ from the reference"
    for k, v in m.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)
    m.to(DEV)
    m.train()
    for n, p in m.named_parameters():
        p.requires_grad = n in meta["trainable"]
    if meta["frozen"]:
        m.freeze = True
    return m


@pytest.mark.parametrize("name", gu.step_cases())
@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 1e-2)])
def test_training_step_matches_reference(name, precision, tol):
    from celligner2_amd.optim import FusedAdam
    z, meta = gu.load(name)
    model = build_model(meta, z, precision)
    eng = model._engine()
    opt = FusedAdam([p for p in model.parameters() if p.requires_grad], lr=meta["adam"]["lr"], eps=meta["adam"]["eps"],
                    weight_decay=meta["adam"]["weight_decay"],
                    shadow_lookup=lambda p: eng.shadow_for_adam(model._pname[id(p)]))
    model._pname = {id(p): n for n, p in model.named_parameters()}
    batch_np = gu.group(z, "batch/")
    for s in range(meta["n_steps"]):
        batch = {k: torch.tensor(v).to(DEV) for k, v in batch_np.items()}
        masks = {k: torch.tensor(v).to(DEV) for k, v in gu.group(z, f"s{s}/mask/").items()}
        model._forced_noise = dict(eps=torch.tensor(z[f"s{s}/eps"]).to(DEV), masks=masks)
        recon, kl, mmd, ce = model(**batch)
        loss = recon + meta["alpha"] * kl + mmd + ce
        opt.zero_grad()
        loss.backward()
        eng.check_status()
        ref = z[f"s{s}/losses"]
        got = [recon.item(), kl.item(), mmd.item(), ce.item()]
        ltol = tol if precision == "fp32" else 2e-2
        for i in range(4):
            assert abs(got[i] - ref[i]) <= ltol * max(abs(ref[i]), 1.0), (i, got, ref)
        for k, g in gu.group(z, f"s{s}/grad/").items():
            p = dict(model.named_parameters())[k]
            assert p.grad is not None, k
            err = gu.relerr(p.grad.cpu().numpy(), g)
            assert err < (tol if precision == "fp32" else 3e-2), (k, err)
        if "classes" in batch:
            assert np.array_equal(batch["classes"].cpu().numpy(), z[f"s{s}/classes_after"])
        opt.step()
        if precision == "fp32":
            for k, pref in gu.group(z, f"s{s}/param/").items():
                p = dict(model.named_parameters())[k]
                assert gu.relerr(p.detach().cpu().numpy(), pref) < 1e-4, (k, s)
        else:  # bf16 shadows follow the fp32 master weights bit for bit
            for n, sh in eng._shadow.items():
                assert torch.equal(sh, dict(model.named_parameters())[n].detach().to(torch.bfloat16)), n


def test_state_dict_and_init_match_reference_layout():
    """Same seed -> same parameter names, shapes and creation order as the reference module."""
    from celligner2_amd.model import Celligner2
    z, meta = gu.load("nb_mmd_z_dropout")
    kw = dict(meta["model_kwargs"])
    torch.manual_seed(0)
    m = Celligner2(conditions=["a", "b", "c"], predictors=list("wxyz"), **kw)
    assert list(m.state_dict().keys()) == [k for k in z.files if k.startswith("p0/")] or \
        set(m.state_dict().keys()) == {k[3:] for k in z.files if k.startswith("p0/")}


def test_eval_forward_no_grad_and_get_latent():
    z, meta = gu.load("nb_basic")
    model = build_model(meta, z, "fp32")
    model.eval()
    batch = {k: torch.tensor(v).to(DEV) for k, v in gu.group(z, "batch/").items()}
    model._forced_noise = dict(eps=torch.tensor(z["s0/eps"]).to(DEV), masks={})
    with torch.no_grad():
        out = model(**batch)
    ref = z["s0/losses"]  # dr_rate = 0 in this case: eval == train forward
    for i in range(4):
        assert abs(out[i].item() - ref[i]) <= 1e-4 * max(abs(ref[i]), 1.0)
    from oracle import c2_oracle as orc
    cfg = gu.cfg_from_meta(meta)
    params = {k: v.astype(np.float64) for k, v in gu.group(z, "p0/").items()}
    o = orc.forward_backward(params, cfg, gu.group(z, "batch/"), z["s0/eps"].astype(np.float64), want_grads=False)
    lat, cls = model.get_latent(batch["x"], batch["batch"], mean=True, add_classpred=True)
    assert gu.relerr(lat.cpu().numpy(), o["mu"]) < 1e-4
    lat2, _ = model.get_latent(batch["x"], batch["batch"], mean=False)
    assert gu.relerr(lat2.cpu().numpy(), o["z"]) < 1e-4
    rec, y1 = model.reconstructLatent(lat2, batch["batch"])
    e = np.exp(o["logits"] - o["logits"].max(1, keepdims=True))
    assert gu.relerr(rec.cpu().numpy(), e / e.sum(1, keepdims=True)) < 1e-4
    assert gu.relerr(y1.cpu().numpy(), o["y1"]) < 1e-4
