"""Generate golden vectors from the REFERENCE itself (build container only).

Runs the reference's `Celligner2` module (imported through tools/ref_harness.py)
for a handful of small configurations and writes, per case, one
``tests/golden/<case>.npz`` holding plain arrays + a JSON blob: configuration,
initial state_dict, batch, every random draw (eps, dropout masks), the four
losses, all parameter gradients and the parameters / Adam moments after each
optimiser step (trainer.py:305-323 semantics).

Two harness-side patches, both recorded in the fixture metadata:
  * randomness is drawn by the harness (torch CPU generator) and recorded, by
    replacing ``torch.nn.functional.dropout`` and
    ``torch.distributions.normal._standard_normal`` with recording versions;
  * the MSE branch's ``Tensor.sum(where=...)`` (TypeError on torch 2.x) is given
    its intended meaning through a wrapper object (masked row sum) -- the
    "defined deviation" of SURVEY.md Appendix B.2.

Usage: python tools/gen_golden.py            (needs /root/reference)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def make_batch(rng, B, G, C, P, multi_hot=False, nan_frac=0.01, unlabeled=0.25, mse=False):
    depth = rng.uniform(0.5, 2.0, size=(B, 1))
    gene = np.exp(rng.normal(1.0, 1.0, size=(1, G)))
    x = rng.poisson(depth * gene).astype(np.float32)
    if mse:
        x = np.log1p(x).astype(np.float32)
    goodloc = rng.uniform(size=(B, G)) > nan_frac
    cond = np.zeros((B, C), np.int64)
    lab = rng.integers(0, C, size=B)
    lab[:C] = np.arange(C)  # every condition present
    cond[np.arange(B), lab] = 1
    if multi_hot:  # a second condition key: rows with two ones
        extra = rng.integers(0, C, size=B)
        cond[np.arange(B), extra] = 1
    batch = dict(x=x, batch=cond, sizefactor=x.sum(1).astype(np.float32), goodloc=goodloc)
    if P > 0:
        cls = np.zeros((B, P), np.int64)
        cls[np.arange(B), rng.integers(0, P, size=B)] = 1
        cls[rng.uniform(size=B) < unlabeled] = -1
        batch["classes"] = cls
    return batch


CASES = {
    # name: (model kwargs, B, multi_hot, n_steps, alpha, frozen)
    "nb_basic": (dict(input_dim=96, hidden_layer_sizes=[32, 16], latent_dim=8, n_cond=2, n_pred=5,
                      recon_loss="nb", dr_rate=0.0, use_mmd=False), 24, False, 2, 1e-3, False),
    "nb_mmd_z_dropout": (dict(input_dim=120, hidden_layer_sizes=[48, 24], latent_dim=10, n_cond=3, n_pred=4,
                              recon_loss="nb", dr_rate=0.05, use_mmd=True, mmd_on="z", beta=500.0,
                              classifier_hidden_layer_sizes=[32, 64]), 40, False, 2, 5e-4, False),
    "mse_mmd_c5": (dict(input_dim=100, hidden_layer_sizes=[32, 32], latent_dim=10, n_cond=5, n_pred=0,
                        recon_loss="mse", dr_rate=0.05, use_mmd=True, mmd_on="z", beta=1.0), 48, False, 2, 1e-3,
                   False),
    "nb_mmd_y_main": (dict(input_dim=80, hidden_layer_sizes=[40, 24, 16], latent_dim=6, n_cond=3, n_pred=3,
                           recon_loss="nb", dr_rate=0.0, use_mmd=True, mmd_on="y", beta=10.0, main_dataset=1,
                           use_own_kl=True), 36, False, 1, 1e-3, False),
    "nb_mmd_boundary_multihot": (dict(input_dim=64, hidden_layer_sizes=[24, 12], latent_dim=5, n_cond=4, n_pred=2,
                                      recon_loss="nb", dr_rate=0.0, use_mmd=True, mmd_on="z", beta=3.0,
                                      mmd_boundary=2, applylog=False), 64, True, 1, 1e-3, False),
    "nb_surgery_frozen": (dict(input_dim=72, hidden_layer_sizes=[32, 16], latent_dim=8, n_cond=3, n_pred=4,
                               recon_loss="nb", dr_rate=0.0, use_mmd=True, mmd_on="z", beta=100.0), 30, False, 2,
                          1e-3, True),
    "nb_wide_row": (dict(input_dim=19000, hidden_layer_sizes=[16, 8], latent_dim=4, n_cond=2, n_pred=0,
                         recon_loss="nb", dr_rate=0.0, use_mmd=False), 4, False, 1, 1e-3, False),
    "zinb_mmd_z": (dict(input_dim=88, hidden_layer_sizes=[40, 16], latent_dim=7, n_cond=3, n_pred=3,
                        recon_loss="zinb", dr_rate=0.0, use_mmd=True, mmd_on="z", beta=20.0), 32, False, 2, 1e-3,
                   False),
    # BatchNorm1d variant (modules.py:298-302 etc.): affine parameters, batch statistics, running-stat buffers
    "nb_bn_mmd_dropout": (dict(input_dim=104, hidden_layer_sizes=[40, 24], latent_dim=9, n_cond=3, n_pred=4,
                               recon_loss="nb", dr_rate=0.05, use_mmd=True, mmd_on="y", beta=50.0, use_bn=True,
                               classifier_hidden_layer_sizes=[32, 16]), 40, False, 2, 1e-3, False),
    # surgery freeze with BatchNorm: the trainer's per-iteration rule (trainer.py:305-312) is applied by the harness
    "nb_bn_surgery_frozen": (dict(input_dim=72, hidden_layer_sizes=[32, 16], latent_dim=8, n_cond=3, n_pred=4,
                                  recon_loss="nb", dr_rate=0.0, use_mmd=True, mmd_on="z", beta=100.0, use_bn=True),
                             30, False, 2, 1e-3, True),
}


def run_case(ns, name, spec, seed):
    import torch
    import torch.nn.functional as F

    kw, B, multi_hot, n_steps, alpha, frozen = spec
    kw = dict(kw)
    C, P = kw.pop("n_cond"), kw.pop("n_pred")
    conditions = [f"c{i}" for i in range(C)]
    predictors = [f"p{i}" for i in range(P)]
    main = kw.pop("main_dataset", None)
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    model = ns.Celligner2(conditions=conditions, predictors=predictors,
                          main_dataset=(conditions[main] if main is not None else None), **kw)
    model.train()
    # weights a little away from init so that every term is exercised
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if n_.endswith("bias"):
                p_.add_(0.05 * torch.randn_like(p_))
        for m_ in model.modules():  # BatchNorm: non-trivial affine parameters and running statistics
            if isinstance(m_, torch.nn.BatchNorm1d):
                m_.weight.add_(0.2 * torch.randn_like(m_.weight))
                m_.running_mean.add_(0.1 * torch.randn_like(m_.running_mean))
                m_.running_var.mul_(1 + 0.3 * torch.rand_like(m_.running_var))
    if frozen:  # othermodels/base/_base.py:342-354 (freeze_expression=True)
        model.freeze = True
        for n_, p_ in model.named_parameters():
            p_.requires_grad = ("theta" in n_) or ("cond_L.weight" in n_)
    batch = make_batch(rng, B, kw["input_dim"], C, P, multi_hot=multi_hot, mse=kw["recon_loss"] == "mse")

    rec = {"masks": [], "eps": []}
    orig_dropout = F.dropout
    from torch.distributions import normal as tnormal

    orig_sn = tnormal._standard_normal

    def rec_dropout(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return inp
        m = torch.bernoulli(torch.full_like(inp, 1 - p))
        rec["masks"].append(m.numpy().astype(np.uint8))
        return inp * m / (1 - p)

    def rec_sn(shape, dtype, device):
        e = torch.randn(shape, dtype=dtype)
        rec["eps"].append(e.numpy().copy())
        return e

    class _MaskedSum:
        def __init__(self, t):
            self.t = t

        def sum(self, dim=-1, where=None):
            return (self.t * where).sum(dim=dim)

    orig_mse = ns.model.mse
    F.dropout = rec_dropout
    tnormal._standard_normal = rec_sn
    ns.model.mse = lambda r, x: _MaskedSum(orig_mse(r, x))
    try:
        params0 = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        opt = torch.optim.Adam([p_ for p_ in model.parameters() if p_.requires_grad],
                               lr=1e-3, eps=0.01, weight_decay=0.04)
        arrays = {}
        for k, v in params0.items():
            arrays["p0/" + k] = v
        for k, v in batch.items():
            arrays["batch/" + k] = v
        dnames = []
        if kw["dr_rate"] > 0:  # RNG order: SURVEY.md Appendix A.9
            nh = len(kw["hidden_layer_sizes"])
            dnames = [f"enc.D{i}" for i in range(nh)] + [f"dec.D{i}" for i in range(nh)]
            if P > 0:
                dnames += [f"cls.D{i}" for i in range(len(kw.get("classifier_hidden_layer_sizes", [32])))]
        hooked = {}
        if name == "nb_wide_row":  # single-op vector: keep only the NB loss' inputs/outputs
            def _hook(mod, inp, out):
                out.retain_grad()
                hooked["o"] = out
            model.decoder.mean_decoder[0].register_forward_hook(_hook)
        for step in range(n_steps):
            rec["masks"].clear()
            rec["eps"].clear()
            tb = {k: torch.tensor(v) for k, v in batch.items()}  # fresh `classes` each step
            if model.freeze:  # what Trainer.on_iteration does before every loss (trainer.py:305-312)
                for m_ in model.modules():
                    if isinstance(m_, torch.nn.BatchNorm1d) and not m_.weight.requires_grad:
                        m_.affine = False
                        m_.track_running_stats = False
            recon, kl, mmd, ce = model(**tb)
            loss = recon + alpha * kl + mmd + ce
            opt.zero_grad()
            loss.backward()
            opt.step()
            arrays[f"s{step}/losses"] = np.array([recon.item(), kl.item(), mmd.item(), ce.item()], np.float64)
            arrays[f"s{step}/eps"] = rec["eps"][0]
            assert len(rec["eps"]) == 1
            assert len(rec["masks"]) == len(dnames), (len(rec["masks"]), dnames)
            for dn, m in zip(dnames, rec["masks"]):
                arrays[f"s{step}/mask/{dn}"] = m
            for n_, p_ in model.named_parameters():
                if p_.grad is not None:
                    arrays[f"s{step}/grad/{n_}"] = p_.grad.numpy().copy()
                arrays[f"s{step}/param/{n_}"] = p_.detach().numpy().copy()
            for n_, b_ in model.named_buffers():
                arrays[f"s{step}/buffer/{n_}"] = b_.detach().numpy().copy()
            if "classes" in batch:
                arrays[f"s{step}/classes_after"] = tb["classes"].numpy().copy()
        if name == "nb_wide_row":
            arrays = {
                "op/logits": hooked["o"].detach().numpy().copy(),
                "op/dlogits": hooked["o"].grad.numpy().copy(),
                "op/x": batch["x"], "op/goodloc": batch["goodloc"], "op/sizefactor": batch["sizefactor"],
                "op/batch": batch["batch"], "op/theta": params0["theta"],
                "op/dtheta": arrays["s0/grad/theta"], "op/recon": arrays["s0/losses"][:1],
                "s0/losses": arrays["s0/losses"],
            }
        meta = dict(
            case=name, seed=seed, n_steps=n_steps, alpha=alpha, frozen=frozen,
            model_kwargs=kw, n_conditions=C, n_predictors=P, main_dataset=main,
            adam=dict(lr=1e-3, eps=0.01, weight_decay=0.04, betas=[0.9, 0.999]),
            trainable=[n_ for n_, p_ in model.named_parameters() if p_.requires_grad],
            bn_untracked=[n_ for n_, m_ in model.named_modules()
                          if isinstance(m_, torch.nn.BatchNorm1d) and not m_.track_running_stats],
            torch_version=torch.__version__,
            patches=["recorded dropout/eps draws", "mse masked-sum wrapper (SURVEY Appendix B.2)"],
        )
        arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
        os.makedirs(OUT, exist_ok=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
        print(name, "losses", arrays["s0/losses"],
              "size KB", os.path.getsize(os.path.join(OUT, name + ".npz")) // 1024)
    finally:
        F.dropout = orig_dropout
        tnormal._standard_normal = orig_sn
        ns.model.mse = orig_mse


def main():
    ns = ref_harness.load()
    only = sys.argv[1:]
    for i, (name, spec) in enumerate(CASES.items()):
        if only and name not in only:
            continue
        run_case(ns, name, spec, seed=100 + i)


if __name__ == "__main__":
    main()
