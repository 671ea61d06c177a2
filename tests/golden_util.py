"""Helpers to read tests/golden/*.npz (fixtures made by tools/gen_golden.py)."""
import glob
import json
import os

import numpy as np

from oracle import c2_oracle as orc

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def step_cases():
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        name = os.path.basename(f)[:-4]
        if name.startswith(("nb_wide_row", "trainer_", "surgery_", "checkpoint_", "wrapper_")):
            continue
        out.append(name)
    return out


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def cfg_from_meta(meta):
    kw = dict(meta["model_kwargs"])
    return orc.Cfg(
        input_dim=kw["input_dim"], hidden_layer_sizes=kw["hidden_layer_sizes"], latent_dim=kw["latent_dim"],
        n_conditions=meta["n_conditions"], n_predictors=meta["n_predictors"],
        classifier_hidden_layer_sizes=kw.get("classifier_hidden_layer_sizes", [32]),
        recon_loss=kw["recon_loss"], dr_rate=kw["dr_rate"], use_own_kl=kw.get("use_own_kl", False),
        applylog=kw.get("applylog", True), use_mmd=kw.get("use_mmd", False), mmd_on=kw.get("mmd_on", "z"),
        mmd_boundary=kw.get("mmd_boundary"), beta=kw.get("beta", 1.0), main_dataset=meta.get("main_dataset"),
        betaclass=kw.get("betaclass", 0.8), use_ln=kw.get("use_ln", True), use_bn=kw.get("use_bn", False))


def split_state(z, cfg):
    """The fixture's initial state_dict as (parameters, BatchNorm buffers)."""
    sd = group(z, "p0/")
    bnames = set(orc.buffer_shapes(cfg))
    return {k: v for k, v in sd.items() if k not in bnames}, {k: v for k, v in sd.items() if k in bnames}


def group(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    return float(np.abs(a - b).max() / den) if b.size else 0.0


def grad_err(name, got, ref, cfg, all_ref):
    """Max-norm relative error of one gradient tensor.  With BatchNorm the batch sum of every gradient that flows
    back through a norm layer is analytically ZERO (the batch mean is subtracted again), so bias gradients are
    exact zeros (a Linear feeding a norm layer) or the small remainder of a cancelling sum (the encoder heads);
    the reference's fp32 result is rounding noise at the scale of the summed terms there, and the error is taken
    relative to the larger of the tensor itself and the same module's weight gradient."""
    if getattr(cfg, "use_bn", False) and name.endswith(".bias") and name[:-4] + "weight" in all_ref:
        scale = max(np.abs(all_ref[name[:-4] + "weight"]).max(), np.abs(ref).max())
        return float(np.abs(np.asarray(got, np.float64) - ref).max() / scale)
    return relerr(got, ref)


class AnnDataLike:
    """Minimal AnnData stand-in (what the reference's trainer touches): .X ndarray, .obs DataFrame, row slicing."""

    def __init__(self, X, obs):
        self.X, self.obs = X, obs

    @property
    def shape(self):
        return self.X.shape

    @property
    def n_vars(self):
        return self.X.shape[1]

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, idx):
        rows = np.asarray(idx[0] if isinstance(idx, tuple) else idx)
        return AnnDataLike(self.X[rows], self.obs.iloc[rows].copy())


def adata_from(z, x_key="X", obs_prefix="obs/"):
    """The AnnData-like object a trainer_* / surgery_* fixture was generated from (a fresh copy)."""
    import pandas as pd
    cols = {k[len(obs_prefix):]: [str(v) for v in z[k]] for k in z.files if k.startswith(obs_prefix)}
    n = len(next(iter(cols.values())))
    return AnnDataLike(z[x_key].copy(), pd.DataFrame(cols, index=[f"s{i}" for i in range(n)]))


def calls_of(z, prefix="calls/"):
    """Per loss() call of the reference trainer: (train flag, rows, eps, lr, alpha, epoch)."""
    size = z[prefix + "size"]
    ends = np.cumsum(size)
    out = []
    for i, (e, n) in enumerate(zip(ends, size)):
        out.append(dict(train=int(z[prefix + "train"][i]), rows=z[prefix + "rows"][e - n:e], eps=z[prefix + "eps"][e - n:e],
                        lr=float(z[prefix + "lr"][i]), alpha=float(z[prefix + "alpha"][i]), epoch=int(z[prefix + "epoch"][i])))
    return out
