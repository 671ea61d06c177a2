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
        if name.startswith("nb_wide_row") or name.startswith("trainer_"):
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
        betaclass=kw.get("betaclass", 0.8))


def group(z, prefix):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def relerr(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    return float(np.abs(a - b).max() / den) if b.size else 0.0
