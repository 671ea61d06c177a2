"""Golden fixture of the reference's WRAPPER class on the inference path: `CELLIGNER2.get_latent` (the 512-row chunk loop,
celligner2/model/celligner2_model.py:392-482) and `CELLIGNER2.reconstruct` (:484-520), run by the reference itself in the
build container on an AnnData-like object of 1100 rows (three chunks, the last ragged).

The fixture holds inputs and outputs only: the expression matrix and obs columns, the reference module's state_dict,
the latent means, the class scores and predicted labels, the reconstructed expression of the first 600 rows, and the
wrapper's public attribute KEYS.  tests/test_wrapper_gpu.py loads the weights into celligner2_amd.wrapper.CELLIGNER2 and
demands the same numbers (fp32, 1e-4) and the same labels.

Harness patches: those of tools/gen_golden_checkpoint.py (pdb no-op, numpy-1.x stand-in, AnnData stand-in).
Usage: python tools/gen_golden_wrapper.py        (needs /root/reference)
"""
import json
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402
from gen_golden_trainer import make_obs_matrix  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    import torch
    ns = ref_harness.load_wrapper()
    rng = np.random.default_rng(77)
    N, G = 1100, 48
    datasets, lineages = ["ccle", "tcga", "gtex"], ["lung", "skin", "blood", "bone"]
    X, obs = make_obs_matrix(rng, N, G, datasets, lineages, ["c"], nan_frac=0.0, miss_frac=0.2)
    adata = ns.AnnDataLike(X.copy(), obs.copy())
    adata.var_names = pd.Index([f"ENSG{i:05d}" for i in range(G)])
    adata.var = pd.DataFrame(index=adata.var_names)
    torch.manual_seed(21)
    kw = dict(hidden_layer_sizes=[24, 12], latent_dim=6, dr_rate=0.05, condition_keys=["dataset"],
              conditions=list(datasets), predictor_keys=["lineage"], predictors=list(lineages), use_mmd=True, mmd_on="z",
              beta=3.0, recon_loss="nb", classifier_hidden_layer_sizes=[10])
    ref = ns.CELLIGNER2(adata, **kw)
    with torch.no_grad():  # weights a little away from init
        for n_, p_ in ref.model.named_parameters():
            if n_.endswith("bias"):
                p_.add_(0.05 * torch.randn_like(p_))
    ref.model.eval()
    res, predictions = ref.get_latent(mean=True, add_classpred=True, get_fullpred=True)
    sub = ns.AnnDataLike(X[:600].copy(), obs.iloc[:600].copy())
    expr = ref.reconstruct(np.asarray(res.X)[:600], sub.obs[["dataset"]])
    arrays = {"X": X, "latent_mean": np.asarray(res.X), "predictions": np.asarray(predictions),
              "expression600": expr.to_numpy()}
    for k, v in ref.model.state_dict().items():
        arrays["p/" + k] = v.detach().numpy().copy()
    meta = dict(case="wrapper_inference", kwargs=kw, obs={c: [str(v) for v in obs[c]] for c in obs.columns},
                obs_index=[str(i) for i in obs.index], res_obs_columns=[str(c) for c in res.obs.columns],
                lineage_pred=[str(v) for v in res.obs["lineage_pred"]], expr_columns=[str(c) for c in expr.columns],
                attr_keys=sorted(ref._get_public_attributes()), torch_version=torch.__version__,
                patches=["pdb.set_trace no-op", "numpy.lib.function_base stand-in", "AnnData stand-in"])
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, "wrapper_inference.npz")
    np.savez_compressed(path, **arrays)
    print("wrapper_inference:", arrays["latent_mean"].shape, arrays["predictions"].shape, arrays["expression600"].shape,
          "res.obs columns", meta["res_obs_columns"], "KB", os.path.getsize(path) // 1024)
    print(meta["attr_keys"])


if __name__ == "__main__":
    main()
