"""Golden fixture of the reference's CHECKPOINT DIRECTORY, written by the reference's own `BaseMixin.save`
(celligner2/othermodels/base/_base.py:47-95) in the build container: model_params.pt, attr.pkl, var_names.csv.

A small reference wrapper `CELLIGNER2` (model/celligner2_model.py:98-312) is built on an AnnData-like object and saved.
The fixture holds the three files' BYTES (data the reference wrote: a state_dict, a dict of plain attributes, a text
table), plus, for reading without unpickling anything, the attribute dict as JSON (key -> [type name, JSON value]), the
state_dict's keys / shapes / dtypes and the gene names.  Nothing of the reference's source ships.
tests/test_surgery_cpu.py rebuilds the directory from the bytes, loads it with celligner2_amd.surgery (restricted
unpickler, torch.load(weights_only=True)) and checks that `surgery.save` writes the same key set and layout.

Harness patches: `pdb.set_trace` no-op (_base.py:32-34), numpy-1.x private import stand-in, and
the AnnData stand-in gets `var_names` (a pandas Index, as AnnData's).

Usage: python tools/gen_golden_checkpoint.py        (needs /root/reference)
"""
import io
import json
import os
import sys
import tempfile

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402
from gen_golden_trainer import make_obs_matrix  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    import torch
    ns = ref_harness.load_wrapper()
    rng = np.random.default_rng(33)
    G = 40
    lineages = ["lung", "skin", "blood"]
    X, obs = make_obs_matrix(rng, 90, G, ["ccle", "tcga"], lineages, ["c"], nan_frac=0.0, miss_frac=0.15)
    adata = ns.AnnDataLike(X.copy(), obs.copy())
    adata.var_names = pd.Index([f"ENSG{i:05d}" for i in range(G)])
    torch.manual_seed(8)
    ref = ns.CELLIGNER2(adata, hidden_layer_sizes=[20, 10], latent_dim=5, dr_rate=0.05, condition_keys=["dataset"],
                        conditions=["ccle", "tcga"], predictor_keys=["lineage"], predictors=list(lineages), use_mmd=True,
                        mmd_on="z", beta=7.0, recon_loss="nb", classifier_hidden_layer_sizes=[12])
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "ck")
        ref.save(d, overwrite=False)
        files = sorted(os.listdir(d))
        raw = {f: open(os.path.join(d, f), "rb").read() for f in files}
    # read back what the reference wrote with loaders that execute nothing from the files
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from celligner2_amd.surgery import _PlainUnpickler
    attrs = _PlainUnpickler(io.BytesIO(raw["attr.pkl"])).load()
    state = torch.load(io.BytesIO(raw["model_params.pt"]), map_location="cpu", weights_only=True)

    def js(v):
        if isinstance(v, (np.generic,)):
            v = v.item()
        if isinstance(v, (list, tuple)):
            return [js(x) for x in v]
        if isinstance(v, dict):
            return {str(k): js(x) for k, x in v.items()}
        if v is None or isinstance(v, (bool, int, float, str)):
            return v
        return repr(v)

    meta = dict(case="checkpoint_dir", files=files,
                attrs={k: [type(v).__name__, js(v)] for k, v in attrs.items()},
                state=[[k, list(v.shape), str(v.dtype)] for k, v in state.items()],
                var_names=[str(v) for v in adata.var_names], torch_version=torch.__version__,
                patches=["pdb.set_trace no-op", "numpy.lib.function_base stand-in", "AnnData stand-in with var_names"])
    arrays = {"file/" + f: np.frombuffer(b, dtype=np.uint8) for f, b in raw.items()}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, "checkpoint_dir.npz")
    np.savez_compressed(path, **arrays)
    print("checkpoint_dir:", files, len(attrs), "attributes,", len(state), "state tensors, KB", os.path.getsize(path) // 1024)
    print(sorted(attrs))


if __name__ == "__main__":
    main()
