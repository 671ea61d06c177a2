"""Host logic of the model wrapper `celligner2_amd.wrapper.CELLIGNER2` (the reference's `CELLIGNER2`,
celligner2/model/celligner2_model.py:97-593 + othermodels/base/_base.py:47-356) -- labels from obs, the public attribute
set, the checkpoint directory, query-data surgery -- on the CPU: nothing here computes (the module's kernels are covered
by tests/test_wrapper_gpu.py against the fixture the reference's own wrapper wrote, tools/gen_golden_wrapper.py)."""
import json
import os

import numpy as np
import pandas as pd
import pytest
import torch

from celligner2_amd.wrapper import CELLIGNER2, LatentTable

HERE = os.path.dirname(os.path.abspath(__file__))


def fixture():
    z = np.load(os.path.join(HERE, "golden", "wrapper_inference.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    obs = pd.DataFrame({k: np.array(v, dtype=object) for k, v in meta["obs"].items()}, index=meta["obs_index"])
    return z, meta, obs


def build(z, meta, obs, **over):
    kw = dict(meta["kwargs"])
    kw.update(over)
    adata = LatentTable(np.array(z["X"]), obs.copy())
    adata.var_names = pd.Index(meta["expr_columns"])
    return CELLIGNER2(adata, device="cpu", **kw)


def test_public_attributes_are_the_reference_wrappers():
    z, meta, obs = fixture()
    w = build(z, meta, obs)
    assert sorted(w._get_public_attributes()) == meta["attr_keys"]
    assert w.conditions_ == ["ccle", "tcga", "gtex"] and w.condition_set_ == {"dataset": {"ccle", "tcga", "gtex"}}
    assert w.predictor_set_ == {"lineage": {"lung", "skin", "blood", "bone"}}  # the missing marker "U" is not a label
    assert w.input_dim_ == 48 and w.model.n_conditions == 3 and w.model.n_predictors == 4 and w.is_trained_ is False
    sd = w.model.state_dict()
    assert sorted(sd) == sorted(k[2:] for k in z.files if k.startswith("p/"))
    assert all(tuple(sd[k].shape) == z["p/" + k].shape for k in sd)


def test_labels_from_obs_sorted_and_missing_condition_refused():
    z, meta, obs = fixture()
    w = build(z, meta, obs, conditions=None, predictors=None)
    assert w.conditions_ == sorted(["ccle", "tcga", "gtex"]) and w.predictors_ == sorted(["lung", "skin", "blood", "bone"])
    bad = obs.copy()
    bad.loc[bad.index[3], "dataset"] = "U"
    with pytest.raises(ValueError, match="missing values"):
        build(z, meta, bad, conditions=None)
    with pytest.raises(NotImplementedError):
        build(z, meta, obs, n_expand=2)


def test_checkpoint_directory_round_trip_and_query_surgery(tmp_path):
    z, meta, obs = fixture()
    w = build(z, meta, obs)
    w.model.load_state_dict({k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("p/")})
    d = str(tmp_path / "ck")
    w.save(d)
    assert sorted(os.listdir(d)) == ["attr.pkl", "model_params.pt", "var_names.csv"]
    with pytest.raises(ValueError, match="already exists"):
        w.save(d)
    back = CELLIGNER2.load(d, w.adata, device="cpu")
    assert sorted(back._get_public_attributes()) == meta["attr_keys"] and back.conditions_ == w.conditions_
    assert all(torch.equal(a, b) for a, b in zip(back.model.state_dict().values(), w.model.state_dict().values()))
    assert not back.model.training
    # query data with a dataset the reference model has not seen: one more condition column, weights in the leading block
    q_obs = obs.iloc[:50].copy()
    q_obs["dataset"] = np.array(["pdx"] * 25 + ["tcga"] * 25, dtype=object)
    q = LatentTable(np.array(z["X"])[:50], q_obs)
    new = CELLIGNER2.load_query_data(q, d, device="cpu")
    assert new.conditions_ == ["ccle", "tcga", "gtex", "pdx"] and new.dr_rate_ == 0.0 and new.model.freeze
    assert tuple(new.model.theta.shape) == (48, 4)
    assert torch.equal(new.model.theta[:, :3], w.model.theta) and not torch.equal(new.model.theta[:, 3], w.model.theta[:, 0])
    trainable = sorted(n for n, p in new.model.named_parameters() if p.requires_grad)
    assert trainable == ["decoder.FirstL.L0.cond_L.weight", "encoder.FC.L0.cond_L.weight", "theta"]
    same = CELLIGNER2.load_query_data(LatentTable(np.array(z["X"])[:50], obs.iloc[:50].copy()), w, freeze=False, device="cpu")
    assert same.conditions_ == w.conditions_ and all(p.requires_grad for p in same.model.parameters())
    q_obs2 = obs.iloc[:50].copy()
    q_obs2["lineage"] = "brain"
    with pytest.raises(ValueError, match="new predictions"):
        CELLIGNER2.load_query_data(LatentTable(np.array(z["X"])[:50], q_obs2), d, device="cpu")
