"""`celligner2_amd.wrapper.CELLIGNER2` against the reference's own wrapper on the inference path
(celligner2/model/celligner2_model.py:392-520): tests/golden/wrapper_inference.npz was written by the reference's
`CELLIGNER2.get_latent(mean=True, add_classpred=True, get_fullpred=True)` and `CELLIGNER2.reconstruct` on 1100 rows
(tools/gen_golden_wrapper.py) -- three chunks of the 512-row loop, the last ragged."""
import numpy as np
import pandas as pd
import pytest
import torch

from tests.test_wrapper_cpu import fixture
from celligner2_amd.wrapper import CELLIGNER2, LatentTable

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 1e-2)])
def test_get_latent_and_reconstruct_match_the_reference_wrapper(precision, tol):
    z, meta, obs = fixture()
    adata = LatentTable(np.array(z["X"]), obs.copy())
    adata.var_names = pd.Index(meta["expr_columns"])
    w = CELLIGNER2(adata, precision=precision, **meta["kwargs"])
    w.model.load_state_dict({k[2:]: torch.tensor(z[k]) for k in z.files if k.startswith("p/")})
    res, predictions = w.get_latent(mean=True, add_classpred=True, get_fullpred=True)
    assert res.X.shape == (1100, 6) and list(res.obs.columns) == meta["res_obs_columns"]
    assert list(res.obs.index) == meta["obs_index"]
    assert rel(res.X, z["latent_mean"]) < tol, rel(res.X, z["latent_mean"])
    assert rel(predictions, z["predictions"]) < tol
    # predicted labels: the reference's, except where its two best scores are closer than the tolerance
    ref_scores = np.sort(z["predictions"], axis=1)
    decided = (ref_scores[:, -1] - ref_scores[:, -2]) > 2 * tol * np.abs(z["predictions"]).max()
    got = np.array(res.obs["lineage_pred"], dtype=object)
    want = np.array(meta["lineage_pred"], dtype=object)
    assert decided.sum() > 800 and (got[decided] == want[decided]).all()
    assert (np.array(res.obs["dataset"]) == np.array(meta["obs"]["dataset"])).all()
    # without the class scores: no prediction column, second value None
    res2, none = w.get_latent(mean=True)
    assert none is None and list(res2.obs.columns) == ["dataset", "lineage"] and np.array_equal(res2.X, res.X)
    # a sampled latent differs from the mean and between calls (eps is drawn per call, as in the reference)
    s1, _ = w.get_latent()
    s2, _ = w.get_latent()
    assert not np.allclose(s1.X, res.X) and not np.allclose(s1.X, s2.X)
    # the decoder on the reference's latent means, first 600 rows (two chunks)
    expr = w.reconstruct(z["latent_mean"][:600], obs.iloc[:600][["dataset"]])
    assert list(expr.columns) == meta["expr_columns"] and expr.shape == (600, 48)
    assert rel(expr.to_numpy(), z["expression600"]) < tol
    np.testing.assert_allclose(expr.to_numpy().sum(1), 1.0, rtol=1e-4)  # NB: softmax means


def test_wrapper_trains_and_surgery_fine_tunes():
    """`CELLIGNER2.train` / `load_query_data` end to end on the GPU: losses finite, only the surgery's parameters move."""
    z, meta, obs = fixture()
    adata = LatentTable(np.array(z["X"])[:256], obs.iloc[:256].copy())
    adata.var_names = pd.Index(meta["expr_columns"])
    torch.manual_seed(0)
    np.random.seed(0)
    w = CELLIGNER2(adata, **meta["kwargs"])
    w.train(n_epochs=2, lr=1e-3, batch_size=64, train_frac=0.8, monitor=False, use_leiden=False)
    assert w.is_trained_ and np.isfinite(w.trainer.logs["epoch_loss"]).all() and len(w.trainer.logs["epoch_loss"]) == 2
    q_obs = obs.iloc[300:428].copy()
    q_obs["dataset"] = "pdx"
    q = LatentTable(np.array(z["X"])[300:428], q_obs)
    new = CELLIGNER2.load_query_data(q, w)
    before = {n: p.detach().clone() for n, p in new.model.named_parameters()}
    new.train(n_epochs=1, lr=1e-2, batch_size=64, train_frac=0.8, monitor=False, use_leiden=False)
    moved = sorted(n for n, p in new.model.named_parameters() if not torch.equal(p.detach(), before[n]))
    assert moved == ["decoder.FirstL.L0.cond_L.weight", "encoder.FC.L0.cond_L.weight", "theta"], moved
    lat, _ = new.get_latent(mean=True)
    assert lat.X.shape == (128, 6) and np.isfinite(lat.X).all()
