"""Trainer mirror on the GPU: the reference's loop semantics (logs keys, KL annealing, early stopping / LR decay
hooks, validation, device-resident batches) on a small synthetic AnnData-like object."""
import numpy as np
import pandas as pd
import pytest
import torch

pytestmark = pytest.mark.gpu


class AnnDataLike:
    def __init__(self, X, obs):
        self.X, self.obs = X, obs

    @property
    def shape(self):
        return self.X.shape

    def __len__(self):
        return self.X.shape[0]

    def __getitem__(self, idx):
        rows = np.asarray(idx[0] if isinstance(idx, tuple) else idx)
        return AnnDataLike(self.X[rows], self.obs.iloc[rows].copy())


def make_adata(n=300, G=64, seed=0):
    rng = np.random.default_rng(seed)
    ds = rng.choice(["ccle", "tcga", "gtex"], n)
    lin = rng.choice(["lung", "skin", "blood", "unknown"], n)
    shift = {"ccle": 0.0, "tcga": 0.5, "gtex": -0.5}
    gene = np.exp(rng.normal(1.0, 1.0, (1, G)))
    X = rng.poisson(gene * np.exp(np.array([shift[d] for d in ds]))[:, None]).astype(np.float32)
    X[rng.uniform(size=X.shape) < 0.01] = np.nan
    return AnnDataLike(X, pd.DataFrame({"dataset": ds, "lineage": lin}))


def test_train_two_epochs_logs_and_decrease():
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer
    np.random.seed(0)
    adata = make_adata()
    torch.manual_seed(1)
    model = Celligner2(64, [32, 16], 8, 0.05, False, "nb", conditions=["ccle", "gtex", "tcga"], use_mmd=True,
                       beta=5.0, predictors=["lung", "skin", "blood"])  # "unknown" is not a predictor -> -1 rows
    tr = Celligner2Trainer(model, adata, condition_keys=["dataset"], predictor_keys=["lineage"], use_leiden=False,
                           batch_size=64, alpha_epoch_anneal=2, monitor=False, train_frac=0.8, seed=2020)
    tr.train(n_epochs=6, lr=1e-2, eps=0.01)
    for k in ("epoch_loss", "epoch_unweighted_loss", "epoch_recon_loss", "epoch_kl_loss", "epoch_class_ce_loss",
              "epoch_mmd_loss", "val_loss", "val_unweighted_loss", "val_recon_loss"):
        assert len(tr.logs[k]) == 6 and np.isfinite(tr.logs[k]).all(), k
    assert tr.logs["epoch_recon_loss"][-1] < tr.logs["epoch_recon_loss"][0]
    assert tr.iters_per_epoch == int(np.ceil(len(tr.train_data) / 64))
    assert not model.training and tr.training_time > 0
    # labels missing from the encoder are -1 in the dataset (dataset/celligner2/_utils.py:69-76)
    assert (tr.train_data.predictors_np == -1).any()
    # goodloc / imputation (trainers/celligner2/_utils.py:143-147)
    assert not np.isnan(adata.X).any() and (~tr.train_data.goodloc.cpu().numpy()).any()


def test_log_sync_step_matches_epoch_sync():
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer
    logs = []
    for mode in ("step", "epoch"):
        np.random.seed(0)
        adata = make_adata(120, 32, 1)
        torch.manual_seed(1)
        model = Celligner2(32, [16, 8], 4, 0.0, False, "nb", conditions=["ccle", "gtex", "tcga"], use_mmd=False,
                           predictors=["lung", "skin", "blood", "unknown"])
        tr = Celligner2Trainer(model, adata, condition_keys=["dataset"], predictor_keys=["lineage"], use_leiden=False,
                               batch_size=32, monitor=False, train_frac=0.75, seed=7, log_sync=mode,
                               use_early_stopping=False)
        tr.train(n_epochs=2, lr=1e-3)
        logs.append(dict(tr.logs))
    for k in logs[0]:
        assert np.allclose(logs[0][k], logs[1][k], rtol=1e-5), k
