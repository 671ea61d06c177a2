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


# ---------------------------------------------------------------------------------------------------------------------
# The loop pinned to the reference: fixtures of tools/gen_golden_trainer.py (the reference's Celligner2Trainer run in the
# build container).  Same rows visited by every loss() call (train and validation), same eps draws, same learning rate
# and KL weight per call, same `logs`, same weights at the end -- fp32 mode, 1e-4.
def _spy(trainer, model):
    """Record, per loss() call, the rows the loader gathered and the eps the model drew."""
    seen = dict(rows=[], eps=[], calls=[])
    for ds in (trainer.train_data, trainer.valid_data):
        if ds is None:
            continue
        orig = ds.gather

        def gather(idx, *a, _orig=orig, **kw):
            seen["rows"].append(idx.cpu().numpy().copy())
            return _orig(idx, *a, **kw)

        ds.gather = gather
    draw = model._draw_noise

    def draw_noise(B, with_classes, device):
        eps, masks = draw(B, with_classes, device)
        seen["eps"].append(eps.cpu().numpy().copy())
        return eps, masks

    model._draw_noise = draw_noise
    base_loss = trainer.loss

    def loss(total_batch=None):
        out = base_loss(total_batch)
        seen["calls"].append(dict(train=int(model.training), lr=float(trainer.optimizer.param_groups[0]["lr"]),
                                  alpha=float(trainer.calc_alpha_coeff()), epoch=int(trainer.epoch)))
        return out

    trainer.loss = loss
    return seen


def _compare_calls(seen, ref_calls):
    assert len(seen["calls"]) == len(ref_calls) == len(seen["rows"]) == len(seen["eps"])
    for i, (c, r) in enumerate(zip(seen["calls"], ref_calls)):
        assert c["train"] == r["train"] and c["epoch"] == r["epoch"], i
        assert np.array_equal(seen["rows"][i], r["rows"]), f"rows of loss() call {i} differ from the reference's"
        assert np.array_equal(seen["eps"][i], r["eps"]), f"eps of loss() call {i}"
        assert abs(c["lr"] - r["lr"]) <= 1e-15 and abs(c["alpha"] - r["alpha"]) <= 1e-15, i


@pytest.mark.parametrize("name", ["trainer_nb_stratified_earlystop", "trainer_nb_shuffled_dropout_clip"])
def test_trainer_loop_matches_the_reference(name):
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer
    from tests import golden_util as gu
    z, meta = gu.load(name)
    adata = gu.adata_from(z)
    model = Celligner2(adata.X.shape[1], conditions=list(meta["conditions"]), predictors=list(meta["predictors"]),
                       **meta["model_kwargs"])
    model.load_state_dict({k: torch.tensor(v) for k, v in gu.group(z, "p0/").items()})
    np.random.seed(meta["np_seed"])
    tr = Celligner2Trainer(model, adata, rng="reference", log_sync="step", **meta["trainer_kwargs"])
    seen = _spy(tr, model)
    tr.train(**meta["train_kwargs"])
    _compare_calls(seen, gu.calls_of(z))
    assert tr.epoch + 1 == meta["epochs_run"] and tr.best_epoch == meta["best_epoch"]
    assert abs(tr.optimizer.param_groups[0]["lr"] - meta["lr_final"]) < 1e-15
    assert tr.iters_per_epoch == meta["iters_per_epoch"]
    ref_logs = gu.group(z, "logs/")
    assert set(tr.logs) == set(ref_logs)
    for k, v in ref_logs.items():
        got = np.asarray(tr.logs[k], np.float64)
        assert got.shape == v.shape, k
        assert np.all(np.abs(got - v) <= 1e-4 * np.maximum(np.abs(v), 1.0)), (k, got, v)
    for k, v in gu.group(z, "final/").items():
        assert gu.relerr(model.state_dict()[k].cpu().numpy(), v) < 1e-4, k
    assert not model.training


def test_surgery_fine_tune_matches_the_reference():
    """The reference's `load_query_data` + `.train` on the query data (fixture surgery_c2_to_c3): expanded model built by
    surgery.load_query_model, fine-tuned by this package's trainer -> same rows, draws, logs; the trainable tensors end
    where the reference's end (1e-4), every frozen tensor is bit-identical to the expanded checkpoint."""
    from celligner2_amd import surgery
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer
    from tests import golden_util as gu
    z, meta = gu.load("surgery_c2_to_c3")
    ra, na = meta["ref_attrs"], meta["new_attrs"]
    ref = Celligner2(**surgery.init_kwargs_from_attrs(ra))
    ref.load_state_dict({k: torch.tensor(v) for k, v in gu.group(z, "ref_state/").items()})
    query = gu.adata_from(z, "query/X", "query/obs/")
    _, new = surgery.new_conditions_from_obs(query.obs, ra["condition_keys_"], ra["conditions_"], miss=ra["miss_"])
    torch.manual_seed(meta["seed_new_model"])
    q = surgery.load_query_model((ref, ra), new_conditions=new, device="cuda:0")
    expanded = {k: v.detach().clone() for k, v in q.state_dict().items()}
    for k, v in gu.group(z, "expanded/").items():
        assert np.array_equal(expanded[k].cpu().numpy(), v), k
    np.random.seed(9)
    tr = Celligner2Trainer(q, query, condition_keys=na["condition_keys_"], predictor_keys=na["predictor_keys_"],
                           rng="reference", log_sync="step", **meta["trainer_kwargs"])
    seen = _spy(tr, q)
    tr.train(**meta["train_kwargs"])
    _compare_calls(seen, gu.calls_of(z))
    for k, v in gu.group(z, "logs/").items():
        got = np.asarray(tr.logs[k], np.float64)
        assert np.all(np.abs(got - v) <= 1e-4 * np.maximum(np.abs(v), 1.0)), (k, got, v)
    moved = 0
    for k, v in gu.group(z, "finetuned/").items():
        got = q.state_dict()[k]
        assert gu.relerr(got.cpu().numpy(), v) < 1e-4, k
        if k in meta["trainable"]:
            moved += int(not torch.equal(got, expanded[k]))
        else:
            assert torch.equal(got, expanded[k]), k
    assert moved == len(meta["trainable"]) == 3


def test_device_index_stream_follows_the_sampling_weights():
    """rng="device": an epoch's rows are drawn with replacement by inverting the cached cumulative distribution of the
    stratifier weights -- zero-weight rows never appear, frequencies follow the weights, epochs differ."""
    from celligner2_amd.trainer import _Loader

    class DS:
        device = torch.device("cuda", 0)

        def __len__(self):
            return 1000

    w = torch.zeros(1000, dtype=torch.float64)
    w[100:200] = 1.0
    w[500:600] = 3.0
    ld = _Loader(DS(), 128, weights=w, num_samples=200000, rng="device")
    torch.manual_seed(3)
    a, b = ld._indices(), ld._indices()
    assert a.shape == (200000,) and a.dtype == torch.int64 and not torch.equal(a, b)
    cnt = torch.bincount(a, minlength=1000).double().cpu()
    assert cnt[w == 0].sum() == 0
    lo, hi = cnt[100:200].sum().item(), cnt[500:600].sum().item()
    assert abs(lo / 200000 - 0.25) < 0.01 and abs(hi / 200000 - 0.75) < 0.01
    assert cnt[100:200].min() > 300 and cnt[500:600].max() < 1900  # 500 and 1500 expected per row
