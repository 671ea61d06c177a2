"""Golden vectors of the reference's TRAINER LOOP and SURGERY, generated from the reference itself (build container
only; nothing of the reference ships -- the fixtures are plain arrays + JSON).

trainer_* cases: the reference's `Celligner2Trainer` (trainers/celligner2/trainer.py:66-370, semisupervised.py:55-74,
utils/monitor.py:4-115, trainers/celligner2/_utils.py:79-185, dataset/celligner2/anndata.py:8-166) is run for a few
epochs on a small AnnData-like object.  Recorded per `loss()` call, in call order: train/validation flag, the row
indices the DataLoader fetched, the eps draw, the learning rate and the KL weight in force; then `logs`, the final
state_dict, the split, the stratifier weights, best_epoch and the number of epochs run.  A run of
celligner2_amd.trainer.Celligner2Trainer(rng="reference", log_sync="step") on the same inputs must visit the same rows
with the same draws and end with the same logs and weights (tests/test_trainer_gpu.py).

surgery_* case: the reference's wrapper `CELLIGNER2` (model/celligner2_model.py:98-336, 522-593) and
`SurgeryMixin.load_query_data` (othermodels/base/_base.py:97-134, 199-356): a 2-condition reference model, a query
AnnData with a third condition -> the expanded state_dict, the set of trainable parameter names, the wrapper's
attribute dict (JSON), and the state after a short fine-tune by the reference trainer.

Harness patches (recorded in the metadata): eps draws are recorded by replacing
`torch.distributions.normal._standard_normal` with a recording version that draws exactly what the original draws;
`pdb.set_trace` is a no-op; the wrapper's numpy-1.x private import gets a stand-in.

Usage: python tools/gen_golden_trainer.py [case ...]        (needs /root/reference)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def make_obs_matrix(rng, N, G, datasets, lineages, celltypes, nan_frac=0.01, miss_frac=0.2):
    import pandas as pd
    depth = rng.uniform(0.5, 2.0, size=(N, 1))
    gene = np.exp(rng.normal(1.0, 1.0, size=(1, G)))
    ds = rng.choice(len(datasets), size=N, p=np.linspace(2, 1, len(datasets)) / np.linspace(2, 1, len(datasets)).sum())
    ds[: len(datasets)] = np.arange(len(datasets))
    shift = np.exp(rng.normal(0.0, 0.3, size=(len(datasets), G)))
    X = rng.poisson(depth * gene * shift[ds]).astype(np.float32)
    X[rng.uniform(size=(N, G)) < nan_frac] = np.nan
    lin = rng.integers(0, len(lineages), size=N)
    lin[: len(lineages)] = np.arange(len(lineages))
    lin_s = np.array(lineages, dtype=object)[lin]
    lin_s[rng.uniform(size=N) < miss_frac] = "U"
    lin_s[: len(lineages)] = np.array(lineages, dtype=object)
    ct = np.array(celltypes, dtype=object)[rng.integers(0, len(celltypes), size=N)]
    obs = pd.DataFrame({"dataset": np.array(datasets, dtype=object)[ds], "lineage": lin_s, "celltype": ct},
                       index=[f"s{i}" for i in range(N)])
    return X, obs


TRAINER_CASES = {
    "trainer_nb_stratified_earlystop": dict(
        data=dict(N=210, G=64, datasets=["ccle", "tcga", "gtex"], lineages=["lung", "skin", "blood", "bone"],
                  celltypes=["t0", "t1"]),
        model=dict(hidden_layer_sizes=[32, 16], latent_dim=8, dr_rate=0.0, recon_loss="nb", use_mmd=True, mmd_on="z",
                   beta=20.0, classifier_hidden_layer_sizes=[16]),
        trainer=dict(condition_keys=["dataset"], predictor_keys=["lineage"], cell_type_key="celltype", batch_size=32,
                     alpha_epoch_anneal=3, alpha_kl=5e-3, train_frac=0.75, seed=2021, monitor=False,
                     early_stopping_kwargs=dict(early_stopping_metric="val_unweighted_loss", threshold=1e9, patience=2,
                                                lr_patience=1, reduce_lr=True, lr_factor=0.1)),
        train=dict(n_epochs=6, lr=2e-3, eps=0.01), np_seed=11),
    "trainer_nb_shuffled_dropout_clip": dict(
        data=dict(N=150, G=48, datasets=["a", "b"], lineages=["x", "y", "z"], celltypes=["c"]),
        model=dict(hidden_layer_sizes=[24, 24, 12], latent_dim=6, dr_rate=0.1, recon_loss="nb", use_mmd=True, mmd_on="y",
                   beta=5.0, classifier_hidden_layer_sizes=[8, 8], use_own_kl=True),
        trainer=dict(condition_keys=["dataset"], predictor_keys=["lineage"], cell_type_key=None, use_leiden=False,
                     batch_size=48, alpha_iter_anneal=5, alpha_kl=1e-3, train_frac=0.8, seed=7, monitor=False,
                     use_stratified_sampling=False, use_early_stopping=False, clip_value=0.5, weight_decay=0.01),
        train=dict(n_epochs=2, lr=1e-3, eps=0.01), np_seed=12),
}


class _Recorder:
    """Records, per loss() call of a reference trainer, what the reference consumed."""

    def __init__(self, ns):
        import torch
        from torch.distributions import normal as tnormal
        self.ns, self.torch, self.tnormal = ns, torch, tnormal
        self.rows, self.calls = [], []
        self._eps = []
        self._orig_get = ns.AnnotatedDataset.__getitem__
        self._orig_sn = tnormal._standard_normal
        rec = self

        def getitem(ds, index):
            rec.rows.append(int(index))
            return rec._orig_get(ds, index)

        def standard_normal(shape, dtype, device):
            e = rec._orig_sn(shape, dtype, device)
            rec._eps.append(e.detach().numpy().copy())
            return e

        ns.AnnotatedDataset.__getitem__ = getitem
        tnormal._standard_normal = standard_normal

    def close(self):
        self.ns.AnnotatedDataset.__getitem__ = self._orig_get
        self.tnormal._standard_normal = self._orig_sn

    def trainer_class(self):
        rec = self

        class Recording(self.ns.Celligner2Trainer):
            def loss(self, total_batch=None):
                n = int(total_batch["x"].shape[0])
                rows = rec.rows[:n]
                assert len(rec.rows) >= n
                del rec.rows[:n]
                before = len(rec._eps)
                out = super().loss(total_batch)
                assert len(rec._eps) == before + 1
                rec.calls.append(dict(train=int(self.model.training), rows=rows, eps=rec._eps[-1],
                                      lr=float(self.optimizer.param_groups[0]["lr"]),
                                      alpha=float(self.calc_alpha_coeff()), epoch=int(self.epoch)))
                return out

        return Recording


def _pack_calls(calls, arrays, prefix="calls/"):
    arrays[prefix + "train"] = np.array([c["train"] for c in calls], np.int64)
    arrays[prefix + "size"] = np.array([len(c["rows"]) for c in calls], np.int64)
    arrays[prefix + "rows"] = np.concatenate([np.array(c["rows"], np.int64) for c in calls])
    arrays[prefix + "eps"] = np.concatenate([c["eps"] for c in calls], 0).astype(np.float32)
    arrays[prefix + "lr"] = np.array([c["lr"] for c in calls], np.float64)
    arrays[prefix + "alpha"] = np.array([c["alpha"] for c in calls], np.float64)
    arrays[prefix + "epoch"] = np.array([c["epoch"] for c in calls], np.int64)


def _obs_arrays(obs, arrays, prefix="obs/"):
    for col in obs.columns:
        arrays[prefix + col] = np.array([str(v) for v in obs[col]], dtype="U")


def run_trainer_case(ns, name, spec):
    import torch
    rng = np.random.default_rng(spec["np_seed"])
    X, obs = make_obs_matrix(rng, **spec["data"])
    datasets, lineages = spec["data"]["datasets"], spec["data"]["lineages"]
    mk = dict(spec["model"])
    torch.manual_seed(spec["np_seed"])
    model = ns.Celligner2(X.shape[1], conditions=list(datasets), predictors=list(lineages), **mk)
    arrays = {"X": X.copy()}
    _obs_arrays(obs, arrays)
    for k, v in model.state_dict().items():
        arrays["p0/" + k] = v.detach().numpy().copy()
    adata = ns.AnnDataLike(X.copy(), obs.copy())
    rec = _Recorder(ns)
    try:
        np.random.seed(spec["np_seed"])  # the split uses numpy's GLOBAL generator (trainers/celligner2/_utils.py:105,112)
        tr = rec.trainer_class()(model, adata, **spec["trainer"])
        arrays["split/train_idx_sizefactors"] = tr.train_data.size_factors.numpy().reshape(-1).astype(np.float64)
        arrays["split/n_train"] = np.array([len(tr.train_data)])
        arrays["split/n_valid"] = np.array([len(tr.valid_data) if tr.valid_data is not None else 0])
        arrays["split/train_conditions"] = tr.train_data.conditions.numpy()
        arrays["split/train_predictors"] = tr.train_data.predictors.numpy()
        arrays["split/train_x"] = tr.train_data.data.numpy()
        arrays["split/train_goodloc"] = tr.train_data.goodloc.numpy()
        if tr.valid_data is not None:
            arrays["split/valid_x_rowsum"] = tr.valid_data.data.numpy().sum(1).astype(np.float64)
        arrays["split/stratifier_weights"] = np.asarray(tr.train_data.stratifier_weights, np.float64)
        tr.train(**spec["train"])
    finally:
        rec.close()
    _pack_calls(rec.calls, arrays)
    for k, v in tr.logs.items():
        arrays["logs/" + k] = np.asarray(v, np.float64)
    for k, v in model.state_dict().items():
        arrays["final/" + k] = v.detach().numpy().copy()
    meta = dict(case=name, kind="trainer", model_kwargs=mk, conditions=list(datasets), predictors=list(lineages),
                trainer_kwargs=spec["trainer"], train_kwargs=spec["train"], np_seed=spec["np_seed"],
                epochs_run=int(tr.epoch) + 1, best_epoch=(None if tr.best_epoch is None else int(tr.best_epoch)),
                lr_final=float(tr.optimizer.param_groups[0]["lr"]), iters_per_epoch=int(tr.iters_per_epoch),
                torch_version=torch.__version__,
                patches=["recording AnnotatedDataset.__getitem__", "recording _standard_normal"])
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(name, "epochs", meta["epochs_run"], "calls", len(rec.calls), "lr_final", meta["lr_final"], "best", meta["best_epoch"],
          "val_unweighted", tr.logs.get("val_unweighted_loss"), "KB", os.path.getsize(path) // 1024)


def _jsonable(v):
    if isinstance(v, (set, frozenset)):
        return sorted(_jsonable(x) for x in v)
    if isinstance(v, dict):
        return {str(k): _jsonable(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if v is None or isinstance(v, (bool, int, float, str)):
        return v
    return repr(v)


def run_surgery_case(ns, name="surgery_c2_to_c3"):
    import torch
    rng = np.random.default_rng(21)
    G = 56
    lineages = ["lung", "skin", "blood"]
    Xr, obs_r = make_obs_matrix(rng, 120, G, ["ccle", "tcga"], lineages, ["c"], nan_frac=0.0, miss_frac=0.15)
    Xq, obs_q = make_obs_matrix(rng, 64, G, ["met500", "tcga"], lineages, ["c"], nan_frac=0.0, miss_frac=0.15)
    torch.manual_seed(5)
    ref = ns.CELLIGNER2(ns.AnnDataLike(Xr.copy(), obs_r.copy()), hidden_layer_sizes=[24, 12], latent_dim=6, dr_rate=0.05,
                        condition_keys=["dataset"], conditions=["ccle", "tcga"], predictor_keys=["lineage"],
                        predictors=list(lineages), use_mmd=True, mmd_on="z", beta=10.0, recon_loss="nb",
                        classifier_hidden_layer_sizes=[16])
    with torch.no_grad():  # a reference model a little away from its initialisation
        for p in ref.model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    arrays = {"ref/X": Xr, "query/X": Xq}
    _obs_arrays(obs_r, arrays, "ref/obs/")
    _obs_arrays(obs_q, arrays, "query/obs/")
    for k, v in ref.model.state_dict().items():
        arrays["ref_state/" + k] = v.detach().numpy().copy()
    ref_attrs = _jsonable(ref._get_public_attributes())
    torch.manual_seed(6)  # the new model's fresh tensors (the added condition columns) come from this seed
    adata_q = ns.AnnDataLike(Xq.copy(), obs_q.copy())
    new = ns.CELLIGNER2.load_query_data(adata_q, ref, freeze=True, freeze_expression=True)
    for k, v in new.model.state_dict().items():
        arrays["expanded/" + k] = v.detach().numpy().copy()
    trainable = [n for n, p in new.model.named_parameters() if p.requires_grad]
    new_attrs = _jsonable(new._get_public_attributes())
    # short fine-tune by the reference trainer
    rec = _Recorder(ns)
    tkw = dict(cell_type_key=None, use_leiden=False, batch_size=32, alpha_kl=1e-3, train_frac=0.9, seed=3, monitor=False,
               use_early_stopping=False)
    try:
        np.random.seed(9)
        orig = ns.wrapper_mod.Celligner2Trainer
        ns.wrapper_mod.Celligner2Trainer = rec.trainer_class()
        try:
            new.train(n_epochs=2, lr=1e-3, eps=0.01, **tkw)
        finally:
            ns.wrapper_mod.Celligner2Trainer = orig
    finally:
        rec.close()
    _pack_calls(rec.calls, arrays)
    for k, v in new.model.state_dict().items():
        arrays["finetuned/" + k] = v.detach().numpy().copy()
    for k, v in new.trainer.logs.items():
        arrays["logs/" + k] = np.asarray(v, np.float64)
    meta = dict(case=name, kind="surgery", ref_attrs=ref_attrs, new_attrs=new_attrs, trainable=trainable,
                seed_new_model=6, trainer_kwargs=tkw, train_kwargs=dict(n_epochs=2, lr=1e-3, eps=0.01),
                frozen=bool(new.model.freeze), torch_version=torch.__version__,
                patches=["pdb.set_trace no-op", "numpy.lib.function_base stand-in", "recording __getitem__/_standard_normal"])
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(name, "conditions", new_attrs["conditions_"], "trainable", trainable, "calls", len(rec.calls),
          "KB", os.path.getsize(path) // 1024)


def main():
    ns = ref_harness.load_wrapper()
    only = sys.argv[1:]
    for name, spec in TRAINER_CASES.items():
        if not only or name in only:
            run_trainer_case(ns, name, spec)
    if not only or "surgery_c2_to_c3" in only:
        run_surgery_case(ns)


if __name__ == "__main__":
    main()
