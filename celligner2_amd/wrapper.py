"""`CELLIGNER2`: the reference's model wrapper (celligner2/model/celligner2_model.py:97-593 with its BaseMixin /
SurgeryMixin, celligner2/othermodels/base/_base.py:47-356) over the MI355X module, trainer and surgery bookkeeping of
this package -- the class a user of the reference builds from an AnnData, trains, maps to the latent space, saves,
loads and extends with query data.

Host logic only: labels from `adata.obs`, the chunk loops of `get_latent` / `reconstruct` (512 rows per call of the
module, celligner2_model.py:428-441, 510-517), the checkpoint directory.  Every number is produced by the module's HIP
kernels (`model.Celligner2.get_latent` / `reconstructLatent`, `trainer.Celligner2Trainer`).

`adata` is anything that looks like an AnnData to the reference's code on this path: `.X` (dense ndarray [n, G], NaN =
missing), `.obs` (pandas DataFrame), `.shape`, `.n_vars`, row slicing `adata[rows]`, optionally `.var_names` /
`.var.index`.  The `anndata` package is not a dependency: where it is importable `get_latent` returns a real AnnData,
otherwise a `LatentTable` with the same `.X` / `.obs` / `.shape`.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import pandas as pd
import torch

from . import surgery
from .dataset import label_encoder_2D
from .model import Celligner2
from .trainer import Celligner2Trainer

CHUNK = 512  # rows per call of the module in get_latent / reconstruct (celligner2_model.py:431, 512)


class LatentTable:
    """What `get_latent` returns where `anndata` is not installed: the latent matrix and the obs columns beside it."""

    def __init__(self, X, obs):
        self.X = X
        self.obs = obs

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
        return LatentTable(self.X[rows], self.obs.iloc[rows].copy())

    def copy(self):
        return LatentTable(self.X.copy(), self.obs.copy())


def _as_anndata(X, obs):
    try:
        from anndata import AnnData  # noqa: PLC0415 -- optional
    except ImportError:
        return LatentTable(X, obs)
    return AnnData(X, obs=obs)


def _var_names(adata):
    if hasattr(adata, "var_names"):
        return np.asarray(adata.var_names).astype(str)
    if hasattr(adata, "var") and hasattr(adata.var, "index"):
        return np.asarray(adata.var.index).astype(str)
    return np.arange(adata.shape[1]).astype(str)


def _labels_of(adata, keys, given, miss, what):
    """Labels of the obs columns `keys` (celligner2_model.py:147-177 conditions, :185-214 predictors): values shared by
    two columns are renamed `<key>_<value>` in `adata.obs` (in place, as the reference does); the list is the set of all
    values minus the missing markers.  The reference takes `list(set(...))`, whose order follows the process' string
    hashing; here the labels are SORTED, so that two runs build the same model -- pass `conditions=` / `predictors=`
    to fix any other order (a model's columns are in the order of these lists)."""
    if given is not None:
        return list(given)
    if keys is None:
        return []
    seen = set()
    for key in keys:
        group = set(adata.obs[key]) - set(miss)
        if what == "condition" and set(adata.obs[key]) & set(miss):
            raise ValueError(f"Condition key '{key}' has missing values. the model can't deal with missing values in its "
                             "condition keys for now, you can run them as predictor to impute them from the data")
        if what == "predictor" and not group:
            raise ValueError(f"Predictor key '{key}' has no values. please check your obs")
        if group & seen:
            adata.obs.replace({key: {val: key + "_" + val for val in group}}, inplace=True)
        seen |= set(adata.obs[key])
    if not keys:
        return []
    return sorted(set(np.asarray(adata.obs[list(keys)].values).flatten()) - set(miss))


class CELLIGNER2:
    """celligner2_model.py:97-312.  Same constructor keywords; `precision` ("fp32" | "bf16") and `device` are this build's.
    The expimap / GNN / HSIC / `n_expand` options raise in the module's constructor."""

    def __init__(self, adata, hidden_layer_sizes: list = [256, 64], latent_dim: int = 10, dr_rate: float = 0.05,
                 use_bn: bool = False, use_ln: bool = True, use_own_kl: bool = False, miss: str = "U",
                 apply_log: bool = True, condition_keys: Optional[list] = None, conditions: Optional[list] = None,
                 use_mmd: bool = True, mmd_on: str = "z", mmd_boundary: Optional[int] = None,
                 recon_loss: Optional[str] = "nb", beta: float = 1, batch_knowledge: bool = True, main_dataset=None,
                 classifier_hidden_layer_sizes: list = [64, 32], betaclass: float = 0.8,
                 predictors: Optional[list] = None, predictor_keys: Optional[list] = [], res_mult: int = 0,
                 graph_layers: int = 0, mask=None, mask_key: str = "", n_unconstrained: int = 0, use_hsic: bool = False,
                 hsic_one_vs_all: bool = False, use_l_encoder: bool = False, n_expand: int = 0, ext_mask=None,
                 ext_n_unconstrained: int = 0, predictor_set={}, condition_set={}, precision: str = "fp32",
                 device="cuda"):
        self.adata = adata
        self.goodloc = ~np.isnan(np.asarray(adata.X))
        self.condition_keys_ = condition_keys
        self.predictor_keys_ = predictor_keys
        self.conditions_ = _labels_of(adata, condition_keys, conditions, miss, "condition")
        if condition_keys is not None:
            self.condition_set_ = {key: set(adata.obs[key]) - set(miss) for key in condition_keys}
        else:
            self.condition_set_ = {}
        self.predictors_ = _labels_of(adata, predictor_keys, predictors, miss, "predictor")
        if predictor_keys is not None:
            self.predictor_set_ = {key: set(adata.obs[key]) - set(miss) for key in predictor_keys}
        else:
            self.predictor_set_ = {}
        for k, v in predictor_set.items():
            self.predictor_set_[k] = set(v) & set(self.predictor_set_[k])
        if mask is not None or mask_key != "":
            raise NotImplementedError("mask / mask_key (expimap mode) belong to the reference's unfinished branches "
                                      "(README.md:31-43) and are outside the accelerated path")
        self.miss_ = miss
        self.hidden_layer_sizes_ = hidden_layer_sizes
        self.classifier_hidden_layer_sizes_ = classifier_hidden_layer_sizes
        self.latent_dim_ = latent_dim
        self.dr_rate_ = dr_rate
        self.use_mmd_ = use_mmd
        self.mmd_on_ = mmd_on
        self.mmd_boundary_ = mmd_boundary
        self.expimap_mode_ = False
        self.mask_ = None
        self.n_unconstrained_ = n_unconstrained
        self.use_hsic_ = use_hsic
        self.hsic_one_vs_all_ = hsic_one_vs_all
        self.ext_mask_ = ext_mask
        self.recon_loss_ = recon_loss
        self.beta_ = beta
        self.betaclass_ = betaclass
        self.use_bn_ = use_bn
        self.use_ln_ = use_ln
        self.use_own_kl_ = use_own_kl
        self.n_expand_ = n_expand
        self.use_l_encoder_ = use_l_encoder
        self.ext_n_unconstrained_ = ext_n_unconstrained
        self.res_mult_ = res_mult
        self.graph_layers_ = graph_layers
        self.batch_knowledge_ = batch_knowledge
        self.input_dim_ = adata.n_vars
        self.apply_log_ = apply_log
        known = set(np.asarray(adata.obs[list(condition_keys)].values).flatten()) if condition_keys else set()
        if main_dataset not in known:
            if main_dataset is not None:
                print("main dataset not in conditions, removing..")
            self.main_dataset_ = None
        else:
            self.main_dataset_ = main_dataset
        self.precision_ = precision
        self.model = Celligner2(
            self.input_dim_, self.hidden_layer_sizes_, self.latent_dim_, self.dr_rate_, self.use_own_kl_, self.recon_loss_,
            self.use_bn_, self.use_ln_, self.apply_log_, self.conditions_, self.use_mmd_, self.mmd_on_, self.mmd_boundary_,
            self.beta_, self.main_dataset_, self.batch_knowledge_, self.predictors_, self.classifier_hidden_layer_sizes_,
            self.betaclass_, self.graph_layers_, self.res_mult_, self.n_expand_, self.expimap_mode_ or None, self.mask_,
            self.ext_mask_, self.n_unconstrained_, self.ext_n_unconstrained_, self.use_l_encoder_, self.use_hsic_,
            self.hsic_one_vs_all_, precision=precision)
        self.model.to(device)
        self.is_trained_ = False
        self.trainer = None

    # ------------------------------------------------------------------ training
    def train(self, n_epochs: int = 400, lr: float = 1e-3, eps: float = 0.01, **kwargs):
        """celligner2_model.py:314-336."""
        self.trainer = Celligner2Trainer(self.model, self.adata, condition_keys=self.condition_keys_,
                                         predictor_keys=self.predictor_keys_, **kwargs)
        self.trainer.train(n_epochs, lr, eps)
        self.is_trained_ = True

    # ------------------------------------------------------------------ inference
    def _conditions_of(self, adata):
        return torch.tensor(label_encoder_2D(adata, encoder=self.model.condition_encoder, label_sets=self.condition_set_),
                            dtype=torch.long)

    @torch.no_grad()
    def get_latent(self, adata=None, mean: bool = False, add_classpred: bool = False, get_fullpred: bool = False):
        """celligner2_model.py:392-482: the rows through the encoder (and the classifier on the latent mean) 512 at a time;
        returns (AnnData-like table of the latent codes with the condition / predictor columns and one `<key>_pred` column
        per predictor key, the raw class scores or None)."""
        device = next(self.model.parameters()).device
        if adata is None:
            adata = self.adata
        c = self._conditions_of(adata)
        x = torch.as_tensor(np.asarray(adata.X))
        latents, scores = [], []
        was_training = self.model.training
        self.model.eval()
        try:
            for rows in torch.arange(x.size(0)).split(CHUNK):
                latent, classes = self.model.get_latent(x[rows, :].to(device), c[rows].to(device), mean, add_classpred)
                latents.append(latent.cpu())
                if add_classpred:
                    scores.append(classes.cpu())
        finally:
            self.model.train(was_training)
        predictions = None
        if add_classpred:
            decoder = {v: k for k, v in self.model.predictor_encoder.items()}
            predictions = torch.cat(scores).numpy()
            columns = []
            for _, labels in self.predictor_set_.items():
                if len(labels) == 0:
                    columns.append([""] * len(predictions))
                else:
                    loc = np.array([self.model.predictor_encoder[item] for item in labels])
                    columns.append([decoder[name] for name in loc[np.argmax(predictions[:, loc], axis=1)]])
            classes = pd.DataFrame(data=np.array(columns).T.reshape(len(predictions), len(columns)),
                                   columns=[k + "_pred" for k in self.predictor_set_.keys()], index=adata.obs.index)
        else:
            classes = pd.DataFrame(index=adata.obs.index)
        keys = list(self.condition_keys_ or []) + list(self.predictor_keys_ or [])
        res = _as_anndata(torch.cat(latents).numpy(), pd.concat([adata.obs[keys], classes], axis=1))
        if get_fullpred and add_classpred:
            return res, predictions
        return res, None

    @torch.no_grad()
    def reconstruct(self, latent: np.ndarray, c):
        """celligner2_model.py:484-520: latent codes [n, L] and their condition columns (a DataFrame with the condition
        keys) through the decoder, 512 rows at a time; a DataFrame [n, G] over the training AnnData's gene names."""
        device = next(self.model.parameters()).device
        latent = np.asarray(latent)
        cond = self._conditions_of(LatentTable(latent, c))
        lat = torch.as_tensor(latent)
        out = []
        was_training = self.model.training
        self.model.eval()
        try:
            for rows in torch.arange(lat.size(0)).split(CHUNK):
                expression = self.model.reconstructLatent(lat[rows, :].to(device), cond[rows].to(device))[0]
                out.append(expression.cpu())
        finally:
            self.model.train(was_training)
        return pd.DataFrame(torch.cat(out).numpy(), columns=_var_names(self.adata))

    # ------------------------------------------------------------------ checkpoint directory
    def _get_public_attributes(self):
        """_base.py:20-30: every `<name>_` attribute."""
        return {k: v for k, v in self.__dict__.items() if k.endswith("_") and not k.startswith("_")
                and k != "precision_"}

    def save(self, dir_path: str, overwrite: bool = False, save_anndata: bool = False):
        """_base.py:47-95: `model_params.pt`, `attr.pkl`, `var_names.csv`.  (`save_anndata` needs the real AnnData's
        `write`: honoured when the object has one.)"""
        surgery.save(dir_path, self.model, self._get_public_attributes(), _var_names(self.adata), overwrite=overwrite)
        if save_anndata:
            if not hasattr(self.adata, "write"):
                raise NotImplementedError("save_anndata=True needs an AnnData (its .write); this object has none")
            self.adata.write(os.path.join(dir_path, "adata.h5ad"))

    @classmethod
    def _get_init_params_from_dict(cls, dct):
        """celligner2_model.py:338-385."""
        pick = lambda k, d=None: dct[k] if k in dct else d  # noqa: E731
        return {
            "condition_keys": dct["condition_keys_"], "conditions": dct["conditions_"],
            "hidden_layer_sizes": dct["hidden_layer_sizes_"],
            "classifier_hidden_layer_sizes": pick("classifier_hidden_layer_sizes_", []),
            "latent_dim": dct["latent_dim_"], "dr_rate": dct["dr_rate_"], "use_mmd": dct["use_mmd_"],
            "mmd_on": dct["mmd_on_"], "mmd_boundary": dct["mmd_boundary_"], "recon_loss": dct["recon_loss_"],
            "beta": dct["beta_"], "betaclass": dct["betaclass_"], "use_bn": dct["use_bn_"], "use_ln": dct["use_ln_"],
            "use_own_kl": dct["use_own_kl_"], "predictors": dct["predictors_"], "predictor_keys": dct["predictor_keys_"],
            "miss": dct["miss_"], "apply_log": dct["apply_log_"], "predictor_set": pick("predictor_set_", {}),
            "condition_set": pick("condition_set_", {}), "main_dataset": pick("main_dataset_"),
            "n_unconstrained": pick("n_unconstrained_", 0), "use_hsic": pick("use_hsic_", False),
            "hsic_one_vs_all": pick("hsic_one_vs_all_", False),
        }

    @classmethod
    def load(cls, dir_path: str, adata=None, map_location=None, precision: str = "fp32", device="cuda"):
        """_base.py:151-194: rebuild the wrapper from a checkpoint directory (written here or by the reference) and the
        AnnData it was trained on; eval mode."""
        attr_dict, state, var_names = surgery.load_params(dir_path, map_location)
        if adata is None:
            raise ValueError("pass the AnnData (reading adata.h5ad needs the anndata package)")
        if adata.n_vars != attr_dict["input_dim_"]:
            raise ValueError("Incorrect var dimension")
        if attr_dict.get("mask_") is not None or attr_dict.get("ext_mask_") is not None:
            raise NotImplementedError("expimap checkpoints are outside the accelerated path")
        new = cls(adata, precision=precision, device=device, **cls._get_init_params_from_dict(attr_dict))
        new.model.load_state_dict(state)
        new.model.eval()
        new.is_trained_ = attr_dict.get("is_trained_", True)
        return new

    @classmethod
    def load_query_data(cls, adata, reference_model, freeze: bool = True, freeze_expression: bool = True,
                        unfreeze_new: bool = True, remove_dropout: bool = True, precision: Optional[str] = None,
                        device="cuda", **kwargs):
        """_base.py:199-356 + celligner2_model.py:522-593: a wrapper for query data whose condition list is the reference
        model's plus the query's new labels; the reference weights fill the leading block of every tensor that grew; the
        surgery freeze leaves the dispersion and the condition weights (or every first layer) trainable."""
        if isinstance(reference_model, (str, os.PathLike)):
            attr_dict, state, _ = surgery.load_params(reference_model)
        else:
            attr_dict = reference_model._get_public_attributes()
            state = {k: v.detach().clone() for k, v in reference_model.model.state_dict().items()}
            if precision is None:
                precision = reference_model.precision_
        if adata.n_vars != attr_dict["input_dim_"]:
            raise ValueError("Incorrect var dimension")
        init = cls._get_init_params_from_dict(attr_dict)
        renamed, new_conditions = surgery.new_conditions_from_obs(adata.obs, init["condition_keys"], init["conditions"],
                                                                  miss=init["miss"])
        for key, values in renamed.items():
            adata.obs[key] = values
        init["conditions"] = list(init["conditions"]) + list(new_conditions)
        for key in init["predictor_keys"] or []:  # _base.py:295-332: predictor labels may be renamed, never added
            group = set(adata.obs[key]) - set(init["miss"])
            if not group:
                raise ValueError(f"Predictor key '{key}' has no values. please check your obs")
            if any(str(v).startswith(key + "_") for v in init["predictors"]):
                adata.obs.replace({key: {val: key + "_" + val for val in group}}, inplace=True)
                group = set(adata.obs[key]) - set(init["miss"])
            new = group - set(init["predictors"])
            if new:
                raise ValueError("can't add new predictions for now, seeing: ", new)
        if remove_dropout:
            init["dr_rate"] = 0.0
        init.update(kwargs)
        new = cls(adata, precision=precision or "fp32", device=device, **init)
        surgery.load_expand_params(new.model, state)
        if freeze:
            surgery.apply_freeze(new.model, freeze_expression)
        return new
