"""Surgery bookkeeping and the checkpoint directory of the reference's model wrapper, for the `Celligner2` module
(celligner2/othermodels/base/_base.py:47-149 save / _load_params / _load_expand_params_from_dict, :199-356
load_query_data; celligner2/model/celligner2_model.py:338-385 _get_init_params_from_dict).

Host logic only (no kernels): a checkpoint written by the reference loads here and one written here loads there --
the directory holds `model_params.pt` (the state_dict, identical keys and shapes), `attr.pkl` (the wrapper's public
attributes, the `<name>_` dict) and `var_names.csv`.  The AnnData side of `load_query_data` (renaming and collecting
condition values from `adata.obs`) is `new_conditions_from_obs`, on any mapping of column name -> values.

`attr.pkl` is a pickle: it is read with a RESTRICTED unpickler that builds only plain containers, scalars and numpy
arrays (what the reference's attribute dict holds when `mask_` is None); anything else is refused.
"""
from __future__ import annotations

import io
import os
import pickle
from copy import deepcopy

import numpy as np
import torch

from .model import Celligner2

# wrapper attribute (attr.pkl key) -> Celligner2 constructor keyword (celligner2_model.py:274-309, 338-385)
_ATTR_TO_KW = {
    "hidden_layer_sizes_": "hidden_layer_sizes", "latent_dim_": "latent_dim", "dr_rate_": "dr_rate",
    "use_own_kl_": "use_own_kl", "recon_loss_": "recon_loss", "use_bn_": "use_bn", "use_ln_": "use_ln",
    "apply_log_": "applylog", "conditions_": "conditions", "use_mmd_": "use_mmd", "mmd_on_": "mmd_on",
    "mmd_boundary_": "mmd_boundary", "beta_": "beta", "main_dataset_": "main_dataset",
    "batch_knowledge_": "batch_knowledge", "predictors_": "predictors",
    "classifier_hidden_layer_sizes_": "classifier_hidden_layer_sizes", "betaclass_": "betaclass",
}
_DEFAULTS = {"use_own_kl_": False, "use_bn_": False, "use_ln_": True, "apply_log_": True, "main_dataset_": None,
             "batch_knowledge_": True, "classifier_hidden_layer_sizes_": [], "mmd_boundary_": None}


def public_attributes(model: Celligner2, condition_keys=None, predictor_keys=None, miss="U", is_trained=False,
                      condition_set=None, predictor_set=None, **extra):
    """The wrapper's `<name>_` attribute dict (celligner2_model.py:171-272) for a bare module: the key set and value types
    of what the reference's `save` pickles (pinned by tests/golden/checkpoint_dir.npz, written by the reference itself).
    condition_set / predictor_set: obs column -> set of its labels (celligner2_model.py:180-223); by default a single key
    owns every label of the module, several keys get empty sets (the split is the AnnData's knowledge, not the module's).
    miss: the wrapper's marker(s) for a missing label, kept as given (the reference's default is the string "U")."""
    def label_sets(keys, given, labels):
        if given is not None:
            return {k: set(v) for k, v in given.items()}
        if not keys:
            return {}
        return {k: (set(labels) if len(keys) == 1 else set()) for k in keys}

    d = {
        "condition_keys_": condition_keys, "conditions_": list(model.conditions), "predictor_keys_": predictor_keys,
        "predictors_": list(model.predictors), "miss_": miss,
        "condition_set_": label_sets(condition_keys, condition_set, model.conditions),
        "predictor_set_": label_sets(predictor_keys, predictor_set, model.predictors),
        "hidden_layer_sizes_": list(model.hidden_layer_sizes),
        "classifier_hidden_layer_sizes_": list(model.classifier_hidden_layer_sizes), "latent_dim_": model.latent_dim,
        "dr_rate_": model.dr_rate, "use_mmd_": model.use_mmd, "mmd_on_": model.mmd_on,
        "mmd_boundary_": model.mmd_boundary, "recon_loss_": model.recon_loss, "beta_": model.beta,
        "betaclass_": model.betaclass, "use_bn_": model.use_bn, "use_ln_": model.use_ln,
        "use_own_kl_": model.use_own_kl, "apply_log_": model.applylog, "main_dataset_": model.main_dataset,
        "batch_knowledge_": model.batch_knowledge, "input_dim_": model.input_dim, "is_trained_": bool(is_trained),
        "mask_": None, "n_unconstrained_": 0, "use_hsic_": False, "hsic_one_vs_all_": False, "n_expand_": 0,
        "expimap_mode_": False, "ext_mask_": None, "use_l_encoder_": False, "ext_n_unconstrained_": 0,
        "res_mult_": model.res_mult, "graph_layers_": model.graph_layers,
    }
    d.update(extra)
    return d


def save(dir_path, model: Celligner2, attrs: dict, var_names, overwrite=False):
    """_base.py:47-95: model_params.pt + attr.pkl + var_names.csv (the AnnData itself is not written)."""
    if not os.path.exists(dir_path) or overwrite:
        os.makedirs(dir_path, exist_ok=overwrite)
    else:
        raise ValueError("{} already exists. Please provide an unexisting directory for saving.".format(dir_path))
    np.savetxt(os.path.join(dir_path, "var_names.csv"), np.asarray(var_names).astype(str), fmt="%s")
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, os.path.join(dir_path, "model_params.pt"))
    with open(os.path.join(dir_path, "attr.pkl"), "wb") as f:
        pickle.dump(attrs, f)


class _PlainUnpickler(pickle.Unpickler):
    _OK = {("builtins", n) for n in ("list", "dict", "set", "frozenset", "tuple", "str", "int", "float", "bool",
                                     "complex", "bytes", "slice", "range")}
    _OK |= {("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"),
            ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
            ("numpy._core.multiarray", "scalar"), ("collections", "OrderedDict")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"attr.pkl holds a {module}.{name} object: only plain containers, scalars and "
                                     "numpy arrays are loaded (expimap masks / custom objects are outside the path)")


def load_params(dir_path, map_location=None):
    """_base.py:136-149 with loaders that execute nothing from the files."""
    with open(os.path.join(dir_path, "attr.pkl"), "rb") as f:
        attr_dict = _PlainUnpickler(io.BytesIO(f.read())).load()
    state = torch.load(os.path.join(dir_path, "model_params.pt"), map_location=map_location or "cpu",
                       weights_only=True)
    var_names = np.genfromtxt(os.path.join(dir_path, "var_names.csv"), delimiter=",", dtype=str)
    return attr_dict, state, var_names


def init_kwargs_from_attrs(attr_dict):
    """Constructor keywords of the module from the wrapper's attribute dict (celligner2_model.py:274-309, 338-385)."""
    kw = {}
    for a, k in _ATTR_TO_KW.items():
        if a in attr_dict:
            kw[k] = deepcopy(attr_dict[a])
        elif a in _DEFAULTS:
            kw[k] = deepcopy(_DEFAULTS[a])
        else:
            raise KeyError(f"attr.pkl lacks {a}")
    for a in ("mask_", "ext_mask_"):
        if attr_dict.get(a) is not None:
            raise NotImplementedError(f"{a}: expimap checkpoints are outside the accelerated path")
    kw["input_dim"] = attr_dict["input_dim_"]
    return kw


def load(dir_path, device="cuda", precision="fp32"):
    """_base.py:151-194 for the bare module: rebuild it from attr.pkl and load model_params.pt; eval mode."""
    attr_dict, state, var_names = load_params(dir_path)
    model = Celligner2(precision=precision, **init_kwargs_from_attrs(attr_dict))
    model.load_state_dict(state)
    model.to(device).eval()
    return model, attr_dict, var_names


def load_expand_params(model: Celligner2, state_dict):
    """Load a checkpoint into a model that may have GROWN since (new condition columns in `theta [G,C]` and
    `*.cond_L.weight [H,C]`): for every tensor of the model, in the model's own key order -- absent from the checkpoint:
    keep the fresh value; same shape: take the checkpoint's; larger in some dimension: checkpoint values in the leading
    block, fresh values elsewhere.  Shrinking, a change of rank or more than 3 dimensions is refused; keys the model does
    not have make `load_state_dict` raise, as in the reference (_base.py:97-134)."""
    fresh = model.state_dict()
    merged = {k: v for k, v in state_dict.items() if k not in fresh}  # unexpected keys: left for load_state_dict to reject
    for key, new in fresh.items():
        old = state_dict.get(key)
        if old is None or old.shape == new.shape:
            merged[key] = new if old is None else old
            continue
        if old.dim() != new.dim():
            raise ValueError(f"{key}: rank changed from {old.dim()} to {new.dim()}")
        if new.dim() > 3:
            raise ValueError("Too many dimensions")
        if any(o > n for o, n in zip(old.shape, new.shape)):
            raise ValueError("Can only increase the size of the model for now")
        grown = new.clone()
        grown[tuple(slice(0, o) for o in old.shape)] = old.to(new.device)
        merged[key] = grown
    model.load_state_dict(merged)
    return model


# parameter-name fragments that stay trainable after surgery (_base.py:342-354)
_TRAINABLE_AFTER_SURGERY = {True: ("theta", "cond_L.weight"),   # freeze_expression: dispersion + condition weights only
                            False: ("theta", "L0", "N0")}       # else: dispersion + every first layer and its norm


def apply_freeze(model: Celligner2, freeze_expression=True):
    """Surgery freeze: `model.freeze = True` and requires_grad only on the parameters whose NAME contains one of the
    fragments above (matching by substring, as the reference does)."""
    model.freeze = True
    keep = _TRAINABLE_AFTER_SURGERY[bool(freeze_expression)]
    for name, p in model.named_parameters():
        p.requires_grad = any(fragment in name for fragment in keep)
    return model


def new_conditions_from_obs(obs, condition_keys, prev_conditions, miss=()):
    """_base.py:236-280: values of the query's condition columns that the reference model does not know, in the
    order the reference appends them; columns are renamed `<key>_<value>` when the reference model's conditions
    were.  `obs`: mapping column name -> sequence of values (a pandas DataFrame works); returns (obs columns after
    the renaming as a dict, list of new conditions)."""
    out_obs, new_all, seen = {}, [], set()
    for key in condition_keys:
        vals = [str(v) for v in obs[key]]
        if set(vals) & set(miss):
            raise ValueError(f"Condition key '{key}' has missing values. the model can't deal with missing values in "
                             "its condition keys for now, you can run them as predictor to impute them from the data")
        if any(str(v).startswith(key + "_") for v in prev_conditions):
            vals = [key + "_" + v for v in vals]
        group = set(vals)
        if group & seen:
            raise ValueError(f"Condition key '{key}' has duplicate values. You can't have duplicate values in your "
                             "condition keys.")
        seen |= group
        new = group - set(prev_conditions) - set(new_all)
        new_all += list(new)  # the reference appends list(set): order of a set, kept as produced here
        out_obs[key] = vals
    return out_obs, new_all


def load_query_model(reference, new_conditions=(), freeze=True, freeze_expression=True, remove_dropout=True,
                     device="cuda", precision="fp32", **kwargs):
    """_base.py:199-356 for the bare module.  reference: a checkpoint directory or a (model, attr_dict) pair.
    Builds a model whose condition list is the reference's plus `new_conditions`, loads the reference weights into
    the leading block of every tensor that grew, and applies the surgery freeze."""
    if isinstance(reference, (str, os.PathLike)):
        attr_dict, state, _ = load_params(reference)
    else:
        ref_model, attr_dict = reference
        state = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}
    kw = init_kwargs_from_attrs(attr_dict)
    kw["conditions"] = list(kw["conditions"]) + [c for c in new_conditions if c not in kw["conditions"]]
    if remove_dropout:
        kw["dr_rate"] = 0.0
    kw.update(kwargs)
    model = Celligner2(precision=precision, **kw)
    model.to(device)
    load_expand_params(model, state)
    if freeze:
        apply_freeze(model, freeze_expression)
    return model
