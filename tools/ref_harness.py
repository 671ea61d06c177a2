"""Loader for the reference's hot-path files (runs ONLY where /root/reference exists).

This is our own harness (it contains no reference code).  It registers empty
package shells so that no reference ``__init__`` runs, stubs the third-party
modules the image lacks (dgl, captum, anndata, scanpy) and then imports the
reference's model / losses / trainer files from where they lie.  It is used by
``tools/gen_golden.py`` to emit the fixtures under ``tests/golden/`` and by
``tools/validate_oracle.py``.  Nothing here ships to, or runs on, the GPU box.
"""
import importlib
import os
import sys
import types

REF = os.environ.get("C2_REFERENCE", "/root/reference")


def available():
    return os.path.isdir(os.path.join(REF, "celligner2", "model"))


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    m.__package__ = name
    sys.modules[name] = m
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class AnnDataLike:
    """Minimal AnnData stand-in: .X ndarray, .obs DataFrame, row slicing."""

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
        rows = idx[0] if isinstance(idx, tuple) else idx
        import numpy as np

        rows = np.asarray(rows)
        return AnnDataLike(self.X[rows], self.obs.iloc[rows].copy())

    def copy(self):
        return AnnDataLike(self.X.copy(), self.obs.copy())


def load():
    """Returns a namespace with Celligner2, losses, Celligner2Trainer, utils."""
    if not available():
        raise RuntimeError("reference tree not present")
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "")
    os.environ.setdefault("HIP_VISIBLE_DEVICES", "")
    root = os.path.join(REF, "celligner2")
    _pkg("celligner2", root)
    for sub in [
        "model",
        "othermodels",
        "othermodels.base",
        "utils",
        "trainers",
        "trainers.celligner2",
        "dataset",
        "dataset.celligner2",
    ]:
        _pkg("celligner2." + sub, os.path.join(root, *sub.split(".")))

    class _Dummy:
        def __init__(self, *a, **k):
            pass

    dgl = _stub("dgl")
    dgl.nn = _stub("dgl.nn", GraphConv=_Dummy)
    captum = _stub("captum")
    captum.attr = _stub("captum.attr", LRP=_Dummy, Saliency=_Dummy)
    _stub("anndata", AnnData=AnnDataLike, read=lambda *a, **k: None)
    _stub("scanpy", AnnData=AnnDataLike)

    ds = importlib.import_module("celligner2.dataset.celligner2.anndata")
    sys.modules["celligner2.dataset"].celligner2Dataset = ds.AnnotatedDataset
    sys.modules["celligner2.dataset"].remove_sparsity = lambda a: a

    ns = types.SimpleNamespace()
    ns.model = importlib.import_module("celligner2.model.celligner2")
    ns.Celligner2 = ns.model.Celligner2
    ns.losses = importlib.import_module("celligner2.model.losses")
    ns.model_utils = importlib.import_module("celligner2.model._utils")
    ns.modules = importlib.import_module("celligner2.model.modules")
    ns.trainer_mod = importlib.import_module(
        "celligner2.trainers.celligner2.semisupervised"
    )
    ns.Celligner2Trainer = ns.trainer_mod.Celligner2Trainer
    ns.trainer_utils = importlib.import_module("celligner2.trainers.celligner2._utils")
    ns.monitor = importlib.import_module("celligner2.utils.monitor")
    ns.AnnotatedDataset = ds.AnnotatedDataset
    ns.AnnDataLike = AnnDataLike
    return ns


def load_wrapper(ns=None):
    """Adds the reference's AnnData-level wrapper class to the namespace: ns.CELLIGNER2
    (celligner2/model/celligner2_model.py).  Two more harness patches, for that file only: the numpy-1.x private
    name it imports is given a stand-in, and the `pdb.set_trace()` left in `_get_user_attributes`
    (othermodels/base/_base.py:32-34) is a no-op."""
    ns = ns or load()
    import pdb

    import numpy.lib

    fb = sys.modules.get("numpy.lib.function_base") or types.ModuleType("numpy.lib.function_base")
    if not hasattr(fb, "_percentile_dispatcher"):
        fb._percentile_dispatcher = lambda *a, **k: None
    sys.modules["numpy.lib.function_base"] = fb
    numpy.lib.function_base = fb
    pdb.set_trace = lambda *a, **k: None
    ns.wrapper_mod = importlib.import_module("celligner2.model.celligner2_model")
    ns.CELLIGNER2 = ns.wrapper_mod.CELLIGNER2
    ns.base = importlib.import_module("celligner2.othermodels.base._base")
    return ns
