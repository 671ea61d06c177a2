"""Host side of the path against fixtures generated from the REFERENCE (tools/gen_golden_trainer.py): the input
pipeline (NaN mask + imputation, size factors, seeded split, multi-hot label encoding, stratifier weights), the
early-stopping monitor, surgery expansion / freeze, and -- where /root/reference exists -- the reference's own wrapper
class constructing this package's module.  No GPU: datasets are staged on the CPU device and nothing is launched."""
import os
import sys

import numpy as np
import pytest
import torch

from tests import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAINER_CASES = ["trainer_nb_stratified_earlystop", "trainer_nb_shuffled_dropout_clip"]


@pytest.mark.parametrize("name", TRAINER_CASES)
def test_make_dataset_matches_the_reference(name):
    """trainers/celligner2/_utils.py:79-185 + dataset/celligner2/anndata.py:28-75,137-166 + _utils.py:39-77."""
    from celligner2_amd.dataset import make_dataset
    z, meta = gu.load(name)
    adata = gu.adata_from(z)
    tk = meta["trainer_kwargs"]
    enc_c = {c: i for i, c in enumerate(meta["conditions"])}
    enc_p = {c: i for i, c in enumerate(meta["predictors"])}
    np.random.seed(meta["np_seed"])
    train, valid = make_dataset(adata, train_frac=tk["train_frac"], condition_keys=tk["condition_keys"],
                                cell_type_key=tk["cell_type_key"], condition_encoder=enc_c, predictor_keys=tk["predictor_keys"],
                                predictor_encoder=enc_p, min_weight=tk.get("min_weight", 0.2), device="cpu")
    assert len(train) == int(z["split/n_train"][0]) and len(valid) == int(z["split/n_valid"][0])
    assert np.array_equal(train.data.numpy(), z["split/train_x"])            # same rows, same imputed values (bitwise)
    assert np.array_equal(train.goodloc.numpy(), z["split/train_goodloc"])
    assert (~z["split/train_goodloc"]).any()
    assert np.array_equal(train.conditions.numpy(), z["split/train_conditions"])
    assert np.array_equal(train.predictors.numpy(), z["split/train_predictors"])
    assert (z["split/train_predictors"] == -1).any()                          # the "U" rows of the lineage column
    assert np.array_equal(train.size_factors.numpy().astype(np.float64), z["split/train_idx_sizefactors"])
    assert np.array_equal(valid.data.numpy().sum(1).astype(np.float64), z["split/valid_x_rowsum"])
    assert np.array_equal(train.stratifier_weights, z["split/stratifier_weights"])  # float64, bit for bit
    assert not np.isnan(adata.X).any()                                        # imputed in place, like the reference


def test_index_stream_of_the_loader_matches_the_reference():
    """The rows the reference's DataLoader (+ WeightedRandomSampler / shuffle) visits -- train and validation, every
    epoch -- reproduced by _Loader(rng="reference") when the other consumers of torch's CPU generator (one eps draw per
    loss() call) are replayed in between.  dr_rate = 0 case: eps is the only other draw."""
    from celligner2_amd.trainer import _Loader
    name = "trainer_nb_stratified_earlystop"
    z, meta = gu.load(name)
    calls = gu.calls_of(z)
    tk = meta["trainer_kwargs"]
    L = meta["model_kwargs"]["latent_dim"]

    class DS:
        device = torch.device("cpu")

        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def gather(self, idx, *a, **kw):
            return idx

    n_train, n_valid = int(z["split/n_train"][0]), int(z["split/n_valid"][0])
    bs = tk["batch_size"]
    train = _Loader(DS(n_train), bs, torch.tensor(z["split/stratifier_weights"]), n_train, "reference")
    valid = _Loader(DS(n_valid), min(bs, n_valid), None, None, "reference")
    torch.manual_seed(tk["seed"])
    it = iter(calls)
    for epoch in range(meta["epochs_run"]):
        for loader, flag in ((train, 1), (valid, 0)):
            for idx in loader:
                c = next(it)
                assert c["train"] == flag and c["epoch"] == epoch
                assert np.array_equal(idx.numpy(), c["rows"]), (epoch, flag)
                eps = torch.empty(idx.numel(), L).normal_()
                assert np.array_equal(eps.numpy(), c["eps"])
    assert next(it, None) is None


def test_early_stopping_monitor_replays_the_reference_run():
    from celligner2_amd.trainer import EarlyStopping
    z, meta = gu.load("trainer_nb_stratified_earlystop")
    es = EarlyStopping(**meta["trainer_kwargs"]["early_stopping_kwargs"])
    vals = z["logs/val_unweighted_loss"]
    lr, best, ran = meta["train_kwargs"]["lr"], None, 0
    for epoch, v in enumerate(vals):
        ran += 1
        if es.update_state(v):
            best = epoch
        go, decay = es.step(v)
        if decay:
            lr *= es.lr_factor
        if not go:
            break
    assert ran == meta["epochs_run"] == len(vals) and best == meta["best_epoch"]
    assert abs(lr - meta["lr_final"]) < 1e-12
    lrs = [c["lr"] for c in gu.calls_of(z) if c["train"]]
    assert lrs[0] == meta["train_kwargs"]["lr"] and abs(lrs[-1] - meta["lr_final"]) < 1e-12
    # mode="max", patience 0 (monitor off: a bare False, as the reference) and the plain-improvement path
    up = EarlyStopping(mode="max", patience=1, lr_patience=1, threshold=0.5)
    assert up.step(1.0) == (True, False) and up.best_performance == 1.0
    assert up.step(1.2) == (True, False) and up.wait == 1 and up.best_performance == 1.2   # gain 0.2 < threshold
    assert up.step(3.0) == (False, False)
    assert EarlyStopping(patience=0).step(1.0) is False


def test_surgery_expansion_matches_the_reference():
    """CELLIGNER2.load_query_data (celligner2_model.py:522-593 -> _base.py:199-356, 97-134): condition list, expanded
    state_dict (the new columns come from the same seed: same construction order as the reference module), trainable set,
    dr_rate = 0."""
    from celligner2_amd import surgery
    from celligner2_amd.model import Celligner2
    z, meta = gu.load("surgery_c2_to_c3")
    ra, na = meta["ref_attrs"], meta["new_attrs"]
    kw = surgery.init_kwargs_from_attrs(ra)
    ref = Celligner2(**kw)
    ref.load_state_dict({k: torch.tensor(v) for k, v in gu.group(z, "ref_state/").items()})
    query = gu.adata_from(z, "query/X", "query/obs/")
    _, new = surgery.new_conditions_from_obs(query.obs, ra["condition_keys_"], ra["conditions_"], miss=ra["miss_"])
    assert ra["conditions_"] + new == na["conditions_"]
    torch.manual_seed(meta["seed_new_model"])
    q = surgery.load_query_model((ref, ra), new_conditions=new, device="cpu")
    want = gu.group(z, "expanded/")
    got = q.state_dict()
    assert list(got) == list(want) or set(got) == set(want)
    for k, v in want.items():
        assert tuple(got[k].shape) == v.shape, k
        assert np.array_equal(got[k].numpy(), v), k
    assert [n for n, p in q.named_parameters() if p.requires_grad] == meta["trainable"]
    assert q.freeze and q.dr_rate == 0.0 == na["dr_rate_"] and q.conditions == na["conditions_"]
    # the wrapper's attribute dict, as far as the bare module determines it
    mine = surgery.public_attributes(q, condition_keys=na["condition_keys_"], predictor_keys=na["predictor_keys_"],
                                     miss=na["miss_"])
    for k, v in mine.items():
        # (condition_set_ / predictor_set_: the labels of the CURRENT AnnData's obs columns, celligner2_model.py:180-223 --
        # the AnnData's knowledge, not the module's)
        if k in na and k not in ("miss_", "mask_", "ext_mask_", "condition_set_", "predictor_set_"):
            assert na[k] == v or (isinstance(v, float) and abs(na[k] - v) < 1e-12), (k, na[k], v)


@pytest.mark.skipif(not os.path.isdir("/root/reference/celligner2/model"), reason="needs the reference tree (build container)")
def test_reference_wrapper_constructs_on_top_of_this_module():
    """The drop-in claim of INTEGRATION.md section 1, exercised with the reference's OWN wrapper class: CELLIGNER2
    (model/celligner2_model.py:98-336) builds `Celligner2` positionally and `.train` builds `Celligner2Trainer` by
    keyword.  With this package's two classes in their place, construction, the attribute dict and `load_query_data`
    (expansion + freeze on our module) work unchanged; `.train` gets as far as our trainer's first statement that needs
    the GPU (this container has none), i.e. every keyword the wrapper passes was accepted."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ref_harness
    ns = ref_harness.load_wrapper()
    from celligner2_amd.model import Celligner2
    from celligner2_amd.trainer import Celligner2Trainer
    z, meta = gu.load("surgery_c2_to_c3")
    wm = ns.wrapper_mod
    keep = wm.Celligner2, wm.Celligner2Trainer
    wm.Celligner2, wm.Celligner2Trainer = Celligner2, Celligner2Trainer
    try:
        ra = meta["ref_attrs"]
        ad = gu.adata_from(z, "ref/X", "ref/obs/")
        torch.manual_seed(5)
        w = ns.CELLIGNER2(ns.AnnDataLike(ad.X, ad.obs), hidden_layer_sizes=ra["hidden_layer_sizes_"], latent_dim=ra["latent_dim_"],
                          dr_rate=ra["dr_rate_"], condition_keys=ra["condition_keys_"], conditions=ra["conditions_"],
                          predictor_keys=ra["predictor_keys_"], predictors=ra["predictors_"], use_mmd=True, mmd_on="z",
                          beta=ra["beta_"], recon_loss="nb", classifier_hidden_layer_sizes=ra["classifier_hidden_layer_sizes_"])
        assert isinstance(w.model, Celligner2)
        want = gu.group(z, "ref_state/")
        assert {k: tuple(v.shape) for k, v in w.model.state_dict().items()} == {k: v.shape for k, v in want.items()}
        w.model.load_state_dict({k: torch.tensor(v) for k, v in want.items()})
        attrs = w._get_public_attributes()
        assert attrs["conditions_"] == ra["conditions_"] and attrs["input_dim_"] == ra["input_dim_"]
        q_ad = gu.adata_from(z, "query/X", "query/obs/")
        torch.manual_seed(meta["seed_new_model"])
        new = ns.CELLIGNER2.load_query_data(ns.AnnDataLike(q_ad.X, q_ad.obs), w)
        assert isinstance(new.model, Celligner2) and new.model.freeze
        for k, v in gu.group(z, "expanded/").items():
            assert np.array_equal(new.model.state_dict()[k].numpy(), v), k
        assert [n for n, p in new.model.named_parameters() if p.requires_grad] == meta["trainable"]
        if not torch.cuda.is_available():
            with pytest.raises(RuntimeError, match="MI355X"):
                new.train(n_epochs=1, lr=1e-3, eps=0.01, **meta["trainer_kwargs"])
    finally:
        wm.Celligner2, wm.Celligner2Trainer = keep
