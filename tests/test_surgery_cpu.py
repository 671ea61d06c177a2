"""Surgery bookkeeping and the reference's checkpoint directory (celligner2_amd/surgery.py; reference:
othermodels/base/_base.py:47-149, 199-356).  Host logic: runs without a GPU (the modules are only built and loaded,
never called)."""
import os
import pickle

import numpy as np
import pytest
import torch

from celligner2_amd import surgery
from celligner2_amd.model import Celligner2


def _model(conds, **kw):
    torch.manual_seed(1)
    return Celligner2(40, [16, 8], 4, 0.05, False, "nb", conditions=list(conds), predictors=["p0", "p1"],
                      use_mmd=True, beta=3.0, **kw)


def test_checkpoint_directory_round_trip(tmp_path):
    m = _model(["a", "b", "c"])
    attrs = surgery.public_attributes(m, condition_keys=["batch"], predictor_keys=["lineage"], miss=["nan"],
                                      is_trained=True)
    d = str(tmp_path / "ck")
    genes = [f"g{i}" for i in range(40)]
    surgery.save(d, m, attrs, genes)
    assert sorted(os.listdir(d)) == ["attr.pkl", "model_params.pt", "var_names.csv"]  # _base.py:85-87
    with pytest.raises(ValueError):
        surgery.save(d, m, attrs, genes)  # exists, overwrite=False (_base.py:71-78)
    surgery.save(d, m, attrs, genes, overwrite=True)
    attr2, state, names = surgery.load_params(d)
    assert attr2 == attrs and list(names) == genes
    assert set(state) == set(m.state_dict())
    m2, _, _ = surgery.load(d, device="cpu")
    assert not m2.training
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    assert m2.conditions == ["a", "b", "c"] and m2.use_mmd and m2.beta == 3.0 and m2.dr_rate == 0.05


def test_attr_pickle_is_loaded_without_executing_it(tmp_path):
    d = tmp_path / "ck"
    m = _model(["a"])
    surgery.save(str(d), m, surgery.public_attributes(m), ["g"] * 40)

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    with open(d / "attr.pkl", "wb") as f:
        pickle.dump({"x": Evil()}, f)
    with pytest.raises(pickle.UnpicklingError):
        surgery.load_params(str(d))


def test_expand_and_freeze():
    ref = _model(["a", "b", "c"])
    attrs = surgery.public_attributes(ref)
    q = surgery.load_query_model((ref, attrs), new_conditions=["d", "e", "a"], device="cpu")
    assert q.conditions == ["a", "b", "c", "d", "e"] and q.dr_rate == 0.0  # remove_dropout (_base.py:334-335)
    rs, qs = ref.state_dict(), q.state_dict()
    grown = {"theta", "encoder.FC.L0.cond_L.weight", "decoder.FirstL.L0.cond_L.weight"}
    for k in rs:
        if k in grown:
            assert qs[k].shape[1] == 5 and torch.equal(qs[k][:, :3], rs[k])       # loaded block (_base.py:117-124)
            assert qs[k][:, 3:].abs().sum() > 0                                    # new columns keep their init
        else:
            assert torch.equal(qs[k], rs[k]), k
    train = {n for n, p in q.named_parameters() if p.requires_grad}
    assert train == grown and q.freeze                                            # _base.py:342-351
    q2 = surgery.load_query_model((ref, attrs), new_conditions=["d"], freeze_expression=False, device="cpu")
    train2 = {n for n, p in q2.named_parameters() if p.requires_grad}
    assert train2 == {n for n, _ in q2.named_parameters() if "theta" in n or "L0" in n or "N0" in n}
    # a tensor may only grow (_base.py:112-116)
    small = _model(["a", "b"])
    with pytest.raises(ValueError, match="Can only increase"):
        surgery.load_expand_params(small, ref.state_dict())


def test_new_conditions_from_obs():
    obs = {"batch": ["x", "y", "x", "z"], "tech": ["t1", "t1", "t2", "t2"]}
    out, new = surgery.new_conditions_from_obs(obs, ["batch", "tech"], ["x", "t1"])
    assert sorted(new) == ["t2", "y", "z"] and out["batch"] == obs["batch"]
    # the reference model's conditions were renamed <key>_<value> (_base.py:253-267)
    out, new = surgery.new_conditions_from_obs(obs, ["batch"], ["batch_x", "batch_q"])
    assert out["batch"] == ["batch_x", "batch_y", "batch_x", "batch_z"] and sorted(new) == ["batch_y", "batch_z"]
    with pytest.raises(ValueError, match="duplicate"):
        surgery.new_conditions_from_obs({"a": ["v"], "b": ["v"]}, ["a", "b"], [])
    with pytest.raises(ValueError, match="missing"):
        surgery.new_conditions_from_obs({"a": ["v", "nan"]}, ["a"], [], miss=["nan"])


def test_checkpoint_directory_written_by_the_reference(tmp_path):
    """tests/golden/checkpoint_dir.npz holds the three files the reference's own BaseMixin.save wrote
    (othermodels/base/_base.py:47-95; tools/gen_golden_checkpoint.py, `pdb.set_trace` a no-op as in SURVEY.md App. C).
    The directory rebuilt from those bytes must load here with loaders that execute nothing from the files, rebuild the
    module with the same state, and `surgery.save` must write the same layout: the same attr.pkl key set and value types,
    var_names.csv byte for byte, the same state_dict keys / shapes / dtypes."""
    import json
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "checkpoint_dir.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    d = tmp_path / "ref_ck"
    os.makedirs(d)
    assert meta["files"] == ["attr.pkl", "model_params.pt", "var_names.csv"]
    for f in meta["files"]:
        with open(d / f, "wb") as fh:
            fh.write(bytes(z["file/" + f]))
    attrs, state, names = surgery.load_params(str(d))
    # what the fixture says the reference pickled (recorded as JSON when it was made): same keys, same types
    assert set(attrs) == set(meta["attrs"])
    for k, (tname, val) in meta["attrs"].items():
        assert type(attrs[k]).__name__ == tname, k
        if tname in ("bool", "int", "float", "str", "NoneType", "list"):
            assert attrs[k] == val, k
    assert attrs["condition_set_"] == {"dataset": {"ccle", "tcga"}} and attrs["miss_"] == "U"
    assert list(names) == meta["var_names"]
    assert [[k, list(v.shape), str(v.dtype)] for k, v in state.items()] == meta["state"]
    # the module rebuilt from the reference's attributes takes the reference's state_dict as it is
    m, attrs2, _ = surgery.load(str(d), device="cpu")
    assert list(m.state_dict()) == [k for k, _, _ in meta["state"]]
    for k, v in state.items():
        assert torch.equal(m.state_dict()[k], v), k
    assert (m.input_dim, m.hidden_layer_sizes, m.latent_dim, m.conditions, m.predictors) == \
        (40, [20, 10], 5, ["ccle", "tcga"], ["lung", "skin", "blood"])
    assert m.recon_loss == "nb" and m.use_mmd and m.beta == 7.0 and m.classifier_hidden_layer_sizes == [12]
    # ... and what surgery.save writes for it has the reference's layout
    mine = surgery.public_attributes(m, condition_keys=attrs["condition_keys_"], predictor_keys=attrs["predictor_keys_"],
                                     miss=attrs["miss_"], predictor_set=attrs["predictor_set_"])
    assert set(mine) == set(attrs)
    for k in attrs:
        assert type(mine[k]) is type(attrs[k]), k
        assert mine[k] == attrs[k], k
    d2 = tmp_path / "my_ck"
    surgery.save(str(d2), m, mine, names)
    assert sorted(os.listdir(d2)) == meta["files"]
    assert open(d2 / "var_names.csv", "rb").read() == bytes(z["file/var_names.csv"])
    attrs3, state3, _ = surgery.load_params(str(d2))
    assert attrs3 == attrs
    assert [[k, list(v.shape), str(v.dtype)] for k, v in state3.items()] == meta["state"]
