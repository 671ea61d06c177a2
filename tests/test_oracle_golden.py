"""Pins the CPU oracle against vectors produced by the reference itself
(tools/gen_golden.py).  Tolerances: the reference computes in fp32; the oracle is
run in fp64 and in fp32.  Bar (north_star): 1e-4 relative."""
import numpy as np
import pytest

from oracle import c2_oracle as orc
from tests import golden_util as gu

TOL = 1e-4


@pytest.mark.parametrize("name", gu.step_cases())
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_step_matches_reference(name, dtype):
    z, meta = gu.load(name)
    cfg = gu.cfg_from_meta(meta)
    params, buffers = gu.split_state(z, cfg)
    params = {k: v.astype(dtype) for k, v in params.items()}
    assert set(buffers) == set(orc.buffer_shapes(cfg))
    assert set(params) == set(orc.param_shapes(cfg)), "state_dict key set differs from the reference"
    for k, shp in orc.param_shapes(cfg).items():
        assert params[k].shape == shp
    batch = gu.group(z, "batch/")
    trainable = meta["trainable"]
    state = {}
    for s in range(meta["n_steps"]):
        masks = {k: v.astype(dtype) for k, v in gu.group(z, f"s{s}/mask/").items()}
        out = orc.forward_backward(params, cfg, batch, z[f"s{s}/eps"], masks,
                                   gscale=(1.0, meta["alpha"], 1.0, 1.0), buffers=buffers,
                                   bn_untracked=meta.get("bn_untracked", ()))
        buffers = out["buffers"]
        for k, b in gu.group(z, f"s{s}/buffer/").items():
            assert gu.relerr(buffers[k], b) < TOL, (k, s)
        ref_losses = z[f"s{s}/losses"]
        for i in range(4):
            assert abs(out["losses"][i] - ref_losses[i]) <= TOL * max(abs(ref_losses[i]), 1.0), (i, out["losses"], ref_losses)
        ref_grads = gu.group(z, f"s{s}/grad/")
        assert set(ref_grads) <= set(out["grads"])
        for k, g in ref_grads.items():
            assert gu.grad_err(k, out["grads"][k], g, cfg, ref_grads) < TOL, (k, gu.relerr(out["grads"][k], g))
        grads = {k: out["grads"][k] for k in ref_grads}  # only what autograd reached
        orc.adam_step(params, grads, state, names=[k for k in trainable if k in grads], **{
            "lr": meta["adam"]["lr"], "eps": meta["adam"]["eps"], "weight_decay": meta["adam"]["weight_decay"]})
        for k, p in gu.group(z, f"s{s}/param/").items():
            assert gu.relerr(params[k], p) < TOL, (k, s)


def test_nb_wide_row_op():
    z, _ = gu.load("nb_wide_row")
    for dtype in (np.float64, np.float32):
        recon, d_o, d_theta = orc.nb_loss(z["op/logits"].astype(dtype), z["op/x"].astype(dtype), z["op/goodloc"],
                                          z["op/sizefactor"].astype(dtype), z["op/theta"].astype(dtype), z["op/batch"])
        assert abs(recon - z["op/recon"][0]) < TOL * abs(z["op/recon"][0])
        assert gu.relerr(d_o, z["op/dlogits"]) < TOL
        assert gu.relerr(d_theta, z["op/dtheta"]) < TOL


def test_classes_mutated_in_place_by_reference():
    z, meta = gu.load("nb_basic")
    after = z["s0/classes_after"]
    assert (after != -1).all() and (z["batch/classes"] == -1).any()
