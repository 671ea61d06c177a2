"""The entry point that allocates nothing: c2_mmd_fwd_bwd_static (the MMD on the library's file-scope device scratch
array, include/c2mi.h) against c2_mmd_fwd_bwd with a caller workspace -- bit-identical losses and gradients, on two
calls in a row of which the second is SMALLER (stale contents of the shared array must not leak into it), for fp32 and
bf16 features, the row-sharded form, and against the CPU oracle (1e-4).  Sizes beyond the array are refused with a
message; torch's allocator is not touched between the calls (every tensor is created before the first one).

Collected last on purpose (the round's end runs pytest -x: a failure here must not hide other tests)."""
import numpy as np
import pytest
import torch

from oracle import c2_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _case(ops, B, D, C, seed, dtype=torch.float32):
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((B, D)).astype(np.float32)
    cond = np.zeros((B, C), np.int64)
    lab = rng.integers(0, C, B)
    lab[:C] = np.arange(C)
    cond[np.arange(B), lab] = 1
    feat = torch.tensor(f).to(DEV).to(dtype).contiguous()
    part = torch.empty(B, device=DEV, dtype=torch.int32)
    wt = torch.empty(64 * 64, device=DEV)
    st = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.partition(torch.tensor(cond).to(DEV), C, None, None, part, wt,
                  torch.empty(int(ops.lib.c2_partition_ws(B)), device=DEV, dtype=torch.uint8), st)
    assert st.item() == 0
    return dict(f=f, cond=cond, C=C, feat=feat, part=part, wt=wt, B=B, D=D,
                ws=torch.full((ops.mmd_ws(B, B, D),), float("nan"), device=DEV),
                loss_a=torch.zeros(1, device=DEV), loss_b=torch.zeros(1, device=DEV),
                d_a=torch.full((B, D), float("nan"), device=DEV), d_b=torch.full((B, D), float("nan"), device=DEV))


def _both(ops, c, beta=7.0):
    ops.mmd(c["feat"], c["part"], c["wt"], 0, c["B"], beta, c["loss_a"], c["d_a"], c["ws"])
    ops.mmd_static(c["feat"], c["part"], c["wt"], 0, c["B"], beta, c["loss_b"], c["d_b"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_static_scratch_mmd_equals_workspace_mmd_twice_second_smaller(dtype):
    from celligner2_amd import ops
    cap = ops.mmd_static_capacity()
    assert cap == 1 << 20
    big = _case(ops, 480, 64, 3, 1, dtype)     # fast four-wave kernel (fp32, D <= 64) / generic tile kernel (bf16)
    small = _case(ops, 72, 20, 4, 2, dtype)    # smaller in every dimension, other partition layout
    assert ops.mmd_ws(480, 480, 64) <= cap
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated()
    _both(ops, big)
    _both(ops, small)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == mem0
    for c in (big, small):
        assert torch.isfinite(c["d_b"]).all() and torch.isfinite(c["loss_b"]).all()
        assert torch.equal(c["loss_a"], c["loss_b"]), (c["loss_a"].item(), c["loss_b"].item())
        assert torch.equal(c["d_a"], c["d_b"])
        if dtype == torch.float32:  # and both are the reference's MMD (losses.py:127-249) to 1e-4
            m64, d64 = orc.mmd_loss(c["f"].astype(np.float64), c["cond"], c["C"], 7.0, None, None)
            assert abs(c["loss_b"].item() - m64) < 1e-4 * max(abs(m64), 1e-3)
            err = np.abs(c["d_b"].cpu().numpy() - d64).max() / np.abs(d64).max()
            assert err < 2e-4, err


def test_static_scratch_mmd_row_sharded_and_repeated():
    """The data-parallel form (own rows x all rows) and a repeat of the same call: same bits every time."""
    from celligner2_amd import ops
    c = _case(ops, 256, 32, 3, 5)
    h = 96
    la, lb = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    da, db = torch.empty(256 - h, 32, device=DEV), torch.empty(256 - h, 32, device=DEV)
    ws = torch.empty(ops.mmd_ws(256 - h, 256, 32), device=DEV)
    ops.mmd(c["feat"], c["part"], c["wt"], h, 256 - h, 3.0, la, da, ws)
    ops.mmd_static(c["feat"], c["part"], c["wt"], h, 256 - h, 3.0, lb, db)
    assert torch.equal(la, lb) and torch.equal(da, db)
    lb2, db2 = torch.zeros(1, device=DEV), torch.empty_like(db)
    ops.mmd_static(c["feat"], c["part"], c["wt"], h, 256 - h, 3.0, lb2, db2)
    assert torch.equal(lb, lb2) and torch.equal(db, db2)


def test_static_scratch_mmd_rejects_sizes_beyond_the_array():
    from celligner2_amd import ops
    from celligner2_amd._lib import C2Error
    B, D = 4096, 64
    assert ops.mmd_ws(B, B, D) > ops.mmd_static_capacity()
    feat = torch.zeros(B, D, device=DEV)
    part = torch.zeros(B, device=DEV, dtype=torch.int32)
    wt = torch.zeros(64 * 64, device=DEV)
    loss, d = torch.zeros(1, device=DEV), torch.zeros(B, D, device=DEV)
    with pytest.raises(C2Error, match="C2_MMD_STATIC_FLOATS"):
        ops.mmd_static(feat, part, wt, 0, B, 1.0, loss, d)
    assert loss.item() == 0.0  # nothing was launched


def test_static_scratch_mmd_sums_its_loss_itself_while_sums_are_deferred():
    """c2_defer_loss_sums leaves the loss kernels' second stages to the next combine launch -- but the static entry point's
    partials live in the shared array, which the next static call overwrites: it always launches its own sum."""
    from celligner2_amd import ops
    a, b = _case(ops, 200, 32, 3, 9), _case(ops, 136, 16, 3, 10)
    ref = []
    for c in (a, b):
        ops.mmd(c["feat"], c["part"], c["wt"], 0, c["B"], 2.0, c["loss_a"], c["d_a"], c["ws"])
        ref.append(c["loss_a"].clone())
    ops.defer_loss_sums(True)
    try:
        for c in (a, b):  # two static calls in a row, nothing combined in between
            ops.mmd_static(c["feat"], c["part"], c["wt"], 0, c["B"], 2.0, c["loss_b"], c["d_b"])
    finally:
        ops.defer_loss_sums(False)
        ops.drop_pending_sums()
    torch.cuda.synchronize()
    assert torch.equal(a["loss_b"], ref[0]) and torch.equal(b["loss_b"], ref[1])
    assert torch.equal(a["d_b"], a["d_a"]) and torch.equal(b["d_b"], b["d_a"])
