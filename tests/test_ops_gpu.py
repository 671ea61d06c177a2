"""Per-entry-point parity: HIP kernels (through the C-ABI) vs the CPU oracle, same seeded inputs.
Tolerances: fp32 1e-4 relative (north_star); bf16-operand GEMMs 1e-2."""
import numpy as np
import pytest
import torch

from oracle import c2_oracle as orc
from tests import golden_util as gu

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _ops():
    from celligner2_amd import ops
    return ops


def T(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


def rel(a, b):
    return gu.relerr(a.detach().float().cpu().numpy() if torch.is_tensor(a) else a, b)


@pytest.mark.parametrize("M,N,K", [(128, 64, 32), (128, 30, 64), (37, 33, 7), (1, 5, 3), (128, 128, 16)])
def test_tiny_layer_backward_in_one_launch(M, N, K):
    """c2_linear_bwd_tiny = c2_linear_dgrad + c2_linear_wgrad on tiny fp32 problems, bit for bit (strided operands too),
    and right against float64; a problem that is not tiny is refused."""
    ops = _ops()
    rng = np.random.default_rng(M + 3 * N + 5 * K)
    dy = T(rng.standard_normal((M, N + 3)).astype(np.float32))[:, :N]   # (row pitch > width)
    w = T(rng.standard_normal((N, K)).astype(np.float32))
    a = T(rng.standard_normal((M, K + 1)).astype(np.float32))[:, :K]
    assert ops.gemm_is_tiny(M, K, N) and ops.gemm_is_tiny(N, K, M)
    for alpha in (None, T(np.array([0.25], np.float32))):
        da0, dw0 = torch.empty(M, K, device=DEV), torch.empty(N, K, device=DEV)
        ops.linear_dgrad(dy, w, da0, alpha=alpha)
        ops.linear_wgrad(dy, a, dw0, alpha=alpha)
        da1, dw1 = torch.full((M, K), float("nan"), device=DEV), torch.full((N, K), float("nan"), device=DEV)
        ops.linear_bwd_tiny(dy, w, a, da1, dw1, alpha=alpha)
        assert torch.equal(da0, da1) and torch.equal(dw0, dw1)
        s = 1.0 if alpha is None else 0.25
        assert rel(da1, s * dy.double().cpu().numpy() @ w.double().cpu().numpy()) < 1e-5
        assert rel(dw1, s * dy.double().cpu().numpy().T @ a.double().cpu().numpy()) < 1e-5
    big = T(np.zeros((512, 512), np.float32))
    with pytest.raises(Exception):
        ops.linear_bwd_tiny(big, big, big, torch.empty(512, 512, device=DEV), torch.empty(512, 512, device=DEV))


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2), (torch.float16, 1e-3)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 72, 19000), (37, 300, 130), (4, 19000, 16), (512, 1000, 515)])
def test_linear_three_layouts(dtype, tol, M, N, K):
    ops = _ops()
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ta, tw, tdy = T(a, dtype), T(w, dtype), T(dy, dtype)
    # the oracle sees the operands the kernel sees (bf16-rounded in bf16 mode)
    a64, w64, dy64 = (t.float().cpu().numpy().astype(np.float64) for t in (ta, tw, tdy))
    y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(ta, tw, y, bias=T(bias))
    assert rel(y, a64 @ w64.T + bias) < tol
    for ns in (2, 3):
        ys = torch.empty(ns, M, N, device=DEV)
        ops.linear_fwd(ta, tw, ys, nslab=ns)
        assert rel(ys.sum(0), a64 @ w64.T) < tol
    alpha = T(np.array([0.5], np.float32))
    da = torch.empty(M, K, device=DEV)
    ops.linear_dgrad(tdy, tw, da, alpha=alpha)
    assert rel(da, 0.5 * dy64 @ w64) < tol
    das = torch.empty(2, M, K, device=DEV)
    ops.linear_dgrad(tdy, tw, das, nslab=2)
    assert rel(das.sum(0), dy64 @ w64) < tol
    dw = torch.empty(N, K, device=DEV)
    ops.linear_wgrad(tdy, ta, dw)
    assert rel(dw, dy64.T @ a64) < tol
    dws = torch.empty(3, N, K, device=DEV)
    ops.linear_wgrad(tdy, ta, dws, nslab=3)
    out = torch.empty(N, K, device=DEV)
    ops.sum_slabs(dws, out)
    assert rel(out, dy64.T @ a64) < tol


@pytest.mark.parametrize("M,N,K", [(1496, 1800, 1416), (2048, 1024, 2048), (1000, 2304, 4096 + 64 + 8), (4096, 19000, 520)])
def test_linear_wide_bf16(M, N, K):
    """Shapes that take the 256x256 LDS-DMA kernel (gemm_wide.hip): ragged tile edges, reduction tails that are
    not a multiple of 64, split reductions.  The oracle sees the bf16-rounded operands, so only the fp32
    accumulation order differs: 2e-5."""
    ops = _ops()
    assert ops.gemm_tile(torch.bfloat16, M, N, K) == 256
    tol = 2e-5
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    ta = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    tw = torch.randn(N, K, generator=g).to(DEV).bfloat16()
    tdy = torch.randn(M, N, generator=g).to(DEV).bfloat16()
    bias = torch.randn(N, generator=g).to(DEV)
    a64, w64, dy64 = ta.double(), tw.double(), tdy.double()
    relt = lambda got, want: float((got.double() - want).abs().max() / want.abs().max())  # noqa: E731
    y = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(ta, tw, y, bias=bias)
    assert relt(y, a64 @ w64.T + bias.double()) < tol
    for ns in (2, 5):
        ys = torch.full((ns, M, N), float("nan"), device=DEV)
        ops.linear_fwd(ta, tw, ys, nslab=ns)
        assert relt(ys.sum(0), a64 @ w64.T) < tol
    alpha = torch.tensor([0.5], device=DEV)
    if ops.gemm_tile(torch.bfloat16, M, K, N) == 256:
        da = torch.full((M, K), float("nan"), device=DEV)
        ops.linear_dgrad(tdy, tw, da, alpha=alpha)
        assert relt(da, 0.5 * dy64 @ w64) < tol
        das = torch.full((3, M, K), float("nan"), device=DEV)
        ops.linear_dgrad(tdy, tw, das, nslab=3)
        assert relt(das.sum(0), dy64 @ w64) < tol
    if ops.gemm_tile(torch.bfloat16, N, K, M) == 256:
        dw = torch.full((N, K), float("nan"), device=DEV)
        ops.linear_wgrad(tdy, ta, dw, alpha=alpha)
        assert relt(dw, 0.5 * dy64.T @ a64) < tol
        dws = torch.full((2, N, K), float("nan"), device=DEV)
        ops.linear_wgrad(tdy, ta, dws, nslab=2)
        assert relt(dws.sum(0), dy64.T @ a64) < tol


def test_linear_unaligned_views():
    """Operands with odd leading dimensions / offsets take the scalar-load path."""
    ops = _ops()
    rng = np.random.default_rng(5)
    M, N, K = 70, 50, 45
    abig = T(rng.standard_normal((M, K + 3)).astype(np.float32))
    wbig = T(rng.standard_normal((N, K + 1)).astype(np.float32))
    a, w = abig[:, 1:K + 1], wbig[:, :K]
    y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(a, w, y)
    assert rel(y, a.cpu().double().numpy() @ w.cpu().double().numpy().T) < 1e-5


@pytest.mark.parametrize("H,C,p,use_ln", [(64, 3, 0.0, 1), (1024, 2, 0.1, 1), (1500, 0, 0.05, 1), (96, 4, 0.0, 0),
                                          (512, 3, 0.05, 1), (30, 2, 0.0, 1)])
@pytest.mark.parametrize("act_dtype", [torch.float32, torch.bfloat16])
def test_epilogue_fwd_bwd(H, C, p, use_ln, act_dtype):
    ops = _ops()
    rng = np.random.default_rng(H + C)
    B, ns = 37, 2
    y = rng.standard_normal((ns, B, H)).astype(np.float32)
    bias = rng.standard_normal(H).astype(np.float32)
    cond = np.zeros((B, max(C, 1)), np.int64)
    cond[np.arange(B), rng.integers(0, max(C, 1), B)] = 1
    cond[0, :] = 1  # a multi-hot row
    wc = rng.standard_normal((H, max(C, 1))).astype(np.float32)
    mask = (rng.uniform(size=(B, H)) > p).astype(np.uint8) if p > 0 else None
    dout = rng.standard_normal((2, B, H)).astype(np.float32)
    extra = rng.standard_normal((B, H)).astype(np.float32)
    # oracle: a = identity input so that h = Y; weights = identity is wasteful, use the block functions directly
    h_in = y.sum(0).astype(np.float64)
    out64, cache = orc._block_fwd(h_in, np.eye(H), bias.astype(np.float64), cond[:, :C].astype(np.float64) if C else None,
                                  wc[:, :C].astype(np.float64) if C else None,
                                  mask.astype(np.float64) if mask is not None else None, p, bool(use_ln))
    act = torch.empty(B, H, device=DEV, dtype=act_dtype)
    yhat = torch.empty(B, H, device=DEV)
    rstd = torch.empty(B, device=DEV)
    want_pl = act_dtype == torch.float32 and H % 4 == 0 and H <= 1024 and (B * H) % 4 == 0
    pl = torch.zeros(2, B, H, device=DEV, dtype=torch.bfloat16) if want_pl else None
    ops.epilogue_fwd(T(y), T(bias), T(cond[:, :C]) if C else None, T(wc[:, :C].copy()) if C else None,
                     T(mask) if mask is not None else None, p, use_ln, act, yhat, rstd, planes=pl)
    if want_pl:  # the planes are the split of the fp32 activation, bit for bit
        assert torch.equal(pl, ops.split_bf16(act))
    tol = 1e-4 if act_dtype == torch.float32 else 1e-2
    assert rel(act, out64) < tol
    assert rel(yhat, cache["yhat"]) < 1e-4
    # backward: dout = sum of 2 slabs + 0.25*extra
    dtot = dout.sum(0).astype(np.float64) + 0.25 * extra
    yh, rs, mk = cache["yhat"], cache["rstd"], cache["mask"]
    d = dtot * (mk / (1 - p) if mk is not None else 1.0) * (yh > 0)
    dh64 = (d - d.mean(1, keepdims=True) - yh * (d * yh).mean(1, keepdims=True)) * rs if use_ln else d
    dh = torch.empty(B, H, device=DEV, dtype=act_dtype)
    pl2 = torch.zeros(2, B, H, device=DEV, dtype=torch.bfloat16) if want_pl else None
    ops.epilogue_bwd(T(dout), T(mask) if mask is not None else None, p, use_ln, yhat, rstd, dh, extra=T(extra),
                     extra_scale=T(np.array([0.25], np.float32)), planes=pl2)
    if want_pl:
        assert torch.equal(pl2, ops.split_bf16(dh))
    assert rel(dh, dh64) < (2e-4 if act_dtype == torch.float32 else 1e-2)
    # bias / cond grads from the kernel's own dh
    dh_np = dh.float().cpu().numpy().astype(np.float64)
    ws = torch.empty(ops.bias_cond_grad_ws(B, H, C), device=DEV)
    dbt = torch.empty(H, device=DEV)
    dwct = torch.empty(H, C, device=DEV) if C else None
    ops.bias_cond_grad(dh, T(cond[:, :C]) if C else None, dbt, dwct, ws)
    assert rel(dbt, dh_np.sum(0)) < 1e-5
    if C:
        assert rel(dwct, dh_np.T @ cond[:, :C]) < 1e-5


@pytest.mark.parametrize("B,H,C,p", [(37, 64, 3, 0.0), (200, 1024, 2, 0.1), (65, 50, 0, 0.05), (1000, 512, 3, 0.0),
                                      (33, 1500, 2, 0.0)])
@pytest.mark.parametrize("act_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nrank", [1, 2])
def test_batchnorm_tail(B, H, C, p, act_dtype, nrank):
    """BatchNorm1d tail (c2_bn_stats / c2_bn_fwd / c2_bn_bwd_reduce / c2_bn_bwd) vs the oracle's block functions;
    nrank = 2 runs it the way two data-parallel ranks do (row halves, partials stacked, sums added) and must give
    the statistics and gradients of the whole batch.  Ragged row blocks, scalar path (H = 50), slabs, extra."""
    ops = _ops()
    rng = np.random.default_rng(B + H + C)
    ns = 2
    y = (rng.standard_normal((ns, B, H)) * 2 + 0.7).astype(np.float32)
    bias = rng.standard_normal(H).astype(np.float32)
    cond = np.zeros((B, max(C, 1)), np.int64)
    cond[np.arange(B), rng.integers(0, max(C, 1), B)] = 1
    cond[0, :] = 1
    wc = rng.standard_normal((H, max(C, 1))).astype(np.float32)
    gamma = (1 + 0.3 * rng.standard_normal(H)).astype(np.float32)
    beta = (0.2 * rng.standard_normal(H)).astype(np.float32)
    rm0 = (0.1 * rng.standard_normal(H)).astype(np.float32)
    rv0 = (1 + 0.3 * rng.uniform(size=H)).astype(np.float32)
    mask = (rng.uniform(size=(B, H)) > p).astype(np.uint8) if p > 0 else None
    dout = rng.standard_normal((2, B, H)).astype(np.float32)
    extra = rng.standard_normal((B, H)).astype(np.float32)
    f8 = lambda v: v.astype(np.float64)  # noqa: E731
    bn = dict(weight=f8(gamma), bias=f8(beta), running_mean=f8(rm0), running_var=f8(rv0), batch_stats=True, track=True)
    out64, cache = orc._block_fwd(f8(y.sum(0)), np.eye(H), f8(bias), f8(cond[:, :C]) if C else None,
                                  f8(wc[:, :C]) if C else None, f8(mask) if mask is not None else None, p, True, bn)
    dtot = f8(dout.sum(0)) + 0.25 * extra
    da64, _, _, _, dgam64, dbet64 = orc._block_bwd(dtot, cache, np.eye(H), True, C > 0)  # identity weight: da = dh
    if nrank == 2 and B % 2:
        pytest.skip("ranks hold equal shares")
    cuts = [0, B] if nrank == 1 else [0, B // 2, B]
    Bs = [b - a for a, b in zip(cuts[:-1], cuts[1:])]
    rm = [T(rm0.copy()) for _ in Bs]
    rv = [T(rv0.copy()) for _ in Bs]
    nbt = [torch.zeros((), dtype=torch.int64, device=DEV) for _ in Bs]
    hs, parts, acts, stats, pls = [], [], [], [], []
    want_pl = act_dtype == torch.float32 and H % 4 == 0
    for r, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        n = b - a
        h = torch.empty(n, H, device=DEV)
        part = torch.empty(2, H, device=DEV)
        ops.bn_stats(T(y[:, a:b].copy()), T(bias), T(cond[a:b, :C].copy()) if C else None,
                     T(wc[:, :C].copy()) if C else None, h, part, torch.empty(ops.bn_ws(n, H), device=DEV))
        hs.append(h)
        parts.append(part)
    allparts = torch.stack(parts).contiguous()
    for r, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        n = b - a
        act = torch.empty(n, H, device=DEV, dtype=act_dtype)
        st = torch.empty(2, H, device=DEV)
        pl = torch.zeros(2, n, H, device=DEV, dtype=torch.bfloat16) if want_pl else None
        ops.bn_fwd(hs[r], allparts, Bs[0], B, T(gamma), T(beta), rm[r], rv[r], nbt[r], 0.1, True,
                   T(mask[a:b].copy()) if mask is not None else None, p, act, st, planes=pl)
        if want_pl:
            assert torch.equal(pl, ops.split_bf16(act))
        acts.append(act)
        stats.append(st)
        assert nbt[r].item() == 1
        assert rel(rm[r], bn["new_running"][0]) < 1e-5 and rel(rv[r], bn["new_running"][1]) < 1e-5
    tol = 1e-4 if act_dtype == torch.float32 else 1e-2
    assert rel(torch.cat(acts).float(), out64) < tol
    assert rel(torch.cat(hs), cache["yhat"]) < 1e-4
    # backward
    dys, sums, dgs, dbs = [], [], [], []
    for r, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
        n = b - a
        dy = torch.empty(n, H, device=DEV)
        sm = torch.empty(2, H, device=DEV)
        dg, db = torch.empty(H, device=DEV), torch.empty(H, device=DEV)
        ops.bn_bwd_reduce(T(dout[:, a:b].copy()), T(mask[a:b].copy()) if mask is not None else None, p, hs[r], T(gamma),
                          T(beta), dy, sm, dg, db, torch.empty(ops.bn_ws(n, H), device=DEV), extra=T(extra[a:b].copy()),
                          extra_scale=T(np.array([0.25], np.float32)))
        dys.append(dy), sums.append(sm), dgs.append(dg), dbs.append(db)
    tot = torch.stack(sums).sum(0)  # the all-reduce
    assert rel(torch.stack(dgs).sum(0), dgam64) < 1e-4 and rel(torch.stack(dbs).sum(0), dbet64) < 1e-4
    dhs = []
    for r, n in enumerate(Bs):
        dh = torch.empty(n, H, device=DEV, dtype=act_dtype)
        pl = torch.zeros(2, n, H, device=DEV, dtype=torch.bfloat16) if want_pl else None
        ops.bn_bwd(dys[r], hs[r], tot, B, True, T(gamma), stats[r], dh, planes=pl)
        if want_pl:
            assert torch.equal(pl, ops.split_bf16(dh))
        dhs.append(dh)
    assert rel(torch.cat(dhs).float(), da64) < (2e-4 if act_dtype == torch.float32 else 1e-2)
    # eval mode: the running statistics normalise, nothing moves; backward without the mean terms
    bn_e = dict(weight=f8(gamma), bias=f8(beta), running_mean=f8(rm0), running_var=f8(rv0), batch_stats=False)
    out_e, cache_e = orc._block_fwd(f8(y.sum(0)), np.eye(H), f8(bias), f8(cond[:, :C]) if C else None,
                                    f8(wc[:, :C]) if C else None, None, 0.0, True, bn_e)
    da_e = orc._block_bwd(dtot, cache_e, np.eye(H), True, C > 0)[0]
    h = torch.empty(B, H, device=DEV)
    part = torch.empty(2, H, device=DEV)
    ws = torch.empty(ops.bn_ws(B, H), device=DEV)
    ops.bn_stats(T(y), T(bias), T(cond[:, :C].copy()) if C else None, T(wc[:, :C].copy()) if C else None, h, part, ws)
    rme, rve = T(rm0.copy()), T(rv0.copy())
    act = torch.empty(B, H, device=DEV, dtype=act_dtype)
    st = torch.empty(2, H, device=DEV)
    ops.bn_fwd(h, None, B, B, T(gamma), T(beta), rme, rve, nbt[0], 0.1, False, None, 0.0, act, st)
    assert rel(act.float(), out_e) < tol and torch.equal(rme, T(rm0)) and nbt[0].item() == 1
    dy, sm = torch.empty(B, H, device=DEV), torch.empty(2, H, device=DEV)
    ops.bn_bwd_reduce(T(dout), None, 0.0, h, T(gamma), T(beta), dy, sm, None, None, ws, extra=T(extra),
                      extra_scale=T(np.array([0.25], np.float32)))
    dh = torch.empty(B, H, device=DEV, dtype=act_dtype)
    ops.bn_bwd(dy, h, sm, B, False, T(gamma), st, dh)
    assert rel(dh.float(), da_e) < (2e-4 if act_dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("B,H,C", [(1000, 515, 11), (300, 19000, 0), (129, 77, 5), (4096, 1024, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bias_cond_grad(B, H, C, dtype):
    """Column sums over many row chunks, more condition columns than one pass holds (8), ragged edges."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(B + H + C)
    dh = torch.randn(B, H, generator=g).to(DEV).to(dtype)
    cond = (torch.rand(B, max(C, 1), generator=g) < 0.4).long().to(DEV)[:, :C].contiguous()
    alpha = torch.tensor([0.7], device=DEV)
    ws = torch.empty(ops.bias_cond_grad_ws(B, H, C), device=DEV)
    db = torch.full((H,), float("nan"), device=DEV)
    dwc = torch.full((H, C), float("nan"), device=DEV) if C else None
    ops.bias_cond_grad(dh, cond if C else None, db, dwc, ws, alpha=alpha)
    d64 = dh.double()
    want_b = 0.7 * d64.sum(0)
    assert float((db.double() - want_b).abs().max() / want_b.abs().max()) < 1e-5
    if C:
        want_c = 0.7 * d64.T @ cond.double()
        assert float((dwc.double() - want_c).abs().max() / want_c.abs().max()) < 1e-5


@pytest.mark.parametrize("use_own_kl", [0, 1])
def test_reparam_kl(use_own_kl):
    ops = _ops()
    rng = np.random.default_rng(3)
    B, Ld = 130, 10
    mu, lv, eps = (rng.standard_normal((B, Ld)).astype(np.float32) for _ in range(3))
    z64 = orc.reparam(mu.astype(np.float64), lv.astype(np.float64), eps.astype(np.float64))
    kl64, dmu_kl, dlv_kl = orc.kl_loss(mu.astype(np.float64), lv.astype(np.float64), bool(use_own_kl))
    z32 = torch.empty(B, Ld, device=DEV)
    zt = torch.empty(B, Ld, device=DEV, dtype=torch.bfloat16)
    kl = torch.zeros(1, device=DEV)
    ws = torch.empty(int(ops.lib.c2_reduce_ws()), device=DEV)
    ops.reparam_kl_fwd(T(mu), T(lv), T(eps), 1.0 / B, use_own_kl, z32, zt, kl, ws)
    assert rel(z32, z64) < 1e-5 and rel(zt, z64) < 1e-2
    assert abs(kl.item() - kl64) < 1e-4 * abs(kl64)
    dz = rng.standard_normal((2, B, Ld)).astype(np.float32)
    gk = 0.3
    dzt = dz.sum(0).astype(np.float64)
    v = np.exp(lv.astype(np.float64)) + 1e-4
    dmu64 = dzt + gk * dmu_kl
    dlv64 = dzt * eps * np.exp(lv.astype(np.float64)) / (2 * np.sqrt(v)) + gk * dlv_kl
    dmu = torch.empty(B, Ld, device=DEV)
    dlv = torch.empty(B, Ld, device=DEV)
    ops.reparam_kl_bwd(T(dz), T(mu), T(lv), T(eps), 1.0 / B, use_own_kl, T(np.array([gk], np.float32)), dmu, dlv)
    assert rel(dmu, dmu64) < 1e-5 and rel(dlv, dlv64) < 1e-5


def _nb_inputs(rng, B, G, C, multi=False, big_counts=False):
    o = rng.standard_normal((B, G)).astype(np.float32) * 2
    depth = rng.uniform(0.5, 2, (B, 1))
    gene = np.exp(rng.normal(1.0 if not big_counts else 5.0, 1.5, (1, G)))
    x = rng.poisson(depth * gene).astype(np.float32)
    good = rng.uniform(size=(B, G)) > 0.01
    cond = np.zeros((B, C), np.int64)
    cond[np.arange(B), rng.integers(0, C, B)] = 1
    if multi:
        cond[0, :] = 1      # pattern "only ones" when C == 2
        cond[1, :] = 0      # pattern "only zeros"
    theta = rng.standard_normal((G, C)).astype(np.float32)
    return o, x, good, x.sum(1).astype(np.float32), theta, cond


@pytest.mark.parametrize("B,G,C,multi", [(5, 96, 2, True), (9, 1000, 3, False), (3, 3001, 2, False), (6, 5000, 2, True),
                                         (4, 19000, 2, False), (3, 25000, 4, False)])
@pytest.mark.parametrize("dl_dtype", [torch.float32, torch.bfloat16])
def test_recon_nb(B, G, C, multi, dl_dtype):
    ops = _ops()
    rng = np.random.default_rng(G + B)
    o, x, good, sf, theta, cond = _nb_inputs(rng, B, G, C, multi, big_counts=(G == 5000))
    recon64, do64, dth64 = orc.nb_loss(o.astype(np.float64), x.astype(np.float64), good, sf.astype(np.float64),
                                       theta.astype(np.float64), cond)
    loss = torch.zeros(1, device=DEV)
    dl = torch.empty(B, G, device=DEV, dtype=dl_dtype)
    dth = torch.empty(G, C, device=DEV)
    ws = torch.empty(ops.recon_nb_ws(B, G, C), device=DEV)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.recon_nb(T(o), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss, dl, dth, ws, status)
    assert status.item() == 0
    assert abs(loss.item() - recon64) < 1e-4 * abs(recon64)
    assert rel(dl, do64) < (1e-4 if dl_dtype == torch.float32 else 1e-2)
    assert rel(dth, dth64) < 1e-4
    # same call with the data-only term sum goodloc*lgamma(x+1) handed in (resident-dataset path)
    from scipy.special import gammaln
    lgx = ops.nb_row_const(T(x), T(good))
    want = (gammaln(x.astype(np.float64) + 1.0) * good).sum(1)
    assert rel(lgx, want) < 1e-5
    loss2 = torch.zeros(1, device=DEV)
    dl2 = torch.empty(B, G, device=DEV, dtype=dl_dtype)
    dth2 = torch.empty(G, C, device=DEV)
    db2 = torch.empty(G, device=DEV)
    ops.recon_nb(T(o), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss2, dl2, dth2, ws, status, lgx1=lgx, dbias=db2)
    assert rel(db2, do64.sum(0)) < (1e-4 if dl_dtype == torch.float32 else 1e-2)
    assert abs(loss2.item() - recon64) < 1e-4 * abs(recon64)
    assert rel(dl2, do64) < (1e-4 if dl_dtype == torch.float32 else 1e-2)
    assert rel(dth2, dth64) < 1e-4
    # same batch read through a row index from a larger "resident" matrix (rows permuted, padded with decoys)
    perm = rng.permutation(B + 3)[:B]
    xf = rng.poisson(3.0, (B + 3, G)).astype(np.float32)
    gf = rng.uniform(size=(B + 3, G)) > 0.5
    xf[perm], gf[perm] = x, good
    loss3 = torch.zeros(1, device=DEV)
    dl3 = torch.empty(B, G, device=DEV, dtype=dl_dtype)
    dth3 = torch.empty(G, C, device=DEV)
    ops.recon_nb(T(o), T(xf), T(gf), T(sf), T(theta), T(cond), 1.0 / B, loss3, dl3, dth3, ws, status, lgx1=lgx,
                 row_index=T(perm.astype(np.int64)))
    assert abs(loss3.item() - loss2.item()) <= 1e-6 * abs(recon64)
    assert torch.equal(dl3, dl2) and torch.equal(dth3, dth2)


@pytest.mark.parametrize("regt", ["0", "1"])
def test_recon_nb_row_walks_at_19k_genes(regt, monkeypatch):
    """Both forms of the fast kernel at the benchmark width (C2_NB_REGT: theta and the accumulator in registers, or theta
    re-read per row), several rows per workgroup, the row pattern changing inside a workgroup's walk (the accumulator
    and the kept-entry counts are flushed per pattern) and a tenth of the entries masked."""
    monkeypatch.setenv("C2_NB_REGT", regt)
    ops = _ops()
    rng = np.random.default_rng(19)
    B, G, C = 600, 19000, 2
    o, x, good, sf, theta, cond = _nb_inputs(rng, B, G, C, multi=True)
    good &= rng.uniform(size=(B, G)) > 0.1
    cond[rng.integers(0, B, 40), :] = 1   # pattern "only ones" here and there: blocks see 2 -> 1 -> 2
    cond[rng.integers(0, B, 40), :] = 0   # and "only zeros"
    recon64, do64, dth64 = orc.nb_loss(o.astype(np.float64), x.astype(np.float64), good, sf.astype(np.float64),
                                       theta.astype(np.float64), cond)
    for dl_dtype in (torch.float32, torch.bfloat16):
        loss = torch.zeros(1, device=DEV)
        dl = torch.empty(B, G, device=DEV, dtype=dl_dtype)
        dth = torch.empty(G, C, device=DEV)
        status = torch.zeros(1, device=DEV, dtype=torch.int32)
        db = torch.empty(G, device=DEV)
        ops.recon_nb(T(o), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss, dl, dth,
                     torch.empty(ops.recon_nb_ws(B, G, C), device=DEV), status, dbias=db)
        assert status.item() == 0
        assert abs(loss.item() - recon64) < 1e-4 * abs(recon64)
        assert rel(dl, do64) < (1e-4 if dl_dtype == torch.float32 else 1e-2)
        assert rel(dth, dth64) < 1e-4
        # the column sums of dlogits (bias gradient of the layer above): in-kernel (regt) or by the column-sum pass
        assert rel(db, do64.sum(0)) < (1e-4 if (dl_dtype == torch.float32 or regt == "1") else 1e-2)


def test_recon_nb_more_than_255_rows_per_workgroup():
    """The kept-entry counts are bytes: a workgroup that walks more than 255 rows of one pattern flushes in between."""
    ops = _ops()
    rng = np.random.default_rng(7)
    B, G, C = 256 * 300, 96, 3
    o, x, good, sf, theta, cond = _nb_inputs(rng, B, G, C)
    recon64, do64, dth64 = orc.nb_loss(o.astype(np.float64), x.astype(np.float64), good, sf.astype(np.float64),
                                       theta.astype(np.float64), cond)
    loss = torch.zeros(1, device=DEV)
    dl = torch.empty(B, G, device=DEV)
    dth = torch.empty(G, C, device=DEV)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.recon_nb(T(o), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss, dl, dth,
                 torch.empty(ops.recon_nb_ws(B, G, C), device=DEV), status)
    assert status.item() == 0
    assert abs(loss.item() - recon64) < 1e-4 * abs(recon64)
    assert rel(dl, do64) < 1e-4 and rel(dth, dth64) < 1e-4


def test_recon_nb_flags_a_dispersion_below_its_domain():
    """exp(theta) < 1e-17 is outside the fast kernel's arithmetic (nb_pair): reported through the status word (2)."""
    ops = _ops()
    rng = np.random.default_rng(1)
    o, x, good, sf, theta, cond = _nb_inputs(rng, 4, 64, 2)
    theta[5, 1] = -45.0
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.recon_nb(T(o), T(x), T(good), T(sf), T(theta), T(cond), 0.25, torch.zeros(1, device=DEV),
                 torch.empty(4, 64, device=DEV), torch.empty(64, 2, device=DEV),
                 torch.empty(ops.recon_nb_ws(4, 64, 2), device=DEV), status)
    assert status.item() == 2


def test_recon_nb_golden_wide_row():
    """The [4,19000] NB vector produced by the reference itself."""
    ops = _ops()
    z, _ = gu.load("nb_wide_row")
    B, G = z["op/x"].shape
    C = z["op/theta"].shape[1]
    loss = torch.zeros(1, device=DEV)
    dl = torch.empty(B, G, device=DEV)
    dth = torch.empty(G, C, device=DEV)
    ws = torch.empty(ops.recon_nb_ws(B, G, C), device=DEV)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.recon_nb(T(z["op/logits"]), T(z["op/x"]), T(z["op/goodloc"]), T(z["op/sizefactor"]), T(z["op/theta"]),
                 T(z["op/batch"]), 1.0 / B, loss, dl, dth, ws, status)
    assert abs(loss.item() - z["op/recon"][0]) < 1e-4 * abs(z["op/recon"][0])
    assert rel(dl, z["op/dlogits"]) < 1e-4
    assert rel(dth, z["op/dtheta"]) < 1e-4


def test_recon_nb_flags_bad_condition_values():
    ops = _ops()
    rng = np.random.default_rng(0)
    o, x, good, sf, theta, cond = _nb_inputs(rng, 4, 64, 2)
    cond[2, 0] = -1
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.recon_nb(T(o), T(x), T(good), T(sf), T(theta), T(cond), 0.25, torch.zeros(1, device=DEV),
                 torch.empty(4, 64, device=DEV), torch.empty(64, 2, device=DEV),
                 torch.empty(ops.recon_nb_ws(4, 64, 2), device=DEV), status)
    assert status.item() != 0


def test_recon_mse():
    ops = _ops()
    rng = np.random.default_rng(1)
    B, G = 17, 333
    o = rng.standard_normal((B, G)).astype(np.float32)
    x = np.abs(rng.standard_normal((B, G))).astype(np.float32)
    good = rng.uniform(size=(B, G)) > 0.05
    r64, d64 = orc.mse_loss(o.astype(np.float64), x.astype(np.float64), good)
    loss = torch.zeros(1, device=DEV)
    dl = torch.empty(B, G, device=DEV)
    ws = torch.empty(int(ops.lib.c2_reduce_ws()), device=DEV)
    ops.recon_mse(T(o), T(x), T(good), 1.0 / B, loss, dl, ws)
    assert abs(loss.item() - r64) < 1e-5 * abs(r64) and rel(dl, d64) < 1e-5


@pytest.mark.parametrize("P", [1, 5, 30, 100])
def test_classifier_loss(P):
    ops = _ops()
    rng = np.random.default_rng(P)
    B = 77
    u = (rng.standard_normal((B, P)) * 3).astype(np.float32)
    y = np.zeros((B, P), np.int64)
    y[np.arange(B), rng.integers(0, P, B)] = 1
    y[rng.uniform(size=B) < 0.3] = -1
    l64, du64 = orc.classifier_hb_loss(u.astype(np.float64), y, 0.8)
    loss = torch.zeros(1, device=DEV)
    du = torch.empty(B, P, device=DEV)
    ty = T(y)
    ws = torch.empty(int(ops.lib.c2_reduce_ws()), device=DEV)
    ops.classifier_loss(T(u), ty, 0.8, loss, du, ws)
    assert abs(loss.item() - l64) < 1e-4 * abs(l64) and rel(du, du64) < 1e-4
    assert (ty.cpu().numpy() == np.where(y == -1, 0, y)).all(), "classes must be mutated in place like the reference"


@pytest.mark.parametrize("B,C,main,boundary", [(40000, 3, None, None), (40000, 4, 1, None), (9000, 4, None, 2),
                                               (37000, 6, 0, None)])
def test_partition_of_a_large_batch(B, C, main, boundary):
    """The table-form partition kernel keeps the keys of its first 4096 rows in registers, those of the next 32 768 in LDS
    and walks the condition matrix a second time only beyond that: batches that reach every one of the three, partition
    ids against the oracle's partition, the pair-weight table bit for bit against the reference's walk over the partition
    pairs (losses.py:232-248) in fp32."""
    ops = _ops()
    rng = np.random.default_rng(B + C)
    cond = np.zeros((B, C), np.int64)
    lab = rng.integers(0, C, B)
    lab[:C] = np.arange(C)
    cond[np.arange(B), lab] = 1
    parts = orc.partition_index(cond, main)
    part_id = torch.empty(B, device=DEV, dtype=torch.int32)
    wt = torch.empty(64 * 64, device=DEV)
    ws_p = torch.empty(int(ops.lib.c2_partition_ws(B)), device=DEV, dtype=torch.uint8)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.partition(T(cond), C, main, boundary, part_id, wt, ws_p, status)
    assert status.item() == 0
    pid = part_id.cpu().numpy()
    want_pid = np.empty(B, np.int32)
    for k, idx in enumerate(parts):
        want_pid[idx] = k
    assert (pid == want_pid).all()
    n = [np.float32(len(idx)) for idx in parts]
    w = np.zeros((64, 64), np.float32)

    def add_pair(i, j):
        w[i, i] += np.float32(1) / (n[i] * n[i])
        w[j, j] += np.float32(1) / (n[j] * n[j])
        w[i, j] -= np.float32(1) / (n[i] * n[j])
        w[j, i] -= np.float32(1) / (n[i] * n[j])

    for i, j in orc.mmd_pairs(len(parts), [len(idx) for idx in parts], C, boundary, main is not None):
        add_pair(i, j)
    assert (wt.cpu().numpy().reshape(64, 64) == w).all()


@pytest.mark.parametrize("B,D,C,main,boundary,multi", [(64, 10, 3, None, None, False), (200, 64, 5, 1, None, False),
                                                       (96, 5, 4, None, 2, True), (150, 100, 2, None, None, False),
                                                       (33, 16, 3, 0, None, True),
                                                       # more than one pass of the 1024-thread block over the rows
                                                       (5000, 8, 3, None, None, False), (2500, 8, 6, 0, None, True),
                                                       # C > 6: the general (sorting) partition kernel
                                                       (300, 8, 7, None, None, True), (260, 12, 9, 2, None, False)])
def test_partition_and_mmd(B, D, C, main, boundary, multi):
    ops = _ops()
    rng = np.random.default_rng(B + D)
    f = rng.standard_normal((B, D)).astype(np.float32)
    cond = np.zeros((B, C), np.int64)
    lab = rng.integers(0, C, B)
    lab[:C] = np.arange(C)
    cond[np.arange(B), lab] = 1
    if multi:
        cond[np.arange(B), rng.integers(0, C, B)] = 1
    try:
        m64, d64 = orc.mmd_loss(f.astype(np.float64), cond, C, 7.0, boundary, main)
    except (IndexError, RuntimeError):
        pytest.skip("reference raises for this partition layout")
    parts = orc.partition_index(cond, main)
    part_id = torch.empty(B, device=DEV, dtype=torch.int32)
    wt = torch.empty(64 * 64, device=DEV)
    ws_p = torch.empty(int(ops.lib.c2_partition_ws(B)), device=DEV, dtype=torch.uint8)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.partition(T(cond), C, main, boundary, part_id, wt, ws_p, status)
    assert status.item() == 0
    pid = part_id.cpu().numpy()
    for k, idx in enumerate(parts):
        assert (pid[idx] == k).all()
    loss = torch.zeros(1, device=DEV)
    df = torch.empty(B, D, device=DEV)
    ws = torch.empty(ops.mmd_ws(B, B, D), device=DEV)
    ops.mmd(T(f), part_id, wt, 0, B, 7.0, loss, df, ws)
    assert abs(loss.item() - m64) < 1e-4 * max(abs(m64), 1e-3), (loss.item(), m64)
    assert rel(df, d64) < 2e-4
    # row-sharded form (data-parallel): two shards reproduce the full gradient and loss
    h = B // 2
    loss2 = torch.zeros(1, device=DEV)
    d0 = torch.empty(h, D, device=DEV)
    d1 = torch.empty(B - h, D, device=DEV)
    ops.mmd(T(f), part_id, wt, 0, h, 7.0, loss2, d0, torch.empty(ops.mmd_ws(h, B, D), device=DEV))
    ops.mmd(T(f), part_id, wt, h, B - h, 7.0, loss2, d1, torch.empty(ops.mmd_ws(B - h, B, D), device=DEV))
    assert abs(loss2.item() - m64) < 1e-4 * max(abs(m64), 1e-3)
    assert rel(torch.cat([d0, d1]), d64) < 2e-4


def test_adam_matches_oracle():
    ops = _ops()
    rng = np.random.default_rng(9)
    shapes = [(19, 7), (1000,), (33, 128), (5,), (5000, 3), (4096,), (3, 4099)]  # whole blocks (4096) and tails
    params = {f"p{i}": rng.standard_normal(s).astype(np.float32) for i, s in enumerate(shapes)}
    tp = {k: T(v.copy()) for k, v in params.items()}
    tm = {k: torch.zeros_like(v) for k, v in tp.items()}
    tv = {k: torch.zeros_like(v) for k, v in tp.items()}
    sh = {k: torch.empty_like(v, dtype=torch.bfloat16) for k, v in tp.items()}
    state = {}
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    for step in range(1, 4):
        grads = {k: rng.standard_normal(v.shape).astype(np.float32) for k, v in params.items()}
        orc.adam_step(p64, {k: g.astype(np.float64) for k, g in grads.items()}, state, clip_value=1.5)
        tg = {k: T(g) for k, g in grads.items()}
        ops.adam_step([(tp[k], tg[k], tm[k], tv[k], sh[k]) for k in tp], 1e-3, 0.9, 0.999, 0.01, 0.04, 1.5, step)
    for k in tp:
        assert rel(tp[k], p64[k]) < 1e-6
        assert rel(tm[k], state[k][0]) < 1e-5 and rel(tv[k], state[k][1]) < 1e-5
        assert torch.equal(sh[k], tp[k].to(torch.bfloat16))


def test_adam_many_tensors_with_an_empty_one():
    """More tensors than one launch's table holds (48), an empty one among them (theta is [G,0] without conditions):
    every tensor is updated exactly once per step."""
    ops = _ops()
    rng = np.random.default_rng(10)
    shapes = [(3 + i % 5, 2 + i % 3) for i in range(60)]
    shapes[7] = (11, 0)
    shapes[52] = (0,)
    params = {f"p{i:02d}": rng.standard_normal(s).astype(np.float32) for i, s in enumerate(shapes)}
    tp = {k: T(v.copy()) for k, v in params.items()}
    tm = {k: torch.zeros_like(v) for k, v in tp.items()}
    tv = {k: torch.zeros_like(v) for k, v in tp.items()}
    state = {}
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    for step in range(1, 3):
        grads = {k: rng.standard_normal(v.shape).astype(np.float32) for k, v in params.items()}
        orc.adam_step(p64, {k: g.astype(np.float64) for k, g in grads.items()}, state)
        tg = {k: T(g) for k, g in grads.items()}
        ops.adam_step([(tp[k], tg[k], tm[k], tv[k], None) for k in tp], 1e-3, 0.9, 0.999, 0.01, 0.04, 0.0, step)
    for k in tp:
        if tp[k].numel():
            assert rel(tp[k], p64[k]) < 1e-6, k
            assert rel(tm[k], state[k][0]) < 1e-5 and rel(tv[k], state[k][1]) < 1e-5, k


def test_prep_and_gather():
    ops = _ops()
    rng = np.random.default_rng(2)
    N, G, C, P = 50, 76, 3, 4
    X = rng.poisson(3.0, (N, G)).astype(np.float32)
    good = rng.uniform(size=(N, G)) > 0.1
    sf = X.sum(1).astype(np.float32)
    cond = rng.integers(0, 2, (N, C)).astype(np.int64)
    cls = rng.integers(-1, 2, (N, P)).astype(np.int64)
    idx = rng.integers(0, N, 21).astype(np.int64)
    xo = torch.empty(21, G, device=DEV)
    go = torch.empty(21, G, device=DEV, dtype=torch.bool)
    so = torch.empty(21, device=DEV)
    co = torch.empty(21, C, device=DEV, dtype=torch.int64)
    po = torch.empty(21, P, device=DEV, dtype=torch.int64)
    xin = torch.empty(21, G, device=DEV, dtype=torch.bfloat16)
    ops.gather_rows(T(X), T(good), T(sf), T(cond), T(cls), T(idx), xo, go, so, co, po, xin, True)
    assert np.array_equal(xo.cpu().numpy(), X[idx]) and np.array_equal(go.cpu().numpy(), good[idx])
    assert np.array_equal(so.cpu().numpy(), sf[idx]) and np.array_equal(co.cpu().numpy(), cond[idx])
    assert np.array_equal(po.cpu().numpy(), cls[idx])
    assert rel(xin, np.log(1 + X[idx].astype(np.float64))) < 1e-2
    # resident-row form: no x / goodloc copies, xin only (fp32 here), vectorised and scalar store paths
    for dt, g_cols in ((torch.float32, G), (torch.bfloat16, G), (torch.bfloat16, G - 1)):
        Xs = np.ascontiguousarray(X[:, :g_cols])
        xin2 = torch.full((21, g_cols), float("nan"), device=DEV, dtype=dt)
        so2 = torch.empty(21, device=DEV)
        ops.gather_rows(T(Xs), None, T(sf), T(cond), None, T(idx), None, None, so2, co, None, xin2, True)
        assert rel(xin2, np.log(1 + Xs[idx].astype(np.float64))) < (1e-6 if dt == torch.float32 else 1e-2)
        assert np.array_equal(so2.cpu().numpy(), sf[idx])
    # more than 256 condition / class columns: the first 256 of a row travel with its other scalars, the rest in a loop
    C2, P2 = 300, 700
    cond2 = rng.integers(0, 2, (N, C2)).astype(np.int64)
    cls2 = rng.integers(-1, 2, (N, P2)).astype(np.int64)
    co2 = torch.empty(21, C2, device=DEV, dtype=torch.int64)
    po2 = torch.empty(21, P2, device=DEV, dtype=torch.int64)
    so3 = torch.empty(21, device=DEV)
    i32 = torch.empty(24, device=DEV, dtype=torch.int32)
    ops.gather_rows(T(X), None, T(sf), T(cond2), T(cls2), T(idx), None, None, so3, co2, po2, None, True, idx32_out=i32)
    assert np.array_equal(co2.cpu().numpy(), cond2[idx]) and np.array_equal(po2.cpu().numpy(), cls2[idx])
    assert np.array_equal(so3.cpu().numpy(), sf[idx])
    assert np.array_equal(i32.cpu().numpy(), np.concatenate([idx, np.repeat(idx[-1:], 3)]).astype(np.int32))
    out = torch.empty(N, G, device=DEV)
    ops.prep_input(T(X), out, True)
    assert rel(out, np.log(1 + X.astype(np.float64))) < 1e-6


def test_deferred_loss_sums_equal_their_own_launches():
    """c2_defer_loss_sums: the loss entry points leave their second-stage sums to the next combine launch, which forms each
    of them as the entry point's own launch would (same strides, same tree) -- bit-identical losses, for sums of fewer
    partials than the combining block has threads, of more than twice as many (the part it walks in a loop), for a sum
    addressed to the loss vector and for one addressed elsewhere."""
    ops = _ops()
    rng = np.random.default_rng(5)

    def case(B, D, C):
        f = T(rng.standard_normal((B, D)).astype(np.float32))
        cond = np.zeros((B, C), np.int64)
        lab = rng.integers(0, C, B)
        lab[:C] = np.arange(C)
        cond[np.arange(B), lab] = 1
        part = torch.empty(B, device=DEV, dtype=torch.int32)
        wt = torch.empty(64 * 64, device=DEV)
        ops.partition(T(cond), C, None, None, part, wt,
                      torch.empty(int(ops.lib.c2_partition_ws(B)), device=DEV, dtype=torch.uint8),
                      torch.zeros(1, device=DEV, dtype=torch.int32))
        return dict(f=f, part=part, wt=wt, B=B, df=torch.empty(B, D, device=DEV),
                    ws=torch.empty(ops.mmd_ws(B, B, D), device=DEV))

    cases = [case(40, 8, 2), case(900, 16, 3), case(4500, 64, 3), case(2100, 32, 4)]

    def run(l4, ext):
        for c, out in zip(cases, (l4[0:1], l4[1:2], l4[2:3], ext)):
            ops.mmd(c["f"], c["part"], c["wt"], 0, c["B"], 1.5, out, c["df"], c["ws"])

    w = (1.0, 0.25, 3.0, 0.5)
    ref4, ref_ext, ref6 = torch.zeros(4, device=DEV), torch.zeros(1, device=DEV), torch.empty(6, device=DEV)
    run(ref4, ref_ext)
    ops.combine_losses(ref4, w, ref6)
    l4, ext, out6 = torch.zeros(4, device=DEV), torch.zeros(1, device=DEV), torch.empty(6, device=DEV)
    ops.defer_loss_sums(True)
    try:
        run(l4, ext)
        ops.combine_losses(l4, w, out6)
    finally:
        ops.defer_loss_sums(False)
        ops.drop_pending_sums()
    torch.cuda.synchronize()
    assert torch.equal(l4, ref4) and torch.equal(ext, ref_ext) and torch.equal(out6, ref6)
    assert all(v != 0 for v in ref4[:3].tolist()) and ref_ext.item() != 0


def test_dropout_masks_rate_and_determinism():
    ops = _ops()
    n = 1_000_003
    a = torch.empty(n, dtype=torch.uint8, device=DEV)
    b = torch.empty(n, dtype=torch.uint8, device=DEV)
    ops.dropout_masks(a, 0.05, 1234, 0)
    ops.dropout_masks(b, 0.05, 1234, 0)
    assert torch.equal(a, b) and set(a.unique().tolist()) <= {0, 1}
    keep = a.float().mean().item()
    assert abs(keep - (1 - 13 / 256)) < 2e-3  # p quantised to 1/256
    ops.dropout_masks(b, 0.05, 1234, n // 16 + 1)
    assert not torch.equal(a, b)


@pytest.mark.parametrize("M,N,K", [(256, 512, 1024), (300, 64, 515), (4096, 100, 70)])
def test_linear_bf16x3_mode(M, N, K):
    """fp32 operands multiplied as hi*hi + hi*lo + lo*hi bf16 MFMA products: ~2^-16 relative product error."""
    ops = _ops()
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = rng.standard_normal((N, K)).astype(np.float32)
    dy = rng.standard_normal((M, N)).astype(np.float32)
    ta, tw, tdy = T(a), T(w), T(dy)
    a64, w64, dy64 = a.astype(np.float64), w.astype(np.float64), dy.astype(np.float64)
    y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(ta, tw, y, x3=True)
    assert rel(y, a64 @ w64.T) < 5e-5
    da = torch.empty(M, K, device=DEV)
    ops.linear_dgrad(tdy, tw, da, x3=True)
    assert rel(da, dy64 @ w64) < 5e-5
    dw = torch.empty(2, N, K, device=DEV)
    ops.linear_wgrad(tdy, ta, dw, nslab=2, x3=True)
    assert rel(dw.sum(0), dy64.T @ a64) < 5e-5


@pytest.mark.parametrize("M,N,K", [(4096, 512, 1024), (1000, 64, 520), (136, 200, 72), (512, 1024, 64)])
def test_linear_bf16_planes(M, N, K):
    """C2_BF16X3P: operands pre-split into bf16 planes (c2_split_bf16), three products on the 128x128 LDS-DMA kernel;
    same ~2^-16 product error as the in-kernel split, ragged tiles, reduction tails, split reductions, bias."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M * 3 + N + K)
    ta = torch.randn(M, K, generator=g).to(DEV)
    tw = torch.randn(N, K, generator=g).to(DEV)
    tdy = torch.randn(M, N, generator=g).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    pa, pw, pdy = ops.split_bf16(ta), ops.split_bf16(tw), ops.split_bf16(tdy)
    # the planes reproduce the operand to ~2^-17
    assert float((pa[0].float() + pa[1].float() - ta).abs().max() / ta.abs().max()) < 2e-5
    a64, w64, dy64 = ta.double(), tw.double(), tdy.double()
    relt = lambda got, want: float((got.double() - want).abs().max() / want.abs().max())  # noqa: E731
    assert ops.planes_ok(M, N, K, (K, K))
    y = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(pa, pw, y, bias=bias)
    assert relt(y, a64 @ w64.T + bias.double()) < 5e-5
    ys = torch.full((3, M, N), float("nan"), device=DEV)
    ops.linear_fwd(pa, pw, ys, nslab=3)
    assert relt(ys.sum(0), a64 @ w64.T) < 5e-5
    alpha = torch.tensor([0.5], device=DEV)
    if ops.planes_ok(M, K, N, (N, K)):
        da = torch.full((2, M, K), float("nan"), device=DEV)
        ops.linear_dgrad(pdy, pw, da, nslab=2, alpha=alpha)
        assert relt(da.sum(0), 0.5 * dy64 @ w64) < 5e-5
    if ops.planes_ok(N, K, M, (N, K)):
        dw = torch.full((N, K), float("nan"), device=DEV)
        ops.linear_wgrad(pdy, pa, dw)
        assert relt(dw, dy64.T @ a64) < 5e-5
        dws = torch.full((4, N, K), float("nan"), device=DEV)
        ops.linear_wgrad(pdy, pa, dws, nslab=4)
        assert relt(dws.sum(0), dy64.T @ a64) < 5e-5


@pytest.mark.parametrize("by_rows", [True, False])
def test_wgrad_two_part_schedule(by_rows):
    """engine.StepEngine._wgrad_two_part: full rounds of 256x256 tiles unsplit + the remaining strip as slabs, along
    the rows or the columns of the weight gradient (strided views of dY / A / dW), against an fp64 product."""
    from celligner2_amd.engine import StepEngine
    ops = _ops()
    eng = StepEngine.__new__(StepEngine)
    eng.x3, eng._ws, eng._pool, eng.recorded, eng.dev = False, {}, {}, None, torch.device(DEV)
    M, big, small = 1024, 66 * 256 + 40, 1024
    N, K = (big, small) if by_rows else (small, big)
    g = torch.Generator(device="cpu").manual_seed(7 + by_rows)
    dy = torch.randn(M, N, generator=g).to(DEV).bfloat16()
    a = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    dw = torch.full((N, K), float("nan"), device=DEV)
    alpha = torch.tensor([0.25], device=DEV)
    assert eng._wgrad_two_part(dy, a, dw, alpha)
    want = 0.25 * dy.double().T @ a.double()
    assert float((dw.double() - want).abs().max() / want.abs().max()) < 2e-5
    # shapes that fill whole rounds (or are too small) are left to the ordinary path
    assert not eng._wgrad_two_part(dy[:, :256 * 4] if by_rows else dy, a if by_rows else a[:, :256 * 4],
                                   torch.empty((256 * 4, K) if by_rows else (N, 256 * 4), device=DEV), None)


@pytest.mark.parametrize("B,G,C,multi", [(6, 200, 2, True), (4, 3001, 3, False), (3, 19000, 2, False), (5, 1500, 2, True),
                                         (4, 4000, 3, False), (3, 8000, 2, True), (3, 12000, 3, False),
                                         (2, 16000, 2, False), (2, 20480, 2, False), (2, 20484, 2, False)])
@pytest.mark.parametrize("d_dtype", [torch.float32, torch.bfloat16])
def test_recon_zinb(B, G, C, multi, d_dtype):
    """c2_recon_zinb_fwd_bwd vs the oracle (pinned to the reference by tests/golden/zinb_mmd_z.npz): every instance of the
    fast kernel (G % 4 == 0, up to 20 480 genes) and the general kernel (odd G, G beyond the fast kernel)."""
    ops = _ops()
    rng = np.random.default_rng(G + B)
    o, x, good, sf, theta, cond = _nb_inputs(rng, B, G, C, multi)
    x[:, ::3] = 0.0  # plenty of zeros: the zero-inflation branch
    pi = rng.standard_normal((B, G)).astype(np.float32) * 2
    pi[:, 5::7] *= 6.0  # saturated dropout logits: exp(-|pi|) below the log1p series switch
    recon64, do64, dpi64, dth64 = orc.zinb_loss(o.astype(np.float64), pi.astype(np.float64), x.astype(np.float64), good,
                                                sf.astype(np.float64), theta.astype(np.float64), cond)
    loss = torch.zeros(1, device=DEV)
    dl = torch.empty(B, G, device=DEV, dtype=d_dtype)
    dpi = torch.empty(B, G, device=DEV, dtype=d_dtype)
    dth = torch.empty(G, C, device=DEV)
    ws = torch.empty(ops.recon_nb_ws(B, G, C), device=DEV)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.recon_zinb(T(o), T(pi), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss, dl, dpi, dth, ws, status)
    assert status.item() == 0
    assert abs(loss.item() - recon64) < 1e-4 * abs(recon64)
    tol = 1e-4 if d_dtype == torch.float32 else 1e-2
    assert rel(dl, do64) < tol and rel(dpi, dpi64) < tol
    assert rel(dth, dth64) < 1e-4
    # the same batch read through a row index from a larger resident matrix: bit-identical
    perm = rng.permutation(B + 3)[:B]
    xf = rng.poisson(3.0, (B + 3, G)).astype(np.float32)
    gf = rng.uniform(size=(B + 3, G)) > 0.5
    xf[perm], gf[perm] = x, good
    loss2 = torch.zeros(1, device=DEV)
    dl2, dpi2, dth2 = torch.empty_like(dl), torch.empty_like(dpi), torch.empty_like(dth)
    ops.recon_zinb(T(o), T(pi), T(xf), T(gf), T(sf), T(theta), T(cond), 1.0 / B, loss2, dl2, dpi2, dth2, ws, status,
                   row_index=T(perm.astype(np.int64)))
    assert loss2.item() == loss.item()
    assert torch.equal(dl2, dl) and torch.equal(dpi2, dpi) and torch.equal(dth2, dth)
    # the data-only term sum goodloc * lgamma(x + 1) handed in (c2_recon_zinb_fwd_bwd_rc): same loss, identical gradients
    lgx = ops.nb_row_const(T(x), T(good))
    loss3 = torch.zeros(1, device=DEV)
    dl3, dpi3, dth3 = torch.empty_like(dl), torch.empty_like(dpi), torch.empty_like(dth)
    ops.recon_zinb(T(o), T(pi), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss3, dl3, dpi3, dth3, ws, status,
                   lgx1=lgx)
    assert abs(loss3.item() - recon64) < 1e-4 * abs(recon64)
    assert torch.equal(dl3, dl) and torch.equal(dpi3, dpi) and torch.equal(dth3, dth)


def test_recon_zinb_row_walk_and_bad_rows():
    """Several rows per workgroup with the row pattern changing inside a walk (the LDS accumulator is flushed per pattern),
    a tenth of the entries masked; then a condition value outside {0, 1} raises the status word."""
    ops = _ops()
    rng = np.random.default_rng(23)
    B, G, C = 700, 2400, 2
    o, x, good, sf, theta, cond = _nb_inputs(rng, B, G, C, multi=True)
    good &= rng.uniform(size=(B, G)) > 0.1
    cond[rng.integers(0, B, 40), :] = 1
    cond[rng.integers(0, B, 40), :] = 0
    x[:, 1::2] = 0.0
    pi = rng.standard_normal((B, G)).astype(np.float32) * 2
    recon64, do64, dpi64, dth64 = orc.zinb_loss(o.astype(np.float64), pi.astype(np.float64), x.astype(np.float64), good,
                                                sf.astype(np.float64), theta.astype(np.float64), cond)
    loss = torch.zeros(1, device=DEV)
    dl, dpi = torch.empty(B, G, device=DEV), torch.empty(B, G, device=DEV)
    dth = torch.empty(G, C, device=DEV)
    ws = torch.empty(ops.recon_nb_ws(B, G, C), device=DEV)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.recon_zinb(T(o), T(pi), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss, dl, dpi, dth, ws, status)
    assert status.item() == 0
    assert abs(loss.item() - recon64) < 1e-4 * abs(recon64)
    assert rel(dl, do64) < 1e-4 and rel(dpi, dpi64) < 1e-4 and rel(dth, dth64) < 1e-4
    cond[17, 0] = 3
    ops.recon_zinb(T(o), T(pi), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss, dl, dpi, dth, ws, status)
    assert status.item() == 1


@pytest.mark.parametrize("M,N,K,nslab", [(2104, 1024, 2312, 1), (4096, 1024, 19000, 4), (2304, 1280, 4096, 2)])
def test_linear_rows_equal_gathered_operands(M, N, K, nslab):
    """Row-indexed forward / weight gradient (the batch read in place from a resident bf16 matrix through an int32 row
    index) against the plain entry points on a gathered copy: the same products in the same order, bit for bit.
    Ragged M (edge tiles re-read rows), reduction tails (K or M not a multiple of 64), split reductions."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    n_res = M + 900
    a_full = torch.randn(n_res, K, generator=g).to(DEV).to(torch.bfloat16)
    rows = torch.randint(0, n_res, (M,), generator=g).to(torch.int32)
    pad = (M + 63) // 64 * 64
    rows_p = torch.cat([rows, rows[-1:].expand(pad - M)]).contiguous().to(DEV)
    a = a_full[rows_p[:M].long()].contiguous()
    w = torch.randn(N, K, generator=g).to(DEV).to(torch.bfloat16)
    assert ops.gemm_tile(torch.bfloat16, M, N, K) == 256
    shape = (nslab, M, N) if nslab > 1 else (M, N)
    y0, y1 = torch.empty(shape, device=DEV), torch.empty(shape, device=DEV)
    bias = torch.randn(N, generator=g).to(DEV) if nslab == 1 else None
    ops.linear_fwd(a, w, y0, bias=bias, nslab=nslab)
    ops.linear_fwd_rows(a_full, rows_p, M, w, y1, bias=bias, nslab=nslab)
    assert torch.equal(y0, y1)
    dy = torch.randn(M, N, generator=g).to(DEV).to(torch.bfloat16)
    alpha = torch.tensor([0.5], device=DEV)
    shape = (nslab, N, K) if nslab > 1 else (N, K)
    d0, d1 = torch.empty(shape, device=DEV), torch.empty(shape, device=DEV)
    ops.linear_wgrad(dy, a, d0, nslab=nslab, alpha=alpha)
    ops.linear_wgrad_rows(dy, a_full, rows_p, d1, nslab=nslab, alpha=alpha)
    assert torch.equal(d0, d1)
    with pytest.raises(ops.L.C2Error):  # a shape the 256x256 kernel does not take: the caller gathers instead
        ops.linear_fwd_rows(a_full[:, :64], rows_p, 64, w[:128, :64].contiguous(), torch.empty(64, 128, device=DEV))


def test_mmd_at_the_eight_gpu_shape():
    """One rank's share of the 8-GPU weak-scaling step: 4096 own rows against 32 768 gathered rows, D = 64.
    Too large for the oracle: the partition (rows beyond the 8192 whose keys stay in registers) is checked against
    numpy's unique rows, and the four-wave fast MMD kernel against the general one-wave kernel (taken for a feature
    matrix whose row stride is not a multiple of 4), which the small cases pin to the oracle."""
    ops = _ops()
    Bg, Bl, D, C, row0 = 32768, 4096, 64, 3, 8192
    g = torch.Generator(device="cpu").manual_seed(5)
    f = torch.randn(Bg, D, generator=g).to(DEV)
    lab = torch.randint(0, C, (Bg,), generator=g)
    cond = torch.zeros(Bg, C, dtype=torch.int64)
    cond[torch.arange(Bg), lab] = 1
    part_id = torch.empty(Bg, device=DEV, dtype=torch.int32)
    wt = torch.empty(64 * 64, device=DEV)
    ws_p = torch.empty(int(ops.lib.c2_partition_ws(Bg)), device=DEV, dtype=torch.uint8)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    ops.partition(cond.to(DEV), C, None, None, part_id, wt, ws_p, status)
    assert status.item() == 0
    parts = orc.partition_index(cond.numpy(), None)
    pid = part_id.cpu().numpy()
    for k, idx in enumerate(parts):
        assert (pid[idx] == k).all()
    out = []
    for ld in (D, D + 1):  # row stride 64: fast kernel; 65: general kernel
        buf = torch.zeros(Bg, ld, device=DEV)
        buf[:, :D] = f
        feat = buf[:, :D]
        loss = torch.zeros(1, device=DEV)
        df = torch.empty(Bl, D, device=DEV)
        ops.mmd(feat, part_id, wt, row0, Bl, 3.0, loss, df, torch.empty(ops.mmd_ws(Bl, Bg, D), device=DEV))
        out.append((loss.item(), df))
    assert abs(out[0][0] - out[1][0]) < 1e-5 * abs(out[1][0]), (out[0][0], out[1][0])
    assert rel(out[0][1], out[1][1].cpu().numpy()) < 1e-4


@pytest.mark.parametrize("M,N,K,nslab", [(4096, 1024, 19000, 1), (2048, 1280, 4096, 2), (1032, 820, 4104, 1)])
def test_weight_gradient_with_a_transposed_operand(M, N, K, nslab):
    """c2_linear_wgrad_bt / c2_linear_wgrad_rows_at (one operand handed over transposed, read row-wise, 16x16x32 MFMA)
    against c2_linear_wgrad / c2_linear_wgrad_rows: the same products in the same order, bit for bit; and the transpose
    kernel itself.  Ragged edges, split reductions, a width that is not a multiple of 8 (padded rows)."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    ldn = (N + 7) // 8 * 8
    dy = torch.randn(M, ldn, generator=g).to(DEV).to(torch.bfloat16)[:, :N]
    a = torch.randn(M + 5, K, generator=g).to(DEV).to(torch.bfloat16)
    rows = torch.randint(0, M + 5, ((M + 63) // 64 * 64,), generator=g).to(torch.int32).to(DEV)
    dyt = ops.transpose_bf16(dy, torch.empty(N, M, device=DEV, dtype=torch.bfloat16))
    assert torch.equal(dyt, dy.t())
    alpha = torch.tensor([0.25], device=DEV)
    shape = (nslab, N, K) if nslab > 1 else (N, K)
    d0, d1 = torch.empty(shape, device=DEV), torch.empty(shape, device=DEV)
    ops.linear_wgrad_rows(dy, a, rows, d0, nslab=nslab, alpha=alpha)
    ops.linear_wgrad_rows_at(dyt, a, rows, d1, nslab=nslab, alpha=alpha)
    assert torch.equal(d0, d1)
    # the other side transposed: dW[K2, N] = dl^T @ dy with dy as the narrow operand
    K2 = 4608
    dl = torch.randn(M, K2, generator=g).to(DEV).to(torch.bfloat16)
    shape = (nslab, K2, N) if nslab > 1 else (K2, N)
    e0, e1 = torch.empty(shape, device=DEV), torch.empty(shape, device=DEV)
    ops.linear_wgrad(dl, dy, e0, nslab=nslab, alpha=alpha)
    ops.linear_wgrad_bt(dl, dyt, e1, nslab=nslab, alpha=alpha)
    assert torch.equal(e0, e1)
    with pytest.raises(ops.L.C2Error):  # a shape the 256x256 kernel does not take
        ops.linear_wgrad_bt(dl[:64, :128], dyt[:128, :64].contiguous(), torch.empty(128, 128, device=DEV))


_W16_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from celligner2_amd import ops
g = torch.Generator(device="cpu").manual_seed(11)
M, N, K = 2104, 1024, 2312
a = torch.randn(M + 7, K, generator=g).cuda().to(torch.bfloat16)
w = torch.randn(N, K, generator=g).cuda().to(torch.bfloat16)
dy = torch.randn(M, N, generator=g).cuda().to(torch.bfloat16)
rows = torch.randint(0, M + 7, ((M + 63) // 64 * 64,), generator=g).to(torch.int32).cuda()
out = {}
out["fwd"] = ops.linear_fwd(a[:M], w, torch.empty(2, M, N, device="cuda"), nslab=2)
out["fwd_rows"] = ops.linear_fwd_rows(a, rows, M, w, torch.empty(M, N, device="cuda"), bias=torch.ones(N, device="cuda"))
out["dgrad"] = ops.linear_dgrad(dy, w, torch.empty(M, K, device="cuda"))
dyt = ops.transpose_bf16(dy, torch.empty(N, M, device="cuda", dtype=torch.bfloat16))
out["wgrad_rows_at"] = ops.linear_wgrad_rows_at(dyt, a, rows, torch.empty(N, K, device="cuda"))
at = ops.transpose_bf16(a[:M].contiguous(), torch.empty(K, M, device="cuda", dtype=torch.bfloat16))
out["wgrad_bt"] = ops.linear_wgrad_bt(dy, at, torch.empty(2, N, K, device="cuda"), nslab=2)
torch.save({k: v.cpu() for k, v in out.items()}, sys.argv[2])
"""


def test_sixteen_wave_form_of_the_gene_width_kernel_equals_the_eight_wave_form(tmp_path):
    """The 256x256 kernel on sixteen waves of 64x64 (default for the (XC, RC) weight gradient only; C2_WIDE_WAVES=16 / 8 puts
    every eligible layout / none on it, read once per process): same K-tile order, same MFMA -- bit-identical outputs for
    the forward, row-indexed forward, input gradient and both transposed-operand weight gradients, ragged edges included."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for waves in ("8", "16"):
        f = str(tmp_path / f"w{waves}.pt")
        r = subprocess.run([sys.executable, "-c", _W16_SCRIPT, root, f], env=dict(os.environ, C2_WIDE_WAVES=waves),
                           capture_output=True, text=True, timeout=100)
        assert r.returncode == 0, r.stderr[-2000:]
        res[waves] = torch.load(f, weights_only=True)
    for k in res["8"]:
        assert torch.equal(res["8"][k], res["16"][k]), k
        assert torch.isfinite(res["8"][k]).all() and res["8"][k].abs().max() > 0


@pytest.mark.parametrize("B,G", [(1, 7), (5, 19000), (130, 1001), (64, 25000)])
def test_decoder_output_softmax_and_relu(B, G):
    """c2_decoder_output: nn.Softmax(dim=-1) / nn.ReLU of the decoder's output layer on the inference path
    (modules.py:467-482, celligner2.py:411-420) vs numpy in float64; strided rows; rows sum to one."""
    ops = _ops()
    rng = np.random.default_rng(B + G)
    lg = (rng.standard_normal((B, G + 3)) * 4).astype(np.float32)
    lg[0, : min(G, 3)] = [30.0, -40.0, 0.0][: min(G, 3)]
    t = T(lg)[:, :G]  # row stride G + 3
    sm = ops.decoder_output(t, "softmax").cpu().numpy()
    e = np.exp(lg[:, :G].astype(np.float64) - lg[:, :G].max(1, keepdims=True))
    ref = e / e.sum(1, keepdims=True)
    assert np.abs(sm - ref).max() <= 1e-5 * ref.max()
    assert np.abs(sm.astype(np.float64).sum(1) - 1).max() < 1e-4
    assert np.array_equal(ops.decoder_output(t, "relu").cpu().numpy(), np.maximum(lg[:, :G], 0))
    assert ops.decoder_output(T(np.zeros((0, 5), np.float32)), "softmax").shape == (0, 5)


@pytest.mark.parametrize("R,G", [(1, 5), (70, 76), (300, 1001), (4096, 19000)])
def test_centred_encoder_input_entry_points(R, G):
    """c2_col_mean / c2_prep_input + c2_gather_rows with a centre / c2_gemv_bias / c2_rank1_add: the four pieces of
    h = (t(x) - c) W^T + (b + W c),  dW = dh^T (t(x) - c) + (sum dh) (x) c  (include/c2mi.h), each against numpy float64."""
    ops = _ops()
    rng = np.random.default_rng(R + G)
    X = rng.poisson(rng.uniform(0.5, 2, (R, 1)) * np.exp(rng.normal(1.0, 1.0, (1, G)))).astype(np.float32)
    t64 = np.log1p(X.astype(np.float64))
    for log in (True, False):
        want = (t64 if log else X.astype(np.float64)).mean(0)
        c = ops.col_mean(T(X), log)
        assert np.abs(c.cpu().numpy() - want).max() <= 2e-6 * max(np.abs(want).max(), 1.0)
    c = ops.col_mean(T(X), True)
    c64 = c.double().cpu().numpy()
    for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2.0 ** -8)):
        out = torch.empty(R, G, device=DEV, dtype=dt)
        ops.prep_input(T(X), out, True, c)
        d = t64 - c64
        assert np.abs(out.double().cpu().numpy() - d).max() <= tol * max(np.abs(d).max(), 1e-3) + 1e-6
    idx = rng.integers(0, R, 9).astype(np.int64)
    xin = torch.empty(9, G, device=DEV, dtype=torch.bfloat16)
    so = torch.empty(9, device=DEV)
    ops.gather_rows(T(X), None, T(X.sum(1)), None, None, T(idx), None, None, so, None, None, xin, True, center=c)
    ref = torch.empty(R, G, device=DEV, dtype=torch.bfloat16)
    ops.prep_input(T(X), ref, True, c)
    assert torch.equal(xin, ref[T(idx)])  # the gather and the whole-matrix conversion round the same values
    # the centre's share of the layer and of its weight gradient
    N = 37
    W = rng.standard_normal((N, G)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    out = torch.empty(N, device=DEV)
    ops.gemv_bias(T(W), c, T(b), out)
    want = b + W.astype(np.float64) @ c64
    assert np.abs(out.cpu().numpy() - want).max() <= 1e-5 * np.abs(want).max()
    # summed in double: the centre does not depend on how the rows are cut (ranks all-reduce their sums)
    if R >= 2:
        h = R // 2
        both = ops.col_sum(T(X[:h]), True) + ops.col_sum(T(X[h:]), True)
        c2 = ops.col_mean_from_sums(both, G, R, torch.empty(G, device=DEV))
        assert torch.equal(c2, c)
    ops.gemv_bias(T(W)[:, :G], c, None, out)
    assert np.abs(out.cpu().numpy() - (want - b)).max() <= 1e-5 * np.abs(want).max()
    g = rng.standard_normal((N, G)).astype(np.float32)
    r = rng.standard_normal(N).astype(np.float32)
    tg = T(g)
    ops.rank1_add(tg, T(r), c)
    assert np.abs(tg.cpu().numpy() - (g + np.outer(r, c64))).max() <= 1e-6 * np.abs(g).max() + 1e-6


@pytest.mark.parametrize("M,N,K", [(256, 1024, 19000), (4096, 1024, 2304), (128, 256, 19000), (37, 300, 130), (5, 64, 2048)])
def test_linear_forward_on_ieee_half_operands(M, N, K):
    """C2_F16 operands (the first layer's forward in bf16 mode): the LDS-DMA kernels' half-precision twin (256x256 and
    128x128 tiles), its row-indexed form and the register-staged fallback, against numpy float64 on the rounded operands
    (only the fp32 accumulation differs) -- and 8x closer to the unrounded product than bf16 operands."""
    ops = _ops()
    rng = np.random.default_rng(M + N + K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ta, tw = T(a, torch.float16), T(w, torch.float16)
    y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(ta, tw, y, bias=T(bias))
    ref = ta.double().cpu().numpy() @ tw.double().cpu().numpy().T + bias
    assert rel(y, ref) < 1e-5
    exact = a.astype(np.float64) @ w.astype(np.float64).T + bias
    yb = torch.empty(M, N, device=DEV)
    ops.linear_fwd(T(a, torch.bfloat16), T(w, torch.bfloat16), yb, bias=T(bias))
    e16, eb = rel(y, exact), rel(yb, exact)
    assert e16 < 0.25 * eb, (e16, eb)
    if ops.gemm_tile(torch.float16, M, N, K) == 256 and M % 8 == 0:
        n_res = M + 300
        full = torch.randn(n_res, K, device=DEV).to(torch.float16)
        rows = torch.randint(0, n_res, ((M + 63) // 64 * 64,), device=DEV).to(torch.int32)
        y0, y1 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
        ops.linear_fwd(full[rows[:M].long()].contiguous(), tw, y0)
        ops.linear_fwd_rows(full, rows, M, tw, y1)
        assert torch.equal(y0, y1)


def test_half_precision_outputs_of_the_input_kernels_and_adam():
    """C2_F16 outputs: c2_prep_input, c2_gather_rows (xin as halves + its bf16 twin in the same pass), c2_cast_f16, and the
    Adam kernel's half-precision operand copy (shadow_f16) -- each the correctly rounded value of the fp32 result."""
    ops = _ops()
    rng = np.random.default_rng(4)
    for R, G in ((9, 76), (33, 1001)):
        X = rng.poisson(4.0, (R, G)).astype(np.float32)
        c = ops.col_mean(T(X), True)
        t32 = torch.empty(R, G, device=DEV)
        ops.prep_input(T(X), t32, True, c)
        h = torch.empty(R, G, device=DEV, dtype=torch.float16)
        ops.prep_input(T(X), h, True, c)
        assert torch.equal(h, t32.to(torch.float16))
        idx = T(rng.integers(0, R, 7).astype(np.int64))
        xin = torch.empty(7, G, device=DEV, dtype=torch.float16)
        g8 = (G + 7) // 8 * 8
        xinb = torch.zeros(7, g8, device=DEV, dtype=torch.bfloat16)[:, :G]
        so = torch.empty(7, device=DEV)
        ops.gather_rows(T(X), None, T(X.sum(1)), None, None, idx, None, None, so, None, None, xin, True, center=c,
                        xin2_out=xinb)
        assert torch.equal(xin, t32[idx].to(torch.float16)) and torch.equal(xinb, t32[idx].to(torch.bfloat16))
    p = torch.randn(1000, device=DEV) * 0.01
    d = torch.empty(1000, device=DEV, dtype=torch.float16)
    ops.cast_bf16(p, d)
    assert torch.equal(d, p.to(torch.float16))
    for n in (1024, 1003):
        p = torch.randn(n, device=DEV) * 0.01
        g, m, v = torch.randn(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        sh = torch.empty(n, device=DEV, dtype=torch.float16)
        p2, m2, v2 = p.clone(), m.clone(), v.clone()
        ops.adam_step([(p, g, m, v, sh)], 1e-3, 0.9, 0.999, 0.01, 0.04, 0.0, 1)
        ops.adam_step([(p2, g, m2, v2, None)], 1e-3, 0.9, 0.999, 0.01, 0.04, 0.0, 1)
        assert torch.equal(p, p2) and torch.equal(sh, p.to(torch.float16))


@pytest.mark.parametrize("M,N,K,dtype", [(256, 19000, 1024, torch.bfloat16), (128, 19000, 256, torch.bfloat16),
                                         (37, 300, 130, torch.bfloat16), (64, 200, 96, torch.float32)])
def test_linear_forward_with_half_precision_output(M, N, K, dtype):
    """c2_linear_fwd_f16out: the forward GEMM's result rounded once to IEEE halves == the fp32 result of c2_linear_fwd
    rounded by torch (every kernel family: 256x256 and 128x128 LDS-DMA tiles, the register-staged kernel, bf16x3)."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(DEV).to(dtype)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV).to(dtype)
    bias = torch.randn(N, generator=g).to(DEV)
    y32 = torch.empty(M, N, device=DEV)
    y16 = torch.empty(M, N, device=DEV, dtype=torch.float16)
    x3 = dtype == torch.float32
    ops.linear_fwd(a, w, y32, bias=bias, x3=x3)
    ops.linear_fwd_f16out(a, w, y16, bias=bias, x3=x3)
    assert torch.equal(y16, y32.to(torch.float16))


@pytest.mark.parametrize("B,G", [(300, 19000), (64, 19000), (40, 20000), (33, 18436)])
def test_recon_nb_on_half_precision_logits(B, G):
    """The NB loss kernel fed IEEE-half logits (both register-resident forms: B above and below the workgroup count) ==
    the same kernel fed those halves widened to fp32, bit for bit: loss, dlogits, dtheta, the bias gradient."""
    ops = _ops()
    assert ops.nb_logits_f16_ok(B, G) and not ops.nb_logits_f16_ok(B, 4096) and not ops.nb_logits_f16_ok(B, 19001)
    rng = np.random.default_rng(B + G)
    C = 3
    x = T(rng.poisson(rng.uniform(0.5, 2, (B, 1)) * np.exp(rng.normal(1.0, 1.5, (1, G)))).astype(np.float32))
    good = T(rng.uniform(size=(B, G)) > 0.01)
    lg16 = T((rng.standard_normal((B, G)) * 2).astype(np.float32)).to(torch.float16)
    cond = np.zeros((B, C), np.int64)
    cond[np.arange(B), rng.integers(0, C, B)] = 1
    theta = T((rng.standard_normal((G, C)) * 0.5).astype(np.float32))
    sf = x.sum(1)
    out = []
    for lg in (lg16, lg16.float()):
        loss = torch.zeros(1, device=DEV)
        dl = torch.empty(B, G, device=DEV, dtype=torch.bfloat16)
        dth = torch.empty(G, C, device=DEV)
        db = torch.empty(G, device=DEV)
        ws = torch.empty(ops.recon_nb_ws(B, G, C), device=DEV)
        st = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.recon_nb(lg, x, good, sf, theta, T(cond), 1.0 / B, loss, dl, dth, ws, st, dbias=db)
        assert int(st.item()) == 0
        out.append((loss.clone(), dl, dth, db))
    for u, v in zip(*out):
        assert torch.equal(u, v)
    with pytest.raises(ops.L.C2Error):  # a width outside the register-resident forms: fp32 logits only
        ops.recon_nb(lg16[:, :4096].contiguous(), x[:, :4096].contiguous(), good[:, :4096].contiguous(), sf,
                     theta[:4096].contiguous(), T(cond), 1.0 / B, loss, torch.empty(B, 4096, device=DEV, dtype=torch.bfloat16),
                     torch.empty(4096, C, device=DEV), ws, st, dbias=torch.empty(4096, device=DEV))


@pytest.mark.parametrize("M,N,K,nslab", [(4096, 1024, 2304, 1), (1024, 512, 2560, 3), (96, 40, 130, 1), (200, 72, 515, 2)])
def test_weight_gradient_with_a_rank_one_term_in_its_epilogue(M, N, K, nslab):
    """rank1_row (x) rank1_col added by the weight-gradient GEMMs' epilogue (every form; the first split of a split
    reduction adds it once) == the same launch followed by c2_rank1_add, to fp32 rounding of one add."""
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    dy = torch.randn(M, N, generator=g).to(DEV).to(torch.bfloat16)
    a = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    r = torch.randn(N, generator=g).to(DEV)
    c = torch.randn((K + 7) // 8 * 8, generator=g).to(DEV)
    alpha = torch.tensor([0.5], device=DEV)
    shape = (nslab, N, K) if nslab > 1 else (N, K)

    def close(got, want):
        if nslab > 1:
            got, want = got.sum(0), want.sum(0)
        assert (got - want).abs().max().item() <= 1e-6 * want.abs().max().item()

    def ref(launch):
        out = torch.empty(shape, device=DEV)
        launch(out, None)
        ops.rank1_add(out if nslab == 1 else out[0], r, c)
        return out

    forms = [lambda out, r1: ops.linear_wgrad(dy, a, out, nslab=nslab, alpha=alpha, rank1=r1)]
    if ops.gemm_tile(torch.bfloat16, N, K, M) == 256 and M % 8 == 0:
        at = a.t().contiguous()
        dyt = dy.t().contiguous()
        full = torch.randn(M + 100, K, generator=g).to(DEV).to(torch.bfloat16)
        rows = torch.randint(0, M + 100, ((M + 63) // 64 * 64,), generator=g).to(torch.int32).to(DEV)
        forms += [lambda out, r1: ops.linear_wgrad_bt(dy, at, out, nslab=nslab, alpha=alpha, rank1=r1),
                  lambda out, r1: ops.linear_wgrad_rows(dy, full, rows, out, nslab=nslab, alpha=alpha, rank1=r1),
                  lambda out, r1: ops.linear_wgrad_rows_at(dyt, full, rows, out, nslab=nslab, alpha=alpha, rank1=r1)]
    for launch in forms:
        out = torch.empty(shape, device=DEV)
        launch(out, (r, c))
        close(out, ref(launch))


def test_sum_slabs_multi_closes_several_split_reductions_in_one_launch():
    ops = _ops()
    g = torch.Generator(device="cpu").manual_seed(3)
    jobs, want = [], []
    for S, shape in ((8, (512, 1024)), (3, (7, 13)), (2, (128, 512)), (16, (64, 64)), (1, (5,))) * 4:  # 20 jobs: two launches
        slabs = torch.randn((S,) + shape, generator=g).to(DEV)
        out = torch.full(shape, float("nan"), device=DEV)
        ref = torch.empty(shape, device=DEV)
        ops.sum_slabs(slabs, ref)
        jobs.append((slabs, out))
        want.append(ref)
    ops.sum_slabs_multi(jobs)
    for (_, out), ref in zip(jobs, want):
        assert torch.equal(out, ref)  # the same order of additions as the single-job kernel
    ops.sum_slabs_multi([])
