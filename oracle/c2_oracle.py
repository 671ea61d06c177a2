"""CPU oracle for the Celligner2 conditional-VAE training step.

TEST INFRASTRUCTURE ONLY.  This file is a numpy restatement of the reference's
algorithm for the hot path; it is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` — as
the checker / the timed CPU port, never as the product.  The product path
(``celligner2_amd``) never imports it and fails loudly without its HIP library.

Parity pin: the reference's own tests hold no numeric fixtures for this path
(SURVEY.md §4), so the oracle is pinned against outputs of the reference itself,
run in the build container through ``tools/ref_harness.py`` and committed as
``tests/golden/*.npz`` by ``tools/gen_golden.py`` (``tests/test_oracle_golden.py``
checks every one of them).

Each function cites the reference file:line it restates (paths relative to the
reference root).  All maths is done in the dtype of the inputs (float64 for a
tight check, float32 to mirror the reference's arithmetic).
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional

import numpy as np
from scipy.special import digamma, gammaln

MMD_ALPHAS = [
    1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1, 5, 10, 15, 20, 25, 30, 35, 100,
    1e3, 1e4, 1e5, 1e6,
]  # celligner2/model/losses.py:179-199


@dataclasses.dataclass
class Cfg:
    """Constructor arguments of the reference module that matter on the path
    (celligner2/model/celligner2.py:69-105)."""

    input_dim: int
    hidden_layer_sizes: List[int]
    latent_dim: int
    n_conditions: int
    n_predictors: int = 0
    classifier_hidden_layer_sizes: List[int] = dataclasses.field(default_factory=lambda: [32])
    recon_loss: str = "nb"
    dr_rate: float = 0.0
    use_own_kl: bool = False
    applylog: bool = True
    use_mmd: bool = False
    mmd_on: str = "z"
    mmd_boundary: Optional[int] = None
    beta: float = 1.0
    main_dataset: Optional[int] = None  # index into the condition encoder
    betaclass: float = 0.8
    use_ln: bool = True
    use_bn: bool = False  # BatchNorm1d(affine) instead of LayerNorm (modules.py:298-302,405-413,445-449,557-566,596-600)

    def to_json(self):
        return dataclasses.asdict(self)


# ----------------------------------------------------------------------------
# parameter naming (state_dict keys of the reference)
# ----------------------------------------------------------------------------
def param_shapes(cfg: Cfg) -> Dict[str, tuple]:
    G, L, C, P = cfg.input_dim, cfg.latent_dim, cfg.n_conditions, cfg.n_predictors
    hs = list(cfg.hidden_layer_sizes)
    assert len(hs) >= 2, "reference needs >= 2 hidden sizes (modules.py:434-435,498-499)"
    out: Dict[str, tuple] = {}
    if cfg.recon_loss in ("nb", "zinb"):
        out["theta"] = (G, C)
    enc = [G] + hs
    for i, (a, b) in enumerate(zip(enc[:-1], enc[1:])):
        if i == 0:
            out["encoder.FC.L0.expr_L.weight"] = (b, a)
            out["encoder.FC.L0.expr_L.bias"] = (b,)
            if C:
                out["encoder.FC.L0.cond_L.weight"] = (b, C)
        else:
            out[f"encoder.FC.L{i}.weight"] = (b, a)
            out[f"encoder.FC.L{i}.bias"] = (b,)
    out["encoder.mean_encoder.weight"] = (L, hs[-1])
    out["encoder.mean_encoder.bias"] = (L,)
    out["encoder.log_var_encoder.weight"] = (L, hs[-1])
    out["encoder.log_var_encoder.bias"] = (L,)
    cl = [L] + list(cfg.classifier_hidden_layer_sizes)
    out["classifier.FirstL.L0.weight"] = (cl[1], cl[0])
    for i, (a, b) in enumerate(zip(cl[1:-1], cl[2:])):
        out[f"classifier.HiddenL.L{i + 1}.weight"] = (b, a)
    out["classifier.classifier.0.weight"] = (P, cl[-1])
    out["classifier.classifier.0.bias"] = (P,)
    dec = [L] + hs[::-1] + [G]
    out["decoder.FirstL.L0.expr_L.weight"] = (dec[1], dec[0])
    if C:
        out["decoder.FirstL.L0.cond_L.weight"] = (dec[1], C)
    for i, (a, b) in enumerate(zip(dec[1:-2], dec[2:-1])):
        out[f"decoder.HiddenL.L{i + 1}.weight"] = (b, a)
    head = "recon_decoder" if cfg.recon_loss == "mse" else "mean_decoder"
    out[f"decoder.{head}.0.weight"] = (G, dec[-2])
    out[f"decoder.{head}.0.bias"] = (G,)
    if cfg.recon_loss == "zinb":  # modules.py:471-477: a second, plain Linear head for the dropout logits
        out["decoder.dropout_decoder.weight"] = (G, dec[-2])
        out["decoder.dropout_decoder.bias"] = (G,)
    if cfg.use_bn:
        for nm, h in norm_names(cfg).values():
            out[nm + ".weight"] = (h,)
            out[nm + ".bias"] = (h,)
    return out


def norm_names(cfg: Cfg) -> Dict[str, tuple]:
    """block key -> (module name of its norm layer in the reference tree, width)."""
    hs = list(cfg.hidden_layer_sizes)
    out = {f"enc{i}": (f"encoder.FC.N{i}", h) for i, h in enumerate(hs)}
    for i, h in enumerate(hs[::-1]):
        out[f"dec{i}"] = ("decoder.FirstL.N0" if i == 0 else f"decoder.HiddenL.N{i}", h)
    for i, h in enumerate(cfg.classifier_hidden_layer_sizes):
        out[f"cls{i}"] = ("classifier.FirstL.N0" if i == 0 else f"classifier.HiddenL.N{i}", h)
    return out


def buffer_shapes(cfg: Cfg) -> Dict[str, tuple]:
    """BatchNorm1d buffers of the reference's state_dict (torch.nn.BatchNorm1d: running_mean, running_var,
    num_batches_tracked)."""
    out: Dict[str, tuple] = {}
    if cfg.use_bn:
        for nm, h in norm_names(cfg).values():
            out[nm + ".running_mean"] = (h,)
            out[nm + ".running_var"] = (h,)
            out[nm + ".num_batches_tracked"] = ()
    return out


def dropout_names(cfg: Cfg) -> List[str]:
    """RNG consumption order of the dropout masks."""
    if cfg.dr_rate <= 0:
        return []
    n = len(cfg.hidden_layer_sizes)
    names = [f"enc.D{i}" for i in range(n)] + [f"dec.D{i}" for i in range(n)]
    names += [f"cls.D{i}" for i in range(len(cfg.classifier_hidden_layer_sizes))]
    return names


# ----------------------------------------------------------------------------
# layer block: linear (+bias) (+cond) -> LayerNorm(no affine) -> ReLU -> dropout
# celligner2/model/modules.py:77-107, 124-130, 298-312
# ----------------------------------------------------------------------------
def _block_fwd(a, W, b, cond, Wc, mask, p, use_ln=True, bn=None):
    """bn: None, or dict(weight, bias, running_mean, running_var, batch_stats, track, momentum) for the
    BatchNorm1d(affine=True, eps 1e-5, momentum 0.1) variant (torch.nn.functional.batch_norm semantics: biased
    batch variance normalises, the UNBIASED one feeds running_var).  bn["new_running"] is set to the updated
    (running_mean, running_var) when batch statistics are tracked."""
    h = a @ W.T
    if b is not None:
        h = h + b
    if Wc is not None:
        h = h + cond @ Wc.T  # multi-hot row cast to float (modules.py:324,487)
    dt = h.dtype.type
    if bn is not None:
        if bn["batch_stats"]:
            n = h.shape[0]
            mean = h.mean(0, keepdims=True)
            var = ((h - mean) ** 2).mean(0, keepdims=True)
            if bn.get("track", True):
                m = dt(bn.get("momentum", 0.1))
                bn["new_running"] = ((1 - m) * bn["running_mean"] + m * mean[0],
                                     (1 - m) * bn["running_var"] + m * var[0] * dt(n / max(n - 1, 1)))
        else:
            mean, var = bn["running_mean"][None, :], bn["running_var"][None, :]
        rstd = 1.0 / np.sqrt(var + dt(1e-5))
        yhat = (h - mean) * rstd
        pre = yhat * bn["weight"] + bn["bias"]
    elif use_ln:
        mean = h.mean(1, keepdims=True)
        var = ((h - mean) ** 2).mean(1, keepdims=True)  # biased, eps 1e-5
        rstd = 1.0 / np.sqrt(var + dt(1e-5))
        pre = yhat = (h - mean) * rstd
    else:
        rstd = np.ones((h.shape[0], 1), h.dtype)
        pre = yhat = h
    r = np.maximum(pre, 0)
    if mask is not None:
        out = r * mask / dt(1.0 - p)
    else:
        out = r
    return out, dict(a=a, cond=cond, yhat=yhat, rstd=rstd, mask=mask, p=p, use_ln=use_ln, bn=bn, pre=pre)


def _block_bwd(dout, cache, W, has_bias, has_cond, need_da=True):
    """Returns (da, dW, db, dWc, dgamma, dbeta); the last two are None without BatchNorm."""
    a, cond, yhat, rstd, mask, p = (cache[k] for k in ("a", "cond", "yhat", "rstd", "mask", "p"))
    bn = cache["bn"]
    if mask is not None:
        dout = dout * mask / dout.dtype.type(1.0 - p)
    dy = dout * (cache["pre"] > 0)
    dgam = dbet = None
    if bn is not None:
        dgam, dbet = (dy * yhat).sum(0), dy.sum(0)
        dyh = dy * bn["weight"]
        if bn["batch_stats"]:
            m1 = dyh.mean(0, keepdims=True)
            m2 = (dyh * yhat).mean(0, keepdims=True)
            dh = (dyh - m1 - yhat * m2) * rstd
        else:
            dh = dyh * rstd
    elif cache["use_ln"]:
        m1 = dy.mean(1, keepdims=True)
        m2 = (dy * yhat).mean(1, keepdims=True)
        dh = (dy - m1 - yhat * m2) * rstd
    else:
        dh = dy
    dW = dh.T @ a
    db = dh.sum(0) if has_bias else None
    dWc = dh.T @ cond if has_cond else None
    da = dh @ W if need_da else None
    return da, dW, db, dWc, dgam, dbet


# ----------------------------------------------------------------------------
# NB dispersion quirk: celligner2/model/_utils.py:8-24 called with the 2-D
# multi-hot matrix (celligner2.py:304-306): scatter_ of the VALUES as indices.
# ----------------------------------------------------------------------------
def quirk_onehot(batch, C, dtype):
    batch = np.asarray(batch)
    if batch.min() < 0 or batch.max() >= C:
        raise IndexError("scatter index out of range (reference raises too)")
    q = np.zeros((batch.shape[0], C), dtype)
    np.put_along_axis(q, batch.astype(np.int64), 1, axis=1)
    return q


def nb_loss(o, x, goodloc, sizefactor, theta, batch):
    """celligner2.py:298-309 + losses.py:27-65.  Returns (recon, d_o, d_theta)
    for unit upstream gradient."""
    dt = o.dtype
    B, G = o.shape
    C = theta.shape[1]
    eps = dt.type(1e-8)
    omax = o.max(1, keepdims=True)
    e = np.exp(o - omax)
    p = e / e.sum(1, keepdims=True)
    mu = p * sizefactor[:, None]
    q = quirk_onehot(batch, C, dt)
    t = q @ theta.T
    th = np.exp(t)
    l_tm = np.log(th + mu + eps)
    ll = (
        th * (np.log(th + eps) - l_tm)
        + x * (np.log(mu + eps) - l_tm)
        + gammaln(x + th)
        - gammaln(th)
        - gammaln(x + 1)
    ).astype(dt)
    m = goodloc.astype(dt)
    recon = -(ll * m).sum(-1).mean()
    # backward
    dll_dmu = x / (mu + eps) - (th + x) / (th + mu + eps)
    dll_dth = (
        np.log(th + eps) - l_tm + th / (th + eps) - (th + x) / (th + mu + eps)
        + digamma(x + th) - digamma(th)
    ).astype(dt)
    gp = -(m / dt.type(B)) * sizefactor[:, None] * dll_dmu
    d_o = p * (gp - (p * gp).sum(1, keepdims=True))
    d_t = -(m / dt.type(B)) * th * dll_dth
    d_theta = d_t.T @ q
    return recon, d_o, d_theta


def _softplus(v):
    return np.logaddexp(v, 0)


def _sigmoid(v):
    return 0.5 * (1.0 + np.tanh(0.5 * v))


def zinb_loss(o, pi, x, goodloc, sizefactor, theta, batch):
    """celligner2.py:286-296 + losses.py:68-124.  o: mean-head logits (softmax inside), pi: dropout logits.
    Returns (recon, d_o, d_pi, d_theta) for unit upstream gradient.
      x < eps : ll = softplus(ptl) - softplus(-pi),   ptl = -pi + theta (log(theta+eps) - log(theta+mu+eps))
      x > eps : ll = NB log-likelihood - pi - softplus(-pi)"""
    dt = o.dtype
    B, G = o.shape
    C = theta.shape[1]
    eps = dt.type(1e-8)
    omax = o.max(1, keepdims=True)
    e = np.exp(o - omax)
    p = e / e.sum(1, keepdims=True)
    mu = p * sizefactor[:, None]
    q = quirk_onehot(batch, C, dt)
    th = np.exp(q @ theta.T)
    l_te = np.log(th + eps)
    l_tm = np.log(th + mu + eps)
    sp = _softplus(-pi)
    ptl = -pi + th * (l_te - l_tm)
    case_zero = _softplus(ptl) - sp
    case_nz = (-sp + ptl + x * (np.log(mu + eps) - l_tm) + gammaln(x + th) - gammaln(th) - gammaln(x + 1))
    z0 = (x < eps).astype(dt)
    z1 = (x > eps).astype(dt)
    ll = (z0 * case_zero + z1 * case_nz).astype(dt)
    m = goodloc.astype(dt)
    recon = -(ll * m).sum(-1).mean()
    # backward
    dptl_dmu = -th / (th + mu + eps)
    dptl_dth = (l_te - l_tm) + th / (th + eps) - th / (th + mu + eps)
    s_ptl = _sigmoid(ptl)
    dll_dmu = z0 * s_ptl * dptl_dmu + z1 * (x / (mu + eps) - (th + x) / (th + mu + eps))
    dll_dth = z0 * s_ptl * dptl_dth + z1 * (dptl_dth - x / (th + mu + eps) + digamma(x + th) - digamma(th))
    dll_dpi = z0 * (-s_ptl + _sigmoid(-pi)) + z1 * (-_sigmoid(pi))
    w = -(m / dt.type(B))
    gp = w * sizefactor[:, None] * dll_dmu
    d_o = p * (gp - (p * gp).sum(1, keepdims=True))
    d_pi = (w * dll_dpi).astype(dt)
    d_t = w * th * dll_dth
    d_theta = d_t.T @ q
    return recon, d_o.astype(dt), d_pi, d_theta.astype(dt)


def mse_loss(o, x, goodloc):
    """Intent of celligner2.py:283-285 (the shipped expression raises TypeError on
    torch 2.x: defined deviation): ReLU output, masked
    row-sum of squared error, batch mean."""
    dt = o.dtype
    B = o.shape[0]
    r = np.maximum(o, 0)
    m = goodloc.astype(dt)
    recon = (m * (r - x) ** 2).sum(-1).mean()
    d_o = (dt.type(2.0) / dt.type(B)) * m * (r - x) * (o > 0)
    return recon, d_o


# ----------------------------------------------------------------------------
# reparameterisation + KL: othermodels/base/_base.py:464-480, celligner2.py:311-322
# ----------------------------------------------------------------------------
def reparam(mu, lv, eps_noise):
    v = np.exp(lv) + mu.dtype.type(1e-4)
    return mu + np.sqrt(v) * eps_noise


def kl_loss(mu, lv, use_own_kl):
    dt = mu.dtype
    B = mu.shape[0]
    v = np.exp(lv) + dt.type(1e-4)
    if use_own_kl:
        kl = dt.type(-0.5) * (1 + lv - mu ** 2 - v).sum()
        dmu = mu.copy()
        dlv = dt.type(-0.5) * (1 - np.exp(lv))
    else:
        kl = (dt.type(0.5) * (v + mu ** 2 - 1 - np.log(v))).sum(1).mean()
        dmu = mu / dt.type(B)
        dlv = dt.type(0.5) * (1 - 1 / v) * np.exp(lv) / dt.type(B)
    return kl, dmu, dlv


# ----------------------------------------------------------------------------
# MMD: losses.py:127-249, _utils.py:27-36
# ----------------------------------------------------------------------------
def partition_index(batch, main_partition=None):
    """Row-index lists in the order `partition` returns its tensors
    (_utils.py:27-36): unique rows ascending (torch.unique(dim=0)), rows whose
    main column is non-zero concatenated and placed first."""
    batch = np.asarray(batch)
    uniq = np.unique(batch, axis=0)
    res, main = [], []
    for u in uniq:
        idx = np.nonzero((batch == u).sum(1) == batch.shape[1])[0]
        if main_partition is not None and u[main_partition]:
            main.append(idx)
        else:
            res.append(idx)
    if main_partition is not None:
        if not main:
            raise RuntimeError("torch.cat of an empty list (reference raises too)")
        return [np.concatenate(main)] + res
    return res


def mmd_pairs(n_parts, sizes, n_conditions, boundary, has_main):
    """(i, j) pairs `mmd` sums over (losses.py:232-248)."""
    pairs = []
    if boundary is not None:
        for i in range(boundary):
            for j in range(boundary, n_conditions):
                if i >= n_parts or j >= n_parts:
                    raise IndexError("list index out of range (reference raises too)")
                if sizes[i] < 2 or sizes[j] < 2:
                    continue
                pairs.append((i, j))
    else:
        for i in range(n_parts):
            if sizes[i] < 1:
                continue
            for j in range(i):
                if sizes[j] < 1:
                    continue
                if has_main and j != 0:
                    continue
                pairs.append((i, j))
    return pairs


def _kernel_sums(x, y, want_grad):
    """sum K(x,y) over all pairs and d(sum)/dx, d(sum)/dy (losses.py:127-160)."""
    dt = x.dtype
    d = ((x[:, None, :] - y[None, :, :]) ** 2).sum(-1)
    ksum = dt.type(0)
    w = np.zeros_like(d)
    for a in MMD_ALPHAS:
        beta = dt.type(1.0) / (dt.type(2.0) * dt.type(a))
        e = np.exp(-beta * d)
        ksum = ksum + e.sum()
        if want_grad:
            w += 2 * beta * e
    if not want_grad:
        return ksum, None, None
    diff = x[:, None, :] - y[None, :, :]
    gx = -(w[:, :, None] * diff).sum(1)
    gy = (w[:, :, None] * diff).sum(0)
    return ksum, gx, gy


def mmd_loss(feat, batch, n_conditions, beta, boundary, main_partition):
    """losses.py:211-249.  Returns (mmd, d_feat) for unit upstream gradient."""
    dt = feat.dtype
    parts = partition_index(batch, main_partition)
    sizes = [len(p) for p in parts]
    pairs = mmd_pairs(len(parts), sizes, n_conditions, boundary, main_partition is not None)
    loss = dt.type(0)
    dfeat = np.zeros_like(feat)
    for (i, j) in pairs:
        X, Y = feat[parts[i]], feat[parts[j]]
        n, m = dt.type(len(X)), dt.type(len(Y))
        kxx, gxx_a, gxx_b = _kernel_sums(X, X, True)
        kyy, gyy_a, gyy_b = _kernel_sums(Y, Y, True)
        kxy, gxy_x, gxy_y = _kernel_sums(X, Y, True)
        loss = loss + kxx / (n * n) + kyy / (m * m) - 2 * kxy / (n * m)
        np.add.at(dfeat, parts[i], (gxx_a + gxx_b) / (n * n) - 2 * gxy_x / (n * m))
        np.add.at(dfeat, parts[j], (gyy_a + gyy_b) / (m * m) - 2 * gxy_y / (n * m))
    b = dt.type(beta)
    return b * loss, b * dfeat


# ----------------------------------------------------------------------------
# classifier loss: losses.py:318-333
# ----------------------------------------------------------------------------
def classifier_hb_loss(u, y, betaclass):
    """Returns (loss, d_u).  `y` is NOT modified here; the reference mutates the
    caller's tensor (-1 -> 0), which the host mirror reproduces."""
    dt = u.dtype
    w = (y != -1).astype(dt)
    yt = np.where(y == -1, 0, y).astype(dt)
    sp = np.maximum(u, 0) + np.log1p(np.exp(-np.abs(u)))
    ce = (w * (sp - yt * u)).sum()
    if u.shape[1] == 0:
        return dt.type(0) * ce, np.zeros_like(u)
    um = u.max(1, keepdims=True)
    lse = um + np.log(np.exp(u - um).sum(1, keepdims=True))
    logp = u - lse
    p = np.exp(logp)
    H = -(p * logp).sum(1, keepdims=True)
    bc = dt.type(betaclass)
    loss = bc * ce + (1 - bc) * H.sum()
    sig = 1 / (1 + np.exp(-u))
    du = bc * w * (sig - yt) + (1 - bc) * (-p * (logp + H))
    return loss, du


# ----------------------------------------------------------------------------
# full step: celligner2.py:259-374 forward + explicit backward
# ----------------------------------------------------------------------------
def forward_backward(params, cfg: Cfg, batch, eps_noise, masks=None,
                     gscale=(1.0, 1.0, 1.0, 1.0), want_grads=True, train=True, buffers=None, bn_untracked=()):
    """batch: dict x [B,G], batch [B,C] int, sizefactor [B], classes [B,P] int or
    None, goodloc [B,G] bool.  gscale = upstream gradient of (recon, kl, mmd, ce):
    the trainer uses (1, alpha, 1, 1) (semisupervised.py:66).
    use_bn: `buffers` holds the BatchNorm running statistics (buffer_shapes); the updated ones come back as
    out["buffers"].  bn_untracked: norm-module names whose running statistics are not tracked -- the trainer's
    rule for BatchNorm layers frozen by surgery (trainer.py:307-312: batch statistics still normalise).
    Returns dict(losses=[4], grads={name: array}, z=..., mu=..., lv=...)."""
    P = params
    dt = P["encoder.FC.L0.expr_L.weight"].dtype
    x = batch["x"].astype(dt)
    cond_i = np.asarray(batch["batch"])
    cond = cond_i.astype(dt)
    has_cond = cfg.n_conditions > 0
    masks = masks or {}
    p = cfg.dr_rate
    nh = len(cfg.hidden_layer_sizes)

    def mk(name):
        return masks.get(name) if (p > 0 and train) else None

    nnames = norm_names(cfg)
    new_buffers = {k: np.array(v) for k, v in (buffers or {}).items()}
    bns = {}

    def bn_of(key):
        if not cfg.use_bn:
            return None
        nm = nnames[key][0]
        bns[key] = dict(name=nm, weight=P[nm + ".weight"], bias=P[nm + ".bias"],
                        running_mean=np.asarray(buffers[nm + ".running_mean"]).astype(dt),
                        running_var=np.asarray(buffers[nm + ".running_var"]).astype(dt),
                        batch_stats=train, track=train and nm not in bn_untracked)
        return bns[key]

    if cfg.applylog and cfg.recon_loss != "mse":
        xin = np.log(1 + x)  # celligner2.py:269-274
    else:
        xin = x
    # ---- encoder (modules.py:322-332)
    caches = {}
    a = xin
    for i in range(nh):
        if i == 0:
            a, caches["enc0"] = _block_fwd(
                a, P["encoder.FC.L0.expr_L.weight"], P["encoder.FC.L0.expr_L.bias"], cond,
                P.get("encoder.FC.L0.cond_L.weight") if has_cond else None, mk("enc.D0"), p, cfg.use_ln,
                bn_of("enc0"))
        else:
            a, caches[f"enc{i}"] = _block_fwd(
                a, P[f"encoder.FC.L{i}.weight"], P[f"encoder.FC.L{i}.bias"], None, None,
                mk(f"enc.D{i}"), p, cfg.use_ln, bn_of(f"enc{i}"))
    henc = a
    mu = henc @ P["encoder.mean_encoder.weight"].T + P["encoder.mean_encoder.bias"]
    lv = henc @ P["encoder.log_var_encoder.weight"].T + P["encoder.log_var_encoder.bias"]
    z = reparam(mu, lv, eps_noise.astype(dt))
    # ---- decoder (modules.py:484-513)
    d, caches["dec0"] = _block_fwd(
        z, P["decoder.FirstL.L0.expr_L.weight"], None, cond,
        P.get("decoder.FirstL.L0.cond_L.weight") if has_cond else None, mk("dec.D0"), p, cfg.use_ln, bn_of("dec0"))
    y1 = d
    for i in range(1, nh):
        d, caches[f"dec{i}"] = _block_fwd(
            d, P[f"decoder.HiddenL.L{i}.weight"], None, None, None, mk(f"dec.D{i}"), p, cfg.use_ln,
            bn_of(f"dec{i}"))
    head = "recon_decoder" if cfg.recon_loss == "mse" else "mean_decoder"
    o = d @ P[f"decoder.{head}.0.weight"].T + P[f"decoder.{head}.0.bias"]
    # ---- classifier (modules.py:619-633), only when classes are given
    classes = batch.get("classes")
    ncl = len(cfg.classifier_hidden_layer_sizes)
    if classes is not None:
        c, caches["cls0"] = _block_fwd(z, P["classifier.FirstL.L0.weight"], None, None, None,
                                       mk("cls.D0"), p, cfg.use_ln, bn_of("cls0"))
        for i in range(1, ncl):
            c, caches[f"cls{i}"] = _block_fwd(c, P[f"classifier.HiddenL.L{i}.weight"], None, None,
                                              None, mk(f"cls.D{i}"), p, cfg.use_ln, bn_of(f"cls{i}"))
        u = c @ P["classifier.classifier.0.weight"].T + P["classifier.classifier.0.bias"]
    # ---- losses
    g = {}
    if cfg.recon_loss == "nb":
        recon, d_o, d_theta = nb_loss(o, x, batch["goodloc"], batch["sizefactor"].astype(dt),
                                      P["theta"], cond_i)
        g["theta"] = d_theta
    elif cfg.recon_loss == "mse":
        recon, d_o = mse_loss(o, xin, batch["goodloc"])
    elif cfg.recon_loss == "zinb":
        pi = d @ P["decoder.dropout_decoder.weight"].T + P["decoder.dropout_decoder.bias"]
        recon, d_o, d_pi, d_theta = zinb_loss(o, pi, x, batch["goodloc"], batch["sizefactor"].astype(dt),
                                              P["theta"], cond_i)
        g["theta"] = d_theta
    else:
        raise NotImplementedError(cfg.recon_loss)
    kl, dmu_kl, dlv_kl = kl_loss(mu, lv, cfg.use_own_kl)
    mmd_val, d_mmd = dt.type(0), None
    if cfg.use_mmd:
        feat = z if cfg.mmd_on == "z" else y1
        mmd_val, d_mmd = mmd_loss(feat, cond_i, cfg.n_conditions, cfg.beta, cfg.mmd_boundary,
                                  cfg.main_dataset)
    ce, du = dt.type(0), None
    if classes is not None:
        ce, du = classifier_hb_loss(u, np.asarray(classes), cfg.betaclass)
    out = dict(losses=np.array([recon, kl, mmd_val, ce], dtype=np.float64), z=z, mu=mu, lv=lv,
               y1=y1, logits=o)
    if classes is not None:
        out["class_logits"] = u
    for bn in bns.values():
        if "new_running" in bn:
            nm = bn["name"]
            new_buffers[nm + ".running_mean"], new_buffers[nm + ".running_var"] = bn["new_running"]
            new_buffers[nm + ".num_batches_tracked"] = np.asarray(buffers[nm + ".num_batches_tracked"]) + 1
    out["buffers"] = new_buffers
    out["caches"] = caches  # per-block intermediates (a, yhat, pre, rstd, ...), for diagnostics
    if not want_grads:
        return out
    gr, gk, gm, gc = [dt.type(v) for v in gscale]

    def bwd(key, dout, W, has_bias, has_c, need_da=True):
        da_, dW_, db_, dWc_, dgam, dbet = _block_bwd(dout, caches[key], W, has_bias, has_c, need_da)
        if dgam is not None:
            g[nnames[key][0] + ".weight"], g[nnames[key][0] + ".bias"] = dgam, dbet
        return da_, dW_, db_, dWc_

    # ---- backward: decoder
    d_o = gr * d_o
    if "theta" in g:
        g["theta"] = gr * g["theta"]
    g[f"decoder.{head}.0.weight"] = d_o.T @ d
    g[f"decoder.{head}.0.bias"] = d_o.sum(0)
    dd = d_o @ P[f"decoder.{head}.0.weight"]
    if cfg.recon_loss == "zinb":
        d_pi = gr * d_pi
        g["decoder.dropout_decoder.weight"] = d_pi.T @ d
        g["decoder.dropout_decoder.bias"] = d_pi.sum(0)
        dd = dd + d_pi @ P["decoder.dropout_decoder.weight"]
    for i in range(nh - 1, 0, -1):
        dd, dW, _, _ = bwd(f"dec{i}", dd, P[f"decoder.HiddenL.L{i}.weight"], False, False)
        g[f"decoder.HiddenL.L{i}.weight"] = dW
    if cfg.use_mmd and cfg.mmd_on != "z":
        dd = dd + gm * d_mmd
    dz, dW, _, dWc = bwd("dec0", dd, P["decoder.FirstL.L0.expr_L.weight"], False, has_cond)
    g["decoder.FirstL.L0.expr_L.weight"] = dW
    if has_cond:
        g["decoder.FirstL.L0.cond_L.weight"] = dWc
    if cfg.use_mmd and cfg.mmd_on == "z":
        dz = dz + gm * d_mmd
    # ---- backward: classifier
    if classes is not None:
        du = gc * du
        g["classifier.classifier.0.weight"] = du.T @ c
        g["classifier.classifier.0.bias"] = du.sum(0)
        dc = du @ P["classifier.classifier.0.weight"]
        for i in range(ncl - 1, 0, -1):
            dc, dW, _, _ = bwd(f"cls{i}", dc, P[f"classifier.HiddenL.L{i}.weight"], False, False)
            g[f"classifier.HiddenL.L{i}.weight"] = dW
        dzc, dW, _, _ = bwd("cls0", dc, P["classifier.FirstL.L0.weight"], False, False)
        g["classifier.FirstL.L0.weight"] = dW
        dz = dz + dzc
    # ---- reparam + KL backward
    v = np.exp(lv) + dt.type(1e-4)
    dmu = dz + gk * dmu_kl
    dlv = dz * eps_noise.astype(dt) * np.exp(lv) / (2 * np.sqrt(v)) + gk * dlv_kl
    g["encoder.mean_encoder.weight"] = dmu.T @ henc
    g["encoder.mean_encoder.bias"] = dmu.sum(0)
    g["encoder.log_var_encoder.weight"] = dlv.T @ henc
    g["encoder.log_var_encoder.bias"] = dlv.sum(0)
    da = dmu @ P["encoder.mean_encoder.weight"] + dlv @ P["encoder.log_var_encoder.weight"]
    for i in range(nh - 1, 0, -1):
        da, dW, db, _ = bwd(f"enc{i}", da, P[f"encoder.FC.L{i}.weight"], True, False)
        g[f"encoder.FC.L{i}.weight"] = dW
        g[f"encoder.FC.L{i}.bias"] = db
    _, dW, db, dWc = bwd("enc0", da, P["encoder.FC.L0.expr_L.weight"], True, has_cond, need_da=False)
    g["encoder.FC.L0.expr_L.weight"] = dW
    g["encoder.FC.L0.expr_L.bias"] = db
    if has_cond:
        g["encoder.FC.L0.cond_L.weight"] = dWc
    out["grads"] = g
    return out


# ----------------------------------------------------------------------------
# torch.optim.Adam as the trainer builds it (trainer.py:262-266): L2 decay.
# ----------------------------------------------------------------------------
def adam_step(params, grads, state, lr=1e-3, betas=(0.9, 0.999), eps=0.01, weight_decay=0.04,
              clip_value=0.0, names=None):
    """In place on params/state.  state: {name: (exp_avg, exp_avg_sq)}, state['step']."""
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    bc1 = 1 - b1 ** t
    bc2 = 1 - b2 ** t
    for k in (names if names is not None else grads.keys()):
        if grads.get(k) is None:
            continue  # parameter that took no part in the graph: Adam skips it
        p = params[k]
        dt = p.dtype
        gk = grads[k].astype(dt)
        if clip_value > 0:
            gk = np.clip(gk, -clip_value, clip_value)
        gk = gk + dt.type(weight_decay) * p
        if k not in state:
            state[k] = (np.zeros_like(p), np.zeros_like(p))
        m, v = state[k]
        m *= dt.type(b1)
        m += dt.type(1 - b1) * gk
        v *= dt.type(b2)
        v += dt.type(1 - b2) * gk * gk
        denom = np.sqrt(v) / dt.type(np.sqrt(bc2)) + dt.type(eps)
        p -= dt.type(lr / bc1) * (m / denom)
    return params, state


def calc_alpha_coeff(alpha_kl, epoch, it, iters_per_epoch, alpha_epoch_anneal=None,
                     alpha_iter_anneal=None):
    """trainer.py:231-253."""
    if alpha_epoch_anneal is not None:
        return min(alpha_kl * epoch / alpha_epoch_anneal, alpha_kl)
    if alpha_iter_anneal is not None:
        return min(alpha_kl * (epoch * iters_per_epoch + it) / alpha_iter_anneal, alpha_kl)
    return alpha_kl


def init_params(cfg: Cfg, seed=0, dtype=np.float32):
    """Random parameters of the right shapes (synthetic; NOT the reference's init)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if k == "theta":
            out[k] = rng.standard_normal(shp).astype(dtype)
        elif len(shp) == 1 and k.endswith("weight"):  # BatchNorm scale
            out[k] = np.ones(shp, dtype)
        elif k.endswith("bias"):
            out[k] = (0.1 * rng.standard_normal(shp)).astype(dtype)
        else:
            fan_in = shp[1]
            out[k] = (rng.uniform(-1, 1, shp) / np.sqrt(max(fan_in, 1))).astype(dtype)
    return out
