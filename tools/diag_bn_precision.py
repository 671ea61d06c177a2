"""Diagnostic (GPU): per-tensor error of the LayerNorm / BatchNorm step in fp32 and bf16 mode against the fp64 oracle
(midsize model of tests/test_step_gpu.py::test_bf16_step_midsize_vs_oracle).  Not part of the product."""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests import golden_util as gu
from oracle import c2_oracle as orc
from celligner2_amd.model import Celligner2
DEV = "cuda:0"
G, hidden, L, C, P, B = 4096, [512, 256], 16, 3, 10, 512
for use_bn in (False, True):
    rng = np.random.default_rng(0)
    cfg = orc.Cfg(input_dim=G, hidden_layer_sizes=hidden, latent_dim=L, n_conditions=C, n_predictors=P,
                  recon_loss="nb", use_mmd=True, beta=100.0, use_bn=use_bn)
    params = orc.init_params(cfg, 0, np.float32)
    buffers = {}
    for k, shp in orc.buffer_shapes(cfg).items():
        buffers[k] = (np.zeros(shp, np.int64) if k.endswith("tracked") else np.ones(shp, np.float32))
    depth = rng.uniform(0.5, 2.0, (B, 1))
    prof = np.exp(rng.normal(1.0, 1.5, (1, G)))
    prof = prof * np.exp(rng.normal(0.0, 1.0, (4, G)))[rng.integers(0, 4, B)]
    x = rng.poisson(depth * prof).astype(np.float32)
    cond = np.zeros((B, C), np.int64); cond[np.arange(B), rng.integers(0, C, B)] = 1
    cls = np.zeros((B, P), np.int64); cls[np.arange(B), rng.integers(0, P, B)] = 1
    batch = dict(x=x, batch=cond, sizefactor=x.sum(1).astype(np.float32), goodloc=np.ones((B, G), bool), classes=cls)
    eps = rng.standard_normal((B, L)).astype(np.float32)
    ref = orc.forward_backward({k: v.astype(np.float64) for k, v in params.items()}, cfg, batch, eps.astype(np.float64),
                               gscale=(1, 1e-3, 1, 1), buffers=buffers)
    for prec in ("fp32", "bf16"):
        m = Celligner2(G, hidden, L, 0.0, False, "nb", use_bn, conditions=[f"c{i}" for i in range(C)], use_mmd=True,
                       beta=100.0, predictors=[f"p{i}" for i in range(P)], precision=prec)
        m.load_state_dict({k: torch.tensor(v) for k, v in {**params, **buffers}.items()})
        m.to(DEV).train()
        m._forced_noise = dict(eps=torch.tensor(eps).to(DEV), masks={})
        r, k, mm, ce = m(**{k_: torch.tensor(v).to(DEV) for k_, v in batch.items()})
        (r + 1e-3 * k + mm + ce).backward()
        eng = m._engine()
        def l2(a, b):
            a = a.double().cpu().numpy(); return np.linalg.norm(a - b) / np.linalg.norm(b)
        print(use_bn, prec, "losses", [round(v.item(), 4) for v in (r, k, mm, ce)], ref["losses"])
        for blk in ("enc0", "enc1", "dec0", "dec1"):
            print("  ", blk, "yhat %.5f" % l2(eng._ws[blk + ".yhat"], ref["caches"][blk]["yhat"]),
                  "dh-norm", float(eng._ws[blk + ".dh"].float().norm()))
        for n in ("encoder.FC.L0.expr_L.weight", "encoder.FC.L1.weight", "encoder.mean_encoder.weight",
                  "decoder.FirstL.L0.expr_L.weight", "decoder.HiddenL.L1.weight", "decoder.mean_decoder.0.weight"):
            print("  ", n, "%.5f" % l2(dict(m.named_parameters())[n].grad, ref["grads"][n]))
