import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from celligner2_amd import ops
from oracle import c2_oracle as orc
dev = "cuda:0"
B, G, C = 4, 19000, 2
rng = np.random.default_rng(1)
o = rng.standard_normal((B, G)).astype(np.float32)
x = rng.poisson(np.exp(rng.normal(1, 1, (1, G))) * rng.uniform(0.5, 2, (B, 1))).astype(np.float32)
good = rng.uniform(size=(B, G)) > 0.05
sf = x.sum(1).astype(np.float32)
theta = rng.standard_normal((G, C)).astype(np.float32)
cond = np.zeros((B, C), np.int64); cond[np.arange(B), rng.integers(0, C, B)] = 1
r64, do64, dth64 = orc.nb_loss(o.astype(np.float64), x.astype(np.float64), good, sf.astype(np.float64), theta.astype(np.float64), cond)
T = lambda a: torch.tensor(a).to(dev)
loss = torch.zeros(1, device=dev); dl = torch.empty(B, G, device=dev); dth = torch.empty(G, C, device=dev)
ws = torch.empty(ops.recon_nb_ws(B, G, C), device=dev); st = torch.zeros(1, dtype=torch.int32, device=dev)
lgx = ops.nb_row_const(T(x), T(good))
for rep in range(4):
    dl.fill_(float("nan")); loss.zero_()
    ops.recon_nb(T(o), T(x), T(good), T(sf), T(theta), T(cond), 1.0 / B, loss, dl, dth, ws, st, lgx1=lgx)
    d = dl.cpu().numpy()
    err = np.abs(d - do64)
    bad = np.argwhere(~(err <= 1e-4 * np.abs(do64).max()))
    print("rep", rep, "loss", loss.item(), "n bad", len(bad), "nan", int(np.isnan(d).sum()), "rows", sorted(set(bad[:, 0])), "cols", bad[:24, 1].tolist())
    if len(bad):
        b, j = bad[0]
        print("   got", d[b, j:j + 4], "want", do64[b, j:j + 4])
