"""Time c2_recon_nb_fwd_bwd at the benchmark shape (4096 x 19000, bf16 dlogits, rows read through the row index,
data-only lgamma term precomputed) and check it against the fp64 oracle on a 64-row slice."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from celligner2_amd import ops  # noqa: E402

B, G, C, N = 4096, 19000, 3, 16384
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
gene = torch.exp(torch.randn(1, G, device=dev, generator=g) * 1.5 + 1.5)
X = torch.poisson(torch.empty(N, 1, device=dev).uniform_(0.5, 2.0, generator=g) * gene, generator=g)
good = torch.rand(N, G, device=dev, generator=g) > 0.005
logits = torch.randn(B, G, device=dev, generator=g)
theta = torch.randn(G, C, device=dev, generator=g)
cond = torch.zeros(B, C, dtype=torch.int64, device=dev)
cond[torch.arange(B), torch.randint(0, C, (B,), device=dev)] = 1
ridx = torch.randint(0, N, (B,), device=dev)
sf = X.sum(1)[ridx].contiguous()
lgx = ops.nb_row_const(X, good)[ridx].contiguous()
loss = torch.zeros(1, device=dev)
dl = torch.empty(B, G, device=dev, dtype=torch.bfloat16)
dth = torch.empty(G, C, device=dev)
ws = torch.empty(ops.recon_nb_ws(B, G, C), device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)


def run():
    ops.recon_nb(logits, X, good, sf, theta, cond, 1.0 / B, loss, dl, dth, ws, st, lgx1=lgx, row_index=ridx)


for _ in range(5):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
s.record()
for _ in range(n):
    run()
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / n
byt = B * G * (4 + 4 + 1 + 2)
print(f"recon_nb {B}x{G}: {us:.1f} us per call (table + loss kernel + dtheta reduction), {byt / us / 1e6:.2f} TB/s algorithmic")
# oracle check on a slice
from oracle import c2_oracle as orc  # noqa: E402
nb = 64
loss.zero_()
dl32 = torch.empty(nb, G, device=dev)
ops.recon_nb(logits[:nb].contiguous(), X, good, sf[:nb].contiguous(), theta, cond[:nb].contiguous(), 1.0 / nb, loss, dl32,
             dth, torch.empty(ops.recon_nb_ws(nb, G, C), device=dev), st, lgx1=lgx[:nb].contiguous(), row_index=ridx[:nb].contiguous())
xs = X[ridx[:nb]].double().cpu().numpy()
gs = good[ridx[:nb]].cpu().numpy()
r, dlo, dto = orc.nb_loss(logits[:nb].double().cpu().numpy(), xs, gs, sf[:nb].double().cpu().numpy(),
                           theta.double().cpu().numpy(), cond[:nb].cpu().numpy())
print("loss rel err", abs(loss.item() - r) / abs(r), "dlogits", np.abs(dl32.cpu().numpy() - dlo).max() / np.abs(dlo).max(),
      "dtheta", np.abs(dth.cpu().numpy() - dto).max() / np.abs(dto).max())
