"""Time c2_recon_zinb_fwd_bwd at the benchmark shape (4096 x 19000, bf16 gradients, rows through the row index)."""
import os
import sys

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
pi = torch.randn(B, G, device=dev, generator=g)
theta = torch.randn(G, C, device=dev, generator=g)
cond = torch.zeros(B, C, dtype=torch.int64, device=dev)
cond[torch.arange(B), torch.randint(0, C, (B,), device=dev)] = 1
ridx = torch.randint(0, N, (B,), device=dev)
sf = X.sum(1)[ridx].contiguous()
loss = torch.zeros(1, device=dev)
dl = torch.empty(B, G, device=dev, dtype=torch.bfloat16)
dp = torch.empty(B, G, device=dev, dtype=torch.bfloat16)
dth = torch.empty(G, C, device=dev)
ws = torch.empty(ops.recon_nb_ws(B, G, C), device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)


def run():
    ops.recon_zinb(logits, pi, X, good, sf, theta, cond, 1.0 / B, loss, dl, dp, dth, ws, st, row_index=ridx)


for _ in range(3):
    run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
s.record()
for _ in range(n):
    run()
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) * 1e3 / n
print(f"recon_zinb {B}x{G}: {us:.1f} us per call, {B * G * (4 + 4 + 4 + 1 + 2 + 2) / us / 1e6:.2f} TB/s algorithmic")
