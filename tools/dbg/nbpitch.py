"""Experiment (GPU box): the NB loss kernel at 4096 x 19000 with the counts / mask rows 19 000 elements apart against rows
padded to 128-byte multiples (counts 19 008 floats, mask 19 072 bytes); half logits and bf16 dlogits at a pitch of 19 072."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from celligner2_amd import ops
B, G, C, N = 4096, 19000, 3, 16384
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
gene = torch.exp(torch.randn(1, G, device=dev, generator=g) * 1.5 + 1.5)
X0 = torch.poisson(torch.empty(N, 1, device=dev).uniform_(0.5, 2.0, generator=g) * gene, generator=g)
good0 = torch.rand(N, G, device=dev, generator=g) > 0.005
theta = torch.randn(G, C, device=dev, generator=g)
cond = torch.zeros(B, C, dtype=torch.int64, device=dev)
cond[torch.arange(B), torch.randint(0, C, (B,), device=dev)] = 1
ridx = torch.randint(0, N, (B,), device=dev)
sf = X0.sum(1)[ridx].contiguous()
lgx = ops.nb_row_const(X0, good0)[ridx].contiguous()
def pitched(t, ld):
    out = torch.zeros(t.shape[0], ld, dtype=t.dtype, device=dev)[:, :t.shape[1]]
    out.copy_(t)
    return out
lg = pitched(torch.randn(B, G, device=dev, generator=g).half(), 19072)
dl = torch.zeros(B, 19072, device=dev, dtype=torch.bfloat16)[:, :G]
dth, db = torch.empty(G, C, device=dev), torch.empty(G, device=dev)
ws = torch.empty(ops.recon_nb_ws(B, G, C), device=dev)
st = torch.zeros(1, dtype=torch.int32, device=dev)
loss = torch.zeros(1, device=dev)
for ldx, ldg in ((19000, 19000), (19008, 19008), (19008, 19072), (19072, 19072)):
    X, good = pitched(X0, ldx), pitched(good0, ldg)
    run = lambda: ops.recon_nb(lg, X, good, sf, theta, cond, 1.0 / B, loss, dl, dth, ws, st, lgx1=lgx, row_index=ridx, dbias=db)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): run()
    e.record(); torch.cuda.synchronize()
    print(f"counts pitch {ldx}, mask pitch {ldg}: {s.elapsed_time(e) * 20:.1f} us per call", flush=True)
