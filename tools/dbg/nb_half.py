import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from celligner2_amd import ops
DEV = "cuda"
B, G, C = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 19000, 3
rng = np.random.default_rng(0)
x = torch.tensor(rng.poisson(3.0, (B, G)).astype(np.float32)).to(DEV)
good = torch.ones(B, G, dtype=torch.bool, device=DEV)
lg16 = torch.tensor((rng.standard_normal((B, G)) * 2).astype(np.float32)).to(DEV).to(torch.float16)
cond = torch.zeros(B, C, dtype=torch.int64, device=DEV); cond[:, 0] = 1
theta = torch.zeros(G, C, device=DEV)
out = []
for lg in (lg16, lg16.float()):
    loss = torch.zeros(1, device=DEV); dl = torch.empty(B, G, device=DEV, dtype=torch.bfloat16)
    dth = torch.empty(G, C, device=DEV); db = torch.empty(G, device=DEV)
    ws = torch.zeros(ops.recon_nb_ws(B, G, C), device=DEV); st = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.recon_nb(lg, x, good, x.sum(1), theta, cond, 1.0 / B, loss, dl, dth, ws, st, dbias=db)
    out.append((loss.item(), dl.float().cpu().numpy()))
print("loss", out[0][0], out[1][0])
d = np.abs(out[0][1] - out[1][1])
bad = np.argwhere(d > 0)
print("mismatching dlogits entries", len(bad), "of", d.size)
if len(bad):
    print("first", bad[:12].tolist(), "last", bad[-5:].tolist())
    cols = np.unique(bad[:, 1]); print("cols mod 2048 histogram", np.bincount((cols // 4) % 512)[:16], "n cols", len(cols))
    r, c = bad[0]; print(out[0][1][r, c:c + 8], out[1][1][r, c:c + 8])
# which of the two is right: the CPU restatement on the widened logits
from oracle import c2_oracle as orc
o = lg16.float().double().cpu().numpy()
xx = x.double().cpu().numpy()
recon, d_o, _ = orc.nb_loss(o, xx, np.ones((B, G), bool), xx.sum(1), np.zeros((G, C)), cond.cpu().numpy())
print("oracle loss", recon)
for k, nm in enumerate(("half-logit kernel", "fp32-logit kernel")):
    e = np.abs(out[k][1] - d_o)
    print(nm, "max |dlogits - oracle| / max|oracle|", e.max() / np.abs(d_o).max(), "worst at", np.unravel_index(e.argmax(), e.shape))
