"""GPU diagnostic: per-gradient error of the bf16-operand mode against the fp64 oracle."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import c2_oracle as orc
from celligner2_amd.model import Celligner2

def run(G, hidden, L, C, P, B, precision, seed=0, mmd=True):
    rng = np.random.default_rng(seed)
    cfg = orc.Cfg(input_dim=G, hidden_layer_sizes=hidden, latent_dim=L, n_conditions=C, n_predictors=P,
                  recon_loss="nb", use_mmd=mmd, beta=100.0)
    params = orc.init_params(cfg, seed, np.float32)
    depth = rng.uniform(0.5, 2.0, (B, 1)); gene = np.exp(rng.normal(1.0, 1.5, (1, G)))
    x = rng.poisson(depth * gene).astype(np.float32)
    cond = np.zeros((B, C), np.int64); cond[np.arange(B), rng.integers(0, C, B)] = 1
    cls = np.zeros((B, P), np.int64); cls[np.arange(B), rng.integers(0, P, B)] = 1; cls[rng.uniform(size=B) < 0.2] = -1
    batch = dict(x=x, batch=cond, sizefactor=x.sum(1).astype(np.float32), goodloc=rng.uniform(size=(B, G)) > 0.005, classes=cls)
    eps = rng.standard_normal((B, L)).astype(np.float32)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    ref = orc.forward_backward(p64, cfg, batch, eps.astype(np.float64), gscale=(1, 1e-3, 1, 1))
    m = Celligner2(G, hidden, L, 0.0, False, "nb", conditions=[f"c{i}" for i in range(C)], use_mmd=mmd, beta=100.0,
                   predictors=[f"p{i}" for i in range(P)], precision=precision)
    m.load_state_dict({k: torch.tensor(v) for k, v in params.items()}); m.cuda().train()
    m._forced_noise = dict(eps=torch.tensor(eps).cuda(), masks={})
    tb = {k: torch.tensor(v).cuda() for k, v in batch.items()}
    r, k, mm, ce = m(**tb); (r + 1e-3 * k + mm + ce).backward()
    print(f"--- G={G} hidden={hidden} B={B} {precision}: losses", [round(v.item(), 4) for v in (r, k, mm, ce)], "ref", np.round(ref["losses"], 4))
    for n, p in m.named_parameters():
        g = p.grad.double().cpu().numpy(); gr = ref["grads"][n]
        l2 = np.linalg.norm(g - gr) / max(np.linalg.norm(gr), 1e-30)
        mx = np.abs(g - gr).max() / max(np.abs(gr).max(), 1e-30)
        print(f"{n:40s} l2rel={l2:.2e} maxrel={mx:.2e}")

if __name__ == "__main__":
    from celligner2_amd.engine import StepEngine
    StepEngine.WIDE = 64
    run(96, [32, 16], 8, 2, 5, 24, "bf16")
    StepEngine.WIDE = 2048
    run(4096, [512, 256], 16, 3, 10, 512, "bf16")
