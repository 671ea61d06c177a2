"""Worker of tests/test_parallel_gpu.py: 2 ranks share cuda:0 (gloo backend, the box has one GPU); each runs the
data-parallel step on its half of a global batch; rank 0 also runs the single-process step on the whole batch.
The summed-gradient result must equal the single-process one (fp32 mode, 1e-4)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    from celligner2_amd.model import Celligner2
    from celligner2_amd.parallel import DataParallelStep
    from oracle import c2_oracle as orc

    G, hidden, L, C, P, B = 256, [48, 24], 8, 3, 4, 64
    cfg = orc.Cfg(input_dim=G, hidden_layer_sizes=hidden, latent_dim=L, n_conditions=C, n_predictors=P,
                  recon_loss="nb", use_mmd=True, beta=20.0)
    rng = np.random.default_rng(0)
    params = orc.init_params(cfg, 0, np.float32)
    x = rng.poisson(rng.uniform(0.5, 2, (B, 1)) * np.exp(rng.normal(1, 1, (1, G)))).astype(np.float32)
    cond = np.zeros((B, C), np.int64)
    cond[np.arange(B), rng.integers(0, C, B)] = 1
    cls = np.zeros((B, P), np.int64)
    cls[np.arange(B), rng.integers(0, P, B)] = 1
    cls[::7] = -1
    full = dict(x=x, batch=cond, sizefactor=x.sum(1), goodloc=rng.uniform(size=(B, G)) > 0.01, classes=cls)
    eps = rng.standard_normal((B, L)).astype(np.float32)

    def build():
        m = Celligner2(G, hidden, L, 0.0, False, "nb", conditions=["a", "b", "c"], use_mmd=True, beta=20.0,
                       predictors=list("wxyz"))
        m.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
        return m.cuda().train()

    def run(m, sl):
        m._forced_noise = dict(eps=torch.tensor(eps[sl]).cuda(), masks={})
        r, k, mm, ce = m(**{k_: torch.tensor(v[sl]).cuda() for k_, v in full.items()})
        (r + 1e-3 * k + mm + ce).backward()
        return np.array([r.item(), k.item(), mm.item(), ce.item()]), {n: p.grad.clone() for n, p in m.named_parameters()}

    ref_l = ref_g = None
    if rank == 0:
        ref_l, ref_g = run(build(), slice(0, B))
    m = build()
    comm = DataParallelStep()
    comm.attach_model(m)
    per = B // world
    losses, grads = run(m, slice(rank * per, (rank + 1) * per))
    m._engine().check_status()
    ok = True
    if rank == 0:
        ok = bool(np.allclose(losses, ref_l, rtol=1e-4, atol=1e-5))
        for n in ref_g:
            err = (grads[n] - ref_g[n]).abs().max().item() / max(ref_g[n].abs().max().item(), 1e-30)
            if err > 1e-4:
                ok = False
                print("MISMATCH", n, err)
        print("losses ddp", losses, "single", ref_l)
    flag = torch.tensor([1.0 if ok else 0.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if flag.item() == 1.0 else 1)


if __name__ == "__main__":
    main()
