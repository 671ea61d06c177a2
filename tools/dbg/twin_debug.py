import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_async_exchange_gpu import _setup, _twin_world
from celligner2_amd.model import Celligner2
from celligner2_amd.optim import FusedAdam
DEV = "cuda:0"

def grads_of(shape, params, batch, eps, world):
    G, hidden, L, C, P, B = shape
    m = Celligner2(G, hidden, L, 0.0, False, "nb", False, conditions=["a", "b", "c"], use_mmd=True, beta=20.0, predictors=list("wxyz"))
    m.load_state_dict({k: torch.tensor(v) for k, v in params.items()}, strict=False)
    m.to(DEV).train()
    if world is not None:
        world.attach_model(m)
    opt = FusedAdam(list(m.parameters()), lr=1e-2, eps=0.01, weight_decay=0.04, grad_sync=world)
    m._forced_noise = dict(eps=torch.tensor(eps).to(DEV), masks={})
    r, k, mm, ce = m(**{k_: torch.tensor(v).to(DEV) for k_, v in batch.items()})
    (r + 1e-3 * k + mm + ce).backward()
    torch.cuda.synchronize()
    if world is not None:
        world.wait_deferred(); torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, m

shape, params, batch, eps = _setup(False)
two = {k: np.concatenate([v, v], 0) for k, v in batch.items()}
ref, _ = grads_of(shape, params, two, np.concatenate([eps, eps], 0), None)
for delay in (0, 30_000_000):
    got, m = grads_of(shape, params, batch, eps, _twin_world(delay))
    print("delay", delay, "buckets", m._engine().buckets)
    for n in ref:
        den = max(ref[n].abs().max().item(), 1e-30)
        e = (got[n] - ref[n]).abs().max().item() / den
        if e > 1e-5:
            print("  ", n, "err", e, "ratio", (got[n].abs().sum() / ref[n].abs().sum()).item())
