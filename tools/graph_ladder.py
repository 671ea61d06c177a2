"""Developer aid: capture/replay ladder -- each rung adds one component to a captured region and compares the replay
with the same work launched the ordinary way.  usage: graph_ladder.py RUNG"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from celligner2_amd import ops  # noqa: E402
from celligner2_amd.model import Celligner2  # noqa: E402
from celligner2_amd.optim import FusedAdam  # noqa: E402
from celligner2_amd.dataset import DeviceDataset  # noqa: E402

rung = int(sys.argv[1])
dev = torch.device("cuda")
print(f"[rung {rung}] start", flush=True)


def capture(fn, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    torch.cuda.synchronize()
    print(f"[rung {rung}] captured", flush=True)
    return g, out


def make(dr=0.1, B=32, G=64):
    torch.manual_seed(3)
    model = Celligner2(G, [32, 16], 8, dr, False, "nb", conditions=["a", "b", "c"], use_mmd=True, beta=3.0,
                       predictors=["x", "y", "z"]).cuda()
    model.rng_mode = "device"
    model.train()
    gen = torch.Generator().manual_seed(0)
    n = 200
    X = torch.poisson(torch.rand(n, G, generator=gen) * 5).to(dev)
    cond = torch.nn.functional.one_hot(torch.randint(0, 3, (n,), generator=gen), 3).to(dev)
    pred = torch.nn.functional.one_hot(torch.randint(0, 3, (n,), generator=gen), 3).to(dev)
    ds = DeviceDataset.from_tensors(X, X.sum(1), torch.ones_like(X, dtype=torch.bool), cond, pred)
    return model, ds


if rung == 1:
    a, b = torch.empty(6, device=dev), torch.empty(6, device=dev)
    l4 = torch.randn(4, device=dev)
    w = torch.tensor([1.0, 0.1, 1.0, 1.0], device=dev)
    sums = torch.zeros(6, dtype=torch.float64, device=dev)
    ops.combine_losses(l4, [1.0, 0.1, 1.0, 1.0], a)
    ops.combine_losses_dev(l4, w, b, sums)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(sums, a.double())
    m = torch.empty(16 * 100 + 3, dtype=torch.uint8, device=dev)
    c = torch.tensor([5], dtype=torch.int64, device=dev)
    ops.dropout_masks_dev(m, 0.3, 1, c, 101)
    torch.cuda.synchronize()
    assert int(c.item()) == 106
    p, g_, m_, v_ = (torch.randn(100, device=dev) for _ in range(4))
    v_.abs_()
    st, blk = ops.adam_state(0, 1e-3, 0.9, 0.999, 0.01, 0.04, 0.0, dev)
    ops.adam_prepare(st, blk)
    ops.adam_step_dev([(p, g_, m_, v_, None)], blk)
    torch.cuda.synchronize()
    assert ops.adam_state_step(st) == 1
elif rung == 2:
    l4 = torch.randn(4, device=dev)
    w = torch.tensor([1.0, 0.1, 1.0, 1.0], device=dev)
    b = torch.empty(6, device=dev)
    sums = torch.zeros(6, dtype=torch.float64, device=dev)
    p, g_, m_, v_ = (torch.randn(1000, device=dev) for _ in range(4))
    v_.abs_()
    st, blk = ops.adam_state(0, 1e-3, 0.9, 0.999, 0.01, 0.04, 0.0, dev)

    def fn():
        ops.combine_losses_dev(l4, w, b, sums)
        ops.adam_prepare(st, blk)
        ops.adam_step_dev([(p, g_, m_, v_, None)], blk)
    g, _ = capture(fn)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert ops.adam_state_step(st) == 5, ops.adam_state_step(st)
    assert abs(sums[0].item() - 5 * l4[0].item()) < 1e-5
elif rung == 3:
    m = torch.empty(16 * 100, dtype=torch.uint8, device=dev)
    c = torch.tensor([0], dtype=torch.int64, device=dev)
    keep = []

    def fn():
        e = torch.randn(32, 8, device=dev)
        ops.dropout_masks_dev(m, 0.3, 1, c, 100)
        return e
    g, e = capture(fn)
    outs = []
    for _ in range(3):
        g.replay()
        outs.append((e.clone(), m.clone()))
    torch.cuda.synchronize()
    assert not torch.equal(outs[0][0], outs[1][0]) and not torch.equal(outs[0][1], outs[1][1])
    assert int(c.item()) == 500
elif rung in (4, 5, 6, 7, 8):
    model, ds = make()
    B = 32
    idx = torch.arange(B, device=dev) * 3
    bufs = {}
    eng0 = model._engine()

    def gather():
        return ds.gather(idx, apply_log=True, bufs=bufs, nb_const=True, resident_rows=True, resident_xin=False,
                         **eng0.input_spec())
    batch = gather()
    w = torch.tensor([1.0, 0.1, 1.0, 1.0], device=dev)
    if rung == 4:
        shapes = model._dropout_shapes(B) + [(f"cls.D{i}", (B, h)) for i, h in enumerate(model.classifier_hidden_layer_sizes)]
        model._forced_noise = dict(eps=torch.randn(B, 8, device=dev),
                                   masks={n: torch.ones(s, dtype=torch.uint8, device=dev) for n, s in shapes})
    if rung in (4, 5):
        def fn():
            with torch.no_grad():
                return model.forward_total(w, **batch)[1]
        g, v = capture(fn)
        g.replay()
        torch.cuda.synchronize()
        print("losses", v.tolist(), flush=True)
        assert torch.isfinite(v).all()
        if rung == 4:
            ref = fn().clone()
            assert torch.equal(ref, v), (ref, v)
    else:
        model.device_noise_counter(True)
        unit = torch.ones((), device=dev)
        lookup, updated = model._adam_hooks()
        opt = FusedAdam([p for p in model.parameters()], lr=1e-3, eps=0.01, weight_decay=0.04, shadow_lookup=lookup,
                        on_updated=updated, capturable=True)

        def fn():
            bt = gather() if rung == 8 else batch
            total, v = model.forward_total(w, **bt)
            opt.zero_grad()
            total.backward(gradient=unit)
            if rung >= 7:
                opt.step()
            return v
        g, v = capture(fn)
        if rung >= 7:
            opt.captured()
        for _ in range(3):
            g.replay()
            if rung >= 7:
                opt.replayed()
        torch.cuda.synchronize()
        print("losses", v.tolist(), flush=True)
        assert torch.isfinite(v).all()
        gsum = sum(float(p.grad.abs().sum()) for p in model.parameters() if p.grad is not None)
        print("grad abs sum", gsum, flush=True)
        assert np.isfinite(gsum) and gsum > 0
elif rung in (9, 10):
    from tests.test_graph_gpu import _run
    # 9: four full batches of 30 per epoch (no short batch); 10: batches of 32 -> 3 full + one of 24
    m, tr = _run(True, 2, batch=30 if rung == 9 else 32)
    print({k: v for k, v in tr.logs.items() if k.startswith("epoch_")}, flush=True)
print(f"[rung {rung}] ok", flush=True)
