"""Fixed cost of the gene-width GEMMs at the reference's batch of 128 (GPU box only)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from celligner2_amd import ops


def bench(fn, reps=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


dev, dt = "cuda", torch.bfloat16
B, H = 128, 256
for G in (19000, 4096, 1024):
    x = torch.randn(B, G, device=dev).to(dt); w0 = torch.randn(H, G, device=dev).to(dt)
    d1 = torch.randn(B, H, device=dev).to(dt); wout = torch.randn(G, H, device=dev).to(dt)
    dl = torch.randn(B, G, device=dev).to(dt); dh0 = torch.randn(B, H, device=dev).to(dt)
    for ns in (8, 32, 64):
        y = torch.empty(ns, B, H, device=dev)
        print(f"G={G} enc_fwd ns={ns}: {bench(lambda: ops.linear_fwd(x, w0, y, nslab=ns)):.1f} us")
        print(f"G={G} dec_dgrad ns={ns}: {bench(lambda: ops.linear_dgrad(dl, wout, y, nslab=ns)):.1f} us")
    yl = torch.empty(B, G, device=dev)
    print(f"G={G} dec_fwd: {bench(lambda: ops.linear_fwd(d1, wout, yl)):.1f} us")
    gw = torch.empty(G, H, device=dev)
    print(f"G={G} dec_wgrad: {bench(lambda: ops.linear_wgrad(dl, d1, gw)):.1f} us")
    gw2 = torch.empty(H, G, device=dev)
    print(f"G={G} enc_wgrad: {bench(lambda: ops.linear_wgrad(dh0, x, gw2)):.1f} us")
z = torch.zeros(1, device=dev)
print(f"empty-ish kernel (fill 1 float): {bench(lambda: z.fill_(1.0)):.1f} us")
