"""Experiment (GPU box): the decoder forward [4096 x 19000] at a reduction length of 820: rows padded to 824 (a partial last
K-tile, staged through registers) against rows padded to 832 (13 whole K-tiles)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from celligner2_amd import ops

def bench(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

B, G, H = 4096, 19000, 820
dev, dt = "cuda", torch.bfloat16
for ld in (824, 832, 896, 1024):
    d = torch.zeros(B, ld, device=dev, dtype=dt); d[:, :H] = torch.randn(B, H, device=dev).to(dt)
    w = torch.zeros(G, ld, device=dev, dtype=dt); w[:, :H] = torch.randn(G, H, device=dev).to(dt)
    y16 = torch.empty(B, G, device=dev, dtype=torch.float16)
    y32 = torch.empty(B, G, device=dev)
    for K in sorted({(H + 7) // 8 * 8, ld}):
        if K > ld: continue
        dv = d.as_strided((B, K), (ld, 1)); wv = w.as_strided((G, K), (ld, 1))
        t16 = bench(lambda: ops.linear_fwd_f16out(dv, wv, y16))
        t32 = bench(lambda: ops.linear_fwd(dv, wv, y32))
        print(f"pitch {ld} K {K}: f16 out {t16:7.1f} us   f32 out {t32:7.1f} us")
# the input gradient [4096 x 820] = dl [4096 x 19000] . w [19000 x 820]: the weight's pitch
dl = torch.randn(B, G, device=dev).to(dt)
for ld in (824, 832, 1024):
    w = torch.zeros(G, ld, device=dev, dtype=dt)[:, :H]
    w.copy_(torch.randn(G, H, device=dev))
    y = torch.empty(4, B, H, device=dev)
    print(f"dgrad, weight pitch {ld}: {bench(lambda: ops.linear_dgrad(dl, w, y, nslab=4)):7.1f} us")
