"""Experiment (GPU box): the gene-width GEMMs at hidden width 820 (the reference's default, C3) whole against split into
768 columns + a 52-column strip.  usage: python tools/dbg/strip820.py [H]"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from celligner2_amd import ops

def bench(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

B, G = 4096, 19000
H = int(sys.argv[1]) if len(sys.argv) > 1 else 820
cut = H // 256 * 256
Hp = (H + 7) // 8 * 8
dev, dt = "cuda", torch.bfloat16
x = torch.randn(B, G, device=dev).to(dt)
w0 = torch.randn(H, G, device=dev).to(dt)
d1 = torch.zeros(B, Hp, device=dev, dtype=dt)[:, :H]; d1.copy_(torch.randn(B, H, device=dev))
dh0 = torch.zeros(B, Hp, device=dev, dtype=dt)[:, :H]; dh0.copy_(torch.randn(B, H, device=dev))
wout = torch.zeros(G, Hp, device=dev, dtype=dt)[:, :H]; wout.copy_(torch.randn(G, H, device=dev))
dl = torch.randn(B, G, device=dev).to(dt)
print("tiles: fwd", ops.gemm_tile(dt, B, H, G), ops.gemm_tile(dt, B, cut, G), ops.gemm_tile(dt, B, H - cut, G))
for ns in (1, 2, 4):
    y = torch.empty(ns, B, H, device=dev)
    t_full = bench(lambda: ops.linear_fwd(x, w0, y, nslab=ns))
    ya, yb = y[:, :, :cut], y[:, :, cut:]
    for ns2 in (ns, 4 * ns, 16):
        yb2 = torch.empty(ns2, B, H - cut, device=dev)
        def split():
            ops.linear_fwd(x, w0[:cut], ya, nslab=ns)
            ops.linear_fwd(x, w0[cut:], yb2, nslab=ns2)
        print(f"enc0_fwd ns={ns}: whole {t_full:7.1f} us   {cut}+{H - cut} (strip ns={ns2}) {bench(split):7.1f} us")
    yd = torch.empty(ns, B, H, device=dev)
    t_full = bench(lambda: ops.linear_dgrad(dl, wout, yd, nslab=ns))
    print(f"head_dgrad ns={ns}: whole {t_full:7.1f} us")
yw = torch.empty(G, H, device=dev)
t_full = bench(lambda: ops.linear_wgrad(dl, d1, yw))
def splitw():
    ops.linear_wgrad(dl, d1[:, :cut], yw[:, :cut])
    ops.linear_wgrad(dl, d1[:, cut:], yw[:, cut:])
print(f"head_wgrad [G,{H}]: whole {t_full:7.1f} us   split {bench(splitw):7.1f} us")
ye = torch.empty(H, G, device=dev)
t_full = bench(lambda: ops.linear_wgrad(dh0, x, ye))
def splite():
    ops.linear_wgrad(dh0[:, :cut], x, ye[:cut])
    ops.linear_wgrad(dh0[:, cut:], x, ye[cut:])
print(f"enc0_wgrad [{H},G]: whole {t_full:7.1f} us   split {bench(splite):7.1f} us")
