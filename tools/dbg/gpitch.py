"""Experiment (GPU box): the gene-width GEMMs of C4 with the 19 000-gene rows 38 000 bytes apart (as now) against rows padded
to a multiple of 128 bytes (19 008 elements)."""
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

B, G, H = 4096, 19000, 1024
dev, dt = "cuda", torch.bfloat16
def mat(rows, cols, ld, dtype=dt):
    t = torch.zeros(rows, ld, device=dev, dtype=dtype)[:, :cols]
    t.copy_(torch.randn(rows, cols, device=dev))
    return t
for ld in (19000, 19008, 19072):
    x, w0, dl = mat(B, G, ld), mat(H, G, ld), mat(B, G, ld)
    x16, w16 = mat(B, G, ld, torch.float16), mat(H, G, ld, torch.float16)
    d1, dh0, wout = mat(B, H, H), mat(B, H, H), mat(G, H, H)
    y4 = torch.empty(4, B, H, device=dev)
    yl = torch.zeros(B, ld, device=dev, dtype=torch.float16)[:, :G]
    gw, ge = torch.empty(G, H, device=dev), torch.empty(H, G, device=dev)
    d1t, dh0t = d1.t().contiguous(), dh0.t().contiguous()
    rows = torch.arange(B, device=dev, dtype=torch.int32)
    print(f"gene pitch {ld}:",
          f"enc0_fwd(f16) {bench(lambda: ops.linear_fwd(x16, w16, y4, nslab=4)):6.1f}",
          f"enc0_fwd_rows(f16) {bench(lambda: ops.linear_fwd_rows(x16, rows, B, w16, y4, nslab=4)):6.1f}",
          f"head_fwd->f16[pitch] {bench(lambda: ops.linear_fwd_f16out(d1, wout, yl)):6.1f}",
          f"head_dgrad {bench(lambda: ops.linear_dgrad(dl, wout, y4, nslab=4)):6.1f}",
          f"head_wgrad_bt {bench(lambda: ops.linear_wgrad_bt(dl, d1t, gw)):6.1f}",
          f"enc0_wgrad_rows_at {bench(lambda: ops.linear_wgrad_rows_at(dh0t, x, rows, ge)):6.1f}", flush=True)
